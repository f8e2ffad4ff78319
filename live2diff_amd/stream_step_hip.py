"""Device-side per-frame pipeline step.

`HipStreamStep` turns what `StreamAnimateDiffusionDepth.predict_x0_batch` does around the UNet call (reference
pipeline_stream_animation_depth.py:573-623: batch assembly, `scheduler_step_batch` :387-401, the stream-batch shift
register :590-601, `update_attn_bias` :416-438) into ops of the SAME static plan as the UNet itself:

    [new frame's latents -> row 0 of the UNet's static input]  (the only per-frame host action: two small copies)
    UNet streaming forward (the ~730 launches of HipStreamingUNet's plan)
    randn            re-noising tensor for rows 1..N-1 (skipped when the caller injects noise / do_add_noise=False)
    stream_shift     x0 of the N rows, output = x0[N-1], rows 1..N-1 of the next batch, depth rows shifted
    ring_update      attn_bias / pe_idx / update_idx advanced in place on the UNet's input buffers

so a frame needs no host-side tensor arithmetic, no `.any()` / `.sum()` sync, no H2D copy of the ring-buffer state, and
the whole step can be one hipGraph.  The ring-buffer and LCM arithmetic is pinned bit-exactly to the host restatement,
which is itself pinned to the trace / goldens captured from the reference (tests/golden/state_machine.npz, lcm.npz).
"""
import ctypes
from typing import List, Optional

import torch

from . import _lib, ops
from .pipeline_stream_animation_depth import ring_buffer_init
from .unet_hip import HipStreamingUNet


class HipStreamStep:
    def __init__(self, unet: HipStreamingUNet, kv_cache: List[torch.Tensor], timesteps: torch.Tensor,
                 prompt_embeds: torch.Tensor, alpha_prod_t_sqrt: torch.Tensor, beta_prod_t_sqrt: torch.Tensor,
                 c_skip: torch.Tensor, c_out: torch.Tensor, do_add_noise: bool = True, seed: int = 0,
                 inject_noise: bool = False, use_graph: bool = False, ring_state=None):
        """timesteps [N] int64; prompt_embeds [N,77,D]; the four scheduler tensors hold one value per row (any shape
        with N elements), in the pipeline's dtype -- their fp16-rounded values are what the reference multiplies by.
        `ring_state` = (attn_bias, pe_idx, update_idx) host tensors to start from (the pipeline's current `_rb` when
        frames already ran on the host path); None = the initial state of a fresh stream."""
        self.unet, self.kv = unet, kv_cache
        cfg, N = unet.cfg, unet.N
        self.N, self.per = N, cfg.in_channels * unet.h * unet.w
        dev = unet.device
        st = unet._plan("stream", kv_cache)
        unet._bind_caches(st, kv_cache)
        self.st = st
        st.in_t.copy_(timesteps.reshape(-1).expand(N))
        self.set_prompt(prompt_embeds)                # also marks the conditioning launches stale
        rb = ring_state if ring_state is not None else ring_buffer_init(N, cfg.window_size, cfg.sink_size)
        st.in_bias.copy_(rb[0].to(torch.float16))
        st.in_pe_idx.copy_(rb[1])
        st.in_upd.copy_(rb[2])
        f = lambda t: t.reshape(-1).to(torch.float16).float()        # the fp16 values the reference's tensors hold
        self.scal = torch.stack([f(alpha_prod_t_sqrt), f(beta_prod_t_sqrt), f(c_skip), f(c_out)], dim=1).contiguous().to(dev)
        assert self.scal.shape == (N, 4)
        self.x0_out = torch.zeros(self.per, dtype=torch.float16, device=dev)
        self.noise = torch.zeros(max(N - 1, 1) * self.per, dtype=torch.float16, device=dev) if do_add_noise else None
        self.frame_ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        self.use_graph = use_graph
        pl = _lib.OpList()
        for j in range(len(st.pl)):                                   # the UNet's own launches, same records
            c = _lib.L2dOp()
            ctypes.memmove(ctypes.byref(c), ctypes.byref(st.pl[j]), ctypes.sizeof(_lib.L2dOp))
            pl.append(c)
        if do_add_noise and N > 1 and not inject_noise:
            pl.append(*ops.randn(self.noise, seed=seed, frame_ctr=self.frame_ctr))
        pl.append(*ops.stream_shift(st.in_sample, st.out_sample, self.scal, self.x0_out, N=N, per=self.per,
                                    noise=(self.noise if (do_add_noise and N > 1) else None), depth=st.in_depth))
        pl.append(*ops.ring_update(st.in_bias, st.in_pe_idx, st.in_upd, N=N, L=cfg.window_size, sink=cfg.sink_size,
                                   frame_ctr=self.frame_ctr))
        self.pl = pl
        self._graph = None
        self._warm = False

    def set_prompt(self, prompt_embeds: torch.Tensor):
        """(Re)load the static text input of the plan (reference update_prompt, pipeline :368-376)."""
        st, cfg = self.st, self.unet.cfg
        st.in_enc[:, : st.text_len, : cfg.cross_attention_dim].copy_(prompt_embeds)
        self.unet.invalidate_text_cache()

    # ---- state the caller owns in the reference (x_t_latent_buffer / depth_latent_buffer, set by `prepare`)
    def load_buffers(self, x_t_latent_buffer: Optional[torch.Tensor], depth_latent_buffer: Optional[torch.Tensor]):
        if self.N > 1:
            self.st.in_sample[1:].copy_(x_t_latent_buffer.reshape(self.N - 1, self.unet.cfg.in_channels, -1))
            self.st.in_depth[1:].copy_(depth_latent_buffer.reshape(self.N - 1, self.unet.cfg.in_channels, -1))

    @property
    def attn_bias(self):
        return self.st.in_bias

    @property
    def pe_idx(self):
        return self.st.in_pe_idx

    @property
    def update_idx(self):
        return self.st.in_upd

    @property
    def x_t_latent_buffer(self):
        return self.st.in_sample[1:].view(self.N - 1, self.unet.cfg.in_channels, 1, self.unet.h, self.unet.w)

    @torch.no_grad()
    def step(self, x_t_latent: torch.Tensor, depth_latent: torch.Tensor) -> torch.Tensor:
        """x_t_latent / depth_latent: the NEW frame's noised latent and depth latent [1,4,1,h,w].  Returns the x0
        prediction that leaves the stream batch this frame, [1,4,1,h,w] (a view of a static buffer)."""
        c = self.unet.cfg.in_channels
        self.st.in_sample[0].copy_(x_t_latent.reshape(c, -1))
        self.st.in_depth[0].copy_(depth_latent.reshape(c, -1))
        self.unet._ensure_cond(self.st)        # time embedding + text K / V^T: only after set_prompt / construction
        if self.use_graph and self._warm:
            if self._graph is None:             # capture on a side stream (the legacy default stream cannot be captured)
                side = torch.cuda.Stream(device=self.unet.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._graph = _lib.Graph(self.pl, stream=int(side.cuda_stream))
                torch.cuda.current_stream().wait_stream(side)
            self._graph.launch()
        else:
            self.pl.run()           # (the first frame always runs directly: one-time kernel attribute / device queries
            self._warm = True       # of the launchers are not allowed inside a stream capture)
        return self.x0_out.view(1, c, 1, self.unet.h, self.unet.w)
