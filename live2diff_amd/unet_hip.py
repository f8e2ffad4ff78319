"""HipStreamingUNet -- the object that occupies the reference's `stream.unet` slot.

Boundary (reference): `StreamAnimateDiffusionDepth.unet`, called at
live2diff/pipeline_stream_animation_depth.py:456-466 with the signature of
`UNet3DConditionStreamingModel.forward` (unet_depth_streaming.py:429-448) /
`UNet2DConditionModelDepthEngine.__call__` (acceleration/tensorrt/engine.py:142-153); the warm-up twin
(`unet_warmup`, unet_depth_warmup.py:407-590, called at pipeline :320-328) is `HipStreamingUNet.warmup`.

Design (MI355X-first, not a module-by-module translation):
  * one channels-last fp16 activation layout `[B*T, C]` end to end; the only layout conversions of a step
    are on the 4-channel latents at the boundary;
  * weights are ingested once from a reference-keyed `state_dict` and re-packed for the kernels
    (3x3 taps-major, q|k fused, q|k|v fused for the temporal layers, GEGLU value/gate interleaved, all
    time_emb_proj / all text K,V projections concatenated into single GEMMs);
  * a UNet step is a static *plan* -- an array of `l2d_op` records, built once per (H, W, N, L) -- that the
    native executor replays (`l2d_run_ops`, or a captured hipGraph): ~650 kernels, no Python in the loop;
  * the KV-cache stays in the reference interchange layout `[N,2,T,L,C]` and is updated IN PLACE, so the
    caller's `kv_cache_list` semantics (pipeline :468-469) hold and nothing cache-sized is ever copied.
No torch compute ops are used on the path (torch provides HBM allocations, the stream, and the tiny
host-to-static-buffer input copies).
"""
from types import SimpleNamespace
from typing import Dict, List, Optional

import os
import torch

from . import _lib, ops
from .config import UNetConfig, motion_module_layout
from .ops import round_up

TEXT_PAD = 80   # 77 CLIP tokens padded to a multiple of 4 (igemm stores 4 channels per lane)


class UNetOutput(dict):
    """`out["sample"]`, `out["kv_cache"]`, `out.sample`, `out.kv_cache`, `out[0]`
    (reference UNet3DConditionStreamingOutput, unet_depth_streaming.py:29-32)."""

    def __init__(self, sample, kv_cache):
        super().__init__(sample=sample, kv_cache=kv_cache)
        self.sample, self.kv_cache = sample, kv_cache

    def __getitem__(self, k):
        if isinstance(k, int):
            return (self.sample, self.kv_cache)[k]
        return dict.__getitem__(self, k)


def sinusoid_pe(max_len: int, dim: int, device) -> torch.Tensor:
    """reference positional_encoding.py:12-16"""
    import math

    pos = torch.arange(max_len, dtype=torch.float32, device=device).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32, device=device) * (-math.log(10000.0) / dim))
    pe = torch.zeros(max_len, dim, device=device)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class _Arena:
    """Size-keyed free list of device buffers: intermediates of a static plan reuse HBM (and stay hot in the
    256 MB Infinity Cache) instead of every op getting a private allocation."""

    def __init__(self, device):
        self.device = device
        self.free: Dict[tuple, List[torch.Tensor]] = {}
        self.all: List[torch.Tensor] = []

    def alloc(self, numel: int, dtype=torch.float16) -> torch.Tensor:
        key = (int(numel), dtype)
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        t = torch.empty(int(numel), dtype=dtype, device=self.device)
        self.all.append(t)
        return t

    def release(self, t: Optional[torch.Tensor]):
        if t is not None:
            self.free.setdefault((t.numel(), t.dtype), []).append(t)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.all)


class _Act:
    """channels-last activation: buf holds [B*H*W, C] halfs (ld == C); `producer` = the igemm op that wrote it (if any)."""
    __slots__ = ("buf", "C", "H", "W", "producer")

    def __init__(self, buf, C, H, W, producer=None):
        self.buf, self.C, self.H, self.W, self.producer = buf, C, H, W, producer


class PackedWeights:
    """Packed weights outside an instance: `W` (name -> device tensor, what `_pack_weights` produced) + `meta` (the strings a
    packed-weight file carries).  Produced by `HipStreamingUNet.packed_state()`, accepted by the constructor -- the unit of the
    multi-GPU weight replication (parallel.replicate_packed_weights: rank 0 packs once, every rank receives the packed tensors)."""
    __slots__ = ("W", "meta")

    def __init__(self, W, meta):
        self.W, self.meta = W, meta


class HipStreamingUNet:
    def __init__(self, state_dict, cfg: UNetConfig, height: int, width: int,
                 denoising_steps_num: int, device="cuda", warmup_frames: Optional[int] = None, use_graph: bool = False,
                 tattn_variant: int = 0, text_len: int = 77, fresh_output: bool = False):
        """height/width are LATENT sizes (image / 8). `state_dict` uses the reference key names; it may also be the
        path of a packed-weight file written by `save_packed`, or ANOTHER HipStreamingUNet of the same
        configuration and latent size whose packed weights this instance then shares (read-only replicas are per GPU, not per
        stream: several independent frame streams on one GPU -- each with its own plan buffers and KV caches, each on its own
        HIP stream -- fill each other's launch gaps)."""
        assert cfg.num_heads == 8 and cfg.temporal_heads == 8
        assert height % 8 == 0 and width % 8 == 0, "latent size must be divisible by 8 (3 down-samplings, T%4==0)"
        self.cfg, self.h, self.w, self.N = cfg, height, width, denoising_steps_num
        self.device = torch.device(device)
        self.F = cfg.sink_size if warmup_frames is None else warmup_frames
        self.use_graph = use_graph
        self.fresh_output = fresh_output   # True: return a private copy of the prediction (the reference's PyTorch path returns a
        #                                    fresh tensor); False (default): a view of the static output buffer, like a TensorRT binding
        self.tattn_variant = tattn_variant
        self.igemm_splitk_off = False       # tuning knob: disable split-K schedules
        self.cond_cache = True           # False: re-run the conditioning launches every call (tests)
        assert 1 <= text_len <= TEXT_PAD
        self.text_len = text_len           # static number of text tokens (77 for CLIP)
        self.dtype = torch.float16
        self.config = SimpleNamespace(in_channels=cfg.in_channels)      # read by the reference wrapper (:524)
        self.device_name = "dry-run" if ops.DRY_RUN else _lib.device_name()   # raises unless a gfx950 is present
        self.mm_layout = motion_module_layout(cfg, height, width)
        # levels whose stream batch is small enough for the weight-streaming GEMM (wsgemm.hip): N * T tokens <= L2D_WSGEMM_MAX_M and
        # samples made of whole 32-token tiles.  Decides the PACKING of those levels' layers (and with it the plan's kernels).
        ws_on = os.environ.get("L2D_WSGEMM", "1") != "0"             # A/B knob: 0 = the round-3 kernels everywhere
        # (levels of up to 1280 tokens take it by default, larger ones -- up to this bound -- per measured shape: ops.wsgemm_wanted)
        ws_max = int(os.environ.get("L2D_WSGEMM_MAX_M", "4608"))
        self.ws_levels = [ws_on and denoising_steps_num * (height >> l) * (width >> l) <= ws_max and ((height >> l) * (width >> l)) % 32 == 0
                          for l in range(cfg.num_levels)]
        if isinstance(state_dict, HipStreamingUNet):
            o = state_dict
            if (o.cfg, o.h, o.w, o.device) != (cfg, self.h, self.w, self.device):
                raise ValueError("shared packed weights need the same configuration, latent size and device")
            if (o.N, o.ws_levels) != (self.N, self.ws_levels):
                # the packing depends on the stream batch too (which levels take the weight-streaming form, the per-shape skip
                # list keyed by M = N T): forms chosen for another N would be missing / forced here (as _load_packed checks)
                raise ValueError(f"shared packed weights were packed for denoising_steps_num = {o.N} (weight-streaming levels "
                                 f"{o.ws_levels}), this instance has {self.N} ({self.ws_levels}): re-pack")
            self.W, self.temb_offsets, self.text_offsets = o.W, o.temb_offsets, o.text_offsets
            self.n_map_blocks, self.temb_total, self.text_total, self.text_kp = o.n_map_blocks, o.temb_total, o.text_total, o.text_kp
        elif isinstance(state_dict, (str, os.PathLike)):
            self._load_packed(state_dict)          # a file written by save_packed(): skips the packing pass
        elif isinstance(state_dict, PackedWeights):
            self._adopt_packed(state_dict.W, state_dict.meta, "packed weights")     # received from another rank
        else:
            self._pack_weights(state_dict)
        self._plans = {}
        self._graph = {}

    # ------------------------------------------------------------------ reference-compatible surface
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def forward(self, *a, **k):
        return self(*a, **k)

    def set_info_for_attn(self, height: int, width: int, *a, **k):
        assert (height, width) == (self.h, self.w), "static shapes per instance (TensorRT precedent: models.py:289-291)"

    def prepare_cache(self, denoising_steps_num: int) -> List[torch.Tensor]:
        """Zero KV caches [N,2,h*w,L,C] fp16 in motion_module_idx order
        (reference unet_depth_streaming.py:283-302 + stream_motion_module.py:57-77)."""
        return [torch.zeros(denoising_steps_num, 2, hh * ww, self.cfg.window_size, c, dtype=torch.float16,
                            device=self.device) for (c, hh, ww, _l) in self.mm_layout]

    # ------------------------------------------------------------------ weights
    def _pack_weights(self, sd):
        cfg, dev = self.cfg, self.device
        g = lambda k: sd[k].to(device=dev)
        W = {}
        self.W = W

        def conv3(name):
            W[name + ".w"] = ops.pack_conv3x3(g(name + ".weight"))
            W[name + ".b"] = ops.f32(g(name + ".bias"))

        use_rg = os.environ.get("L2D_ROWGEMM", "1") != "0"     # A/B knob: 0 = every linear layer on igemm + separate norm launches
        RG_PLAIN_MAX_K = int(os.environ.get("L2D_ROWGEMM_PLAIN_MAX_K", "640"))
        RG_FF1_MAX_K = int(os.environ.get("L2D_ROWGEMM_FF1_MAX_K", "1280"))     # A/B knob: GEGLU GEMMs wider than this stay on igemm
        ws_lv = self.ws_levels
        chain_on = os.environ.get("L2D_ROWCHAIN", "1") != "0"        # A/B knob: 0 = the four separate launches of a block's tail

        def rg_ok(wname):
            n, k = sd[wname].shape[0], sd[wname][0].numel()
            return use_rg and ops.rowgemm_ok(n, k)

        def ws_ok(wname, lvl, n_mul=1, epi=0, pro=0, ntr=0, taps=1):
            """the weight-streaming GEMM (wsgemm.hip) takes this layer: a level of few tokens, 32-row weight tiles, 64-column chunks,
            and the in-frame tuner did not find the round-3 kernel faster for the shape (ops.wsgemm_wanted); n_mul: q | k | v"""
            n, k = sd[wname].shape[0] * n_mul, sd[wname][0].numel()
            if lvl is None or not ws_lv[lvl] or n % 32 or (k // taps) % 64:
                return False
            M_ = self.N * (self.h >> lvl) * (self.w >> lvl)
            return ops.wsgemm_wanted(taps, M_, k, n, ntr, epi, pro)

        def lin(name, bias=True, norm=None, old=False, lvl=None, gnorm=False):
            """Linear layer `name`; `norm` = the LayerNorm (or, `gnorm`, GroupNorm) whose output feeds it.  At the few-token levels
            (`lvl` in self.ws_levels) the weight-streaming packing (wsgemm.hip: fragment order, a LayerNorm folded into weight,
            bias and column sums) -- except behind a GroupNorm, whose per-group scale cannot move to the accumulator side.
            Else token-row GEMM packing (rowgemm.hip: fragment order, the norm's affine folded into weight and bias) when the shape
            allows, else -- and with `old` in addition -- the implicit-GEMM packing with the norm applied by its own launch."""
            if not gnorm and ws_ok(name + ".weight", lvl, pro=(1 if norm else 0)):
                W[name + ".ww"], wb, wcs = ops.pack_wsgemm(g(name + ".weight"), g(name + ".bias") if bias else None,
                                                           g(norm + ".weight") if norm else None, g(norm + ".bias") if norm else None)
                if wb is not None:
                    W[name + ".wb"] = wb
                if wcs is not None:
                    W[name + ".wcs"] = wcs
                return
            # Row GEMM where it fuses a norm, and for the narrow levels (K <= 640).  A plain Linear at K = 1280 stays on the
            # implicit-GEMM kernel: with 32-token row tiles every block ingests its whole weight band (80 KB per 32-row tile), and
            # the probe (profiles/round3_b_rowgemm_block_phases_before.txt) shows those launches bound by ~30 B/clk of ingest per CU.
            rg = rg_ok(name + ".weight") and (norm is not None or sd[name + ".weight"][0].numel() <= RG_PLAIN_MAX_K)
            if rg:
                W[name + ".rw"], rb = ops.pack_rowgemm(g(name + ".weight"), g(name + ".bias") if bias else None,
                                                       g(norm + ".weight") if norm else None, g(norm + ".bias") if norm else None)
                if rb is not None:
                    W[name + ".rb"] = rb
            if not rg or old:
                W[name + ".w"] = ops.pack_linear(g(name + ".weight"))
                if bias:
                    W[name + ".b"] = ops.f32(g(name + ".bias"))

        def norm(name):
            W[name + ".g"] = g(name + ".weight").to(torch.float16).contiguous()
            W[name + ".beta"] = g(name + ".bias").to(torch.float16).contiguous()

        def ff(name, norm, old=False, lvl=None):
            pw, pb = name + ".net.0.proj.weight", name + ".net.0.proj.bias"
            if ws_ok(pw, lvl, epi=1, pro=1) and sd[pw].shape[0] % 64 == 0:
                W[name + ".ww1"], W[name + ".wb1"], W[name + ".wcs1"] = ops.pack_wsgemm(g(pw), g(pb), g(norm + ".weight"), g(norm + ".bias"),
                                                                                         geglu=True)
                lin(name + ".net.2", old=old, lvl=lvl)
                return
            rg = rg_ok(pw) and sd[pw].shape[0] % 64 == 0 and sd[pw][0].numel() <= RG_FF1_MAX_K
            if rg:
                W[name + ".rw1"], W[name + ".rb1"] = ops.pack_rowgemm(g(pw), g(pb), g(norm + ".weight"), g(norm + ".bias"), geglu=True)
                if chain_on and sd[pw][0].numel() == ops.ROWCHAIN_C:
                    # the token-resident block tail (rowchain.hip) streams FF2 in the row GEMM's fragment order too (K = 4 C).  Its own
                    # keys (".chw" / ".chb"): `linear()` picks the row GEMM for a layer by the presence of ".rw", and a plain K = 1280
                    # Linear must stay on the implicit-GEMM kernel wherever the chain does not run (round-5 advisor finding)
                    W[name + ".net.2.chw"], W[name + ".net.2.chb"] = ops.pack_rowgemm(g(name + ".net.2.weight"), g(name + ".net.2.bias"))
            if not rg or old:
                W[name + ".w1"], W[name + ".b1"] = ops.pack_geglu(g(pw), g(pb))
            lin(name + ".net.2", old=old, lvl=lvl)

        self.temb_names, self.temb_offsets = [], {}
        temb_w, temb_b = [], []
        self.text_offsets = {}
        text_k, text_v = [], []

        def level_of(name):
            if name.startswith("mid_block"):
                return cfg.num_levels - 1
            lvl = int(name.split(".")[1])
            return cfg.num_levels - 1 - lvl if name.startswith("up_blocks") else lvl

        def conv3cc(name, lvl_out, ups=0) -> bool:
            """3x3 stride-1 conv whose OUTPUT lives at level `lvl_out`: the patch-resident / register-streamed packing of cconv.hip
            where the plan wants that kernel (ops.cconv_wanted: measured per shape class); the K-group count of the packing is the
            stream plan's (ops.cconv_schedule on the stream batch), the warm-up plan re-uses it"""
            cw = sd[name + ".weight"]
            if lvl_out is None:
                return False
            Ho, Wo = self.h >> lvl_out, self.w >> lvl_out
            if cw.shape[1] % 64 or not ops.cconv_wanted(self.N, Ho, Wo, cw.shape[1], cw.shape[0], ups):
                return False
            kg = ops.cconv_schedule(self.N, Ho, Wo, cw.shape[0], cw.shape[1])[1]
            W[name + ".cw"] = ops.pack_cconv(g(name + ".weight"), kg)
            W[name + ".b"] = ops.f32(g(name + ".bias"))
            return True

        def conv3ws(name, lvl):
            """resnet 3x3 conv: cconv packing where that kernel is wanted, weight-streaming packing at the few-token levels, else the
            implicit-GEMM / patch-conv packing"""
            cw = sd[name + ".weight"]
            if conv3cc(name, lvl):
                return
            # (the kernel's loader walks 8 NL pixels per DMA instruction with at most two row wraps: W >= 8, wsgemm.hip; narrower
            # levels -- tall / narrow latents such as 64 x 32 -- stay on the implicit-GEMM / patch kernels like in round 3)
            if (lvl is not None and (self.w >> lvl) >= 8 and cw.shape[0] % 32 == 0 and cw.shape[1] % 64 == 0
                    and ws_ok(name + ".weight", lvl, taps=9)):
                W[name + ".ww"] = ops.pack_wsgemm_conv3x3(g(name + ".weight"))
                W[name + ".b"] = ops.f32(g(name + ".bias"))
            else:
                conv3(name)

        def concat_parts_ok(name):
            """wsgemm takes whole 64-channel chunks from EACH input of a two-pointer concat (up blocks: hidden | skip)"""
            cout, cin = sd[name + ".conv_shortcut.weight"].shape[:2]
            if not name.startswith("up_blocks"):
                return True                                   # one input
            i, j = int(name.split(".")[1]), int(name.split(".")[3])
            if j > 0:
                c1 = cout
            else:
                c1 = sd[f"up_blocks.{i - 1}.resnets.0.conv1.weight" if i > 0 else "mid_block.resnets.0.conv1.weight"].shape[0]
            return c1 % 64 == 0 and (cin - c1) % 64 == 0

        def resnet(name):
            lvl = level_of(name)
            norm(name + ".norm1"); conv3ws(name + ".conv1", lvl); norm(name + ".norm2"); conv3ws(name + ".conv2", lvl)
            if (name + ".conv_shortcut.weight") in sd:        # (two-input concat GEMM)
                if ws_ok(name + ".conv_shortcut.weight", lvl) and concat_parts_ok(name):
                    lin(name + ".conv_shortcut", lvl=lvl)
                else:
                    W[name + ".conv_shortcut.w"] = ops.pack_linear(g(name + ".conv_shortcut.weight"))
                    W[name + ".conv_shortcut.b"] = ops.f32(g(name + ".conv_shortcut.bias"))
            self.temb_offsets[name] = sum(t.shape[0] for t in temb_w)
            temb_w.append(g(name + ".time_emb_proj.weight").to(torch.float16))
            temb_b.append(g(name + ".time_emb_proj.bias").float())

        def spatial(name):
            # the mid block sits at the lowest resolution, where T = h w / 64 need not be a multiple of the row GEMM's 32-token
            # tile (its transposed V output and GroupNorm prologue need that): it keeps the implicit-GEMM packing as well
            # (likewise any level of THIS instance where T % 32 != 0: small test latents; a packed-weight file written there
            # holds both forms, one written at an SD resolution holds the second form for the mid block only)
            lvl = level_of(name)
            Tl = (self.h >> lvl) * (self.w >> lvl)
            old = name.startswith("mid_block") or Tl % 32 != 0
            b = name + ".transformer_blocks.0"
            norm(name + ".norm"); lin(name + ".proj_in", norm=name + ".norm", old=old, gnorm=True); lin(name + ".proj_out", old=old, lvl=lvl)
            for n in ("norm1", "norm2", "norm3"):
                norm(b + "." + n)
            wq, wk, wv = (g(b + f".attn1.to_{c}.weight") for c in "qkv")
            if Tl % 128 == 0 and ws_ok(b + ".attn1.to_q.weight", lvl, n_mul=3, pro=1, ntr=wq.shape[0]):
                # q | k | v in one weight-streaming launch behind norm1 (V leaves transposed: a sample is whole 128-token tiles)
                W[b + ".attn1.qkv.ww"], W[b + ".attn1.qkv.wb"], W[b + ".attn1.qkv.wcs"] = ops.pack_wsgemm(
                    torch.cat([wq, wk, wv], 0), None, g(b + ".norm1.weight"), g(b + ".norm1.bias"))
            elif rg_ok(b + ".attn1.to_q.weight"):
                # q | k | v in one launch behind norm1 (V leaves transposed): rowgemm.hip
                W[b + ".attn1.qkv.rw"], W[b + ".attn1.qkv.rb"] = ops.pack_rowgemm(
                    torch.cat([wq, wk, wv], 0), None, g(b + ".norm1.weight"), g(b + ".norm1.bias"))
            if (b + ".attn1.qkv.ww") not in W and (not rg_ok(b + ".attn1.to_q.weight") or old):
                W[b + ".attn1.qk"] = ops.pack_linear(torch.cat([wq, wk], 0))
                W[b + ".attn1.v"] = ops.pack_linear(wv)
            lin(b + ".attn1.to_out.0", old=old, lvl=lvl)
            lin(b + ".attn2.to_q", bias=False, norm=b + ".norm2", old=old, lvl=lvl)
            self.text_offsets[name] = sum(t.shape[0] for t in text_k)
            text_k.append(g(b + ".attn2.to_k.weight").to(torch.float16))
            text_v.append(g(b + ".attn2.to_v.weight").to(torch.float16))
            lin(b + ".attn2.to_out.0", old=old, lvl=lvl)
            ff(b + ".ff", b + ".norm3", old=old, lvl=lvl)

        self.pe_tables = {}

        def motion(name, C):
            t = name + ".temporal_transformer"
            lvl = level_of(name)
            norm(t + ".norm"); lin(t + ".proj_in", norm=t + ".norm", gnorm=True); lin(t + ".proj_out", lvl=lvl)
            b = t + ".transformer_blocks.0"
            L = cfg.window_size
            if C not in self.pe_tables:
                self.pe_tables[C] = sinusoid_pe(max(cfg.temporal_max_len, L), C, dev)
            pe = self.pe_tables[C][:L]
            for j in range(2):
                a = b + f".attention_blocks.{j}"
                wq, wk, wv = g(a + ".to_q.weight"), g(a + ".to_k.weight"), g(a + ".to_v.weight")
                if ws_ok(a + ".to_q.weight", lvl, n_mul=3, pro=1):
                    W[a + ".qkv.ww"], W[a + ".qkv.wb"], W[a + ".qkv.wcs"] = ops.pack_wsgemm(
                        torch.cat([wq, wk, wv], 0), None, g(b + f".norms.{j}.weight"), g(b + f".norms.{j}.bias"))
                elif rg_ok(a + ".to_q.weight"):
                    W[a + ".qkv.rw"], W[a + ".qkv.rb"] = ops.pack_rowgemm(
                        torch.cat([wq, wk, wv], 0), None, g(b + f".norms.{j}.weight"), g(b + f".norms.{j}.bias"))
                else:
                    W[a + ".qkv"] = ops.pack_linear(torch.cat([wq, wk, wv], 0))
                # pre-projected positional encodings (reference prepare_pe_buffer, stream_motion_module.py:79-97)
                for nm, w_ in (("q_pe", wq), ("k_pe", wk), ("v_pe", wv)):
                    W[a + "." + nm] = (pe @ w_.float().t()).to(torch.float16).contiguous()
                lin(a + ".to_out.0", lvl=lvl)
                norm(b + f".norms.{j}")
            norm(b + ".ff_norm")
            ff(b + ".ff", b + ".ff_norm", lvl=lvl)

        conv3("conv_in")
        conv3("flow_conv_in.conv_in")
        i = 0
        while f"flow_conv_in.blocks.{i}.weight" in sd:
            conv3(f"flow_conv_in.blocks.{i}")
            i += 1
        self.n_map_blocks = i
        conv3("flow_conv_in.conv_out")
        W["time_embedding.linear_1.w"] = g("time_embedding.linear_1.weight").to(torch.float16).contiguous()
        W["time_embedding.linear_1.b"] = ops.f32(g("time_embedding.linear_1.bias"))
        W["time_embedding.linear_2.w"] = g("time_embedding.linear_2.weight").to(torch.float16).contiguous()
        W["time_embedding.linear_2.b"] = ops.f32(g("time_embedding.linear_2.bias"))
        nl, ch = cfg.num_levels, cfg.block_out_channels
        for i in range(nl):
            for j in range(cfg.layers_per_block):
                resnet(f"down_blocks.{i}.resnets.{j}")
                if i != nl - 1:
                    spatial(f"down_blocks.{i}.attentions.{j}")
                motion(f"down_blocks.{i}.motion_modules.{j}", ch[i])
            if i != nl - 1:
                conv3(f"down_blocks.{i}.downsamplers.0.conv")
        resnet("mid_block.resnets.0"); spatial("mid_block.attentions.0"); resnet("mid_block.resnets.1")
        rev = list(reversed(ch))
        for i in range(nl):
            for j in range(cfg.layers_per_block + 1):
                resnet(f"up_blocks.{i}.resnets.{j}")
                if i != 0:
                    spatial(f"up_blocks.{i}.attentions.{j}")
                motion(f"up_blocks.{i}.motion_modules.{j}", rev[i])
            if i != nl - 1:
                if not conv3cc(f"up_blocks.{i}.upsamplers.0.conv", nl - 2 - i, ups=1):       # (output: one level up)
                    conv3(f"up_blocks.{i}.upsamplers.0.conv")
        norm("conv_norm_out"); conv3("conv_out")
        W["temb_all.w"] = torch.cat(temb_w, 0).contiguous()          # [sum Cout, 4*c0]
        W["temb_all.b"] = torch.cat(temb_b, 0).contiguous()
        self.temb_total = W["temb_all.w"].shape[0]
        W["text_k.w"] = ops.pack_linear(torch.cat(text_k, 0))         # [sum C, Kp(text)]
        W["text_v.w"] = ops.pack_linear(torch.cat(text_v, 0))
        self.text_total = W["text_k.w"].shape[0]
        self.text_kp = W["text_k.w"].shape[1]

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.W.values())

    # ------------------------------------------------------------------ packed-weight cache
    PACK_FORMAT = 4      # bump when _pack_weights changes layout (packed conv / GEGLU order, fused projections, ...)

    def save_packed(self, path) -> None:
        """Write the packed weights (what `_pack_weights` produced from the reference-keyed state dict: merged
        DreamBooth / LoRA weights, permuted, fused and padded for the kernels, PE tables pre-projected) as one
        safetensors file.  The analogue of the reference's TensorRT engine cache (wrapper.py:300-332, :505-560): a style
        switch that was seen before skips the conversion.  Packed weights depend on the weights, on the window length and on
        WHICH KERNEL serves each layer: the levels with few stream tokens (`ws_levels`: denoising steps x latent size) hold the
        weight-streaming forms, levels whose samples are not whole 32-token tiles the implicit-GEMM forms, and the L2D_ROWGEMM*
        knobs move layers between kernels.  The file records all of that; `_load_packed` refuses a file packed for another layout
        with a "re-pack" error instead of failing on a missing tensor later."""
        from safetensors.torch import save_file
        save_file({k: v.detach().cpu().contiguous() for k, v in self.W.items()}, str(path), metadata=self._packed_meta())

    def _packed_meta(self) -> dict:
        import json
        return dict(format=str(self.PACK_FORMAT), abi=str(_lib.ABI_VERSION), window=str(self.cfg.window_size),
                    block_out_channels=json.dumps(list(self.cfg.block_out_channels)),
                    temb_offsets=json.dumps(self.temb_offsets), text_offsets=json.dumps(self.text_offsets),
                    n_map_blocks=str(self.n_map_blocks), layout=json.dumps(self._pack_layout()))

    def packed_state(self) -> "PackedWeights":
        """The packed weights of this instance as (tensors, metadata) -- what a packed-weight file holds, without the file."""
        return PackedWeights(self.W, self._packed_meta())

    def _load_packed(self, path) -> None:
        import json

        from safetensors import safe_open
        with safe_open(str(path), framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            self._check_packed_meta(meta, path)
            # (a copy in every case: on the CPU -- dry-run plans of the test-suite -- get_tensor() returns a view into the file buffer
            #  whose address need not be 16-byte aligned, which the launch validation requires of every pointer)
            W = {k: (f.get_tensor(k).to(self.device) if torch.device(self.device).type != "cpu" else f.get_tensor(k).clone())
                 for k in f.keys()}
        self._adopt_packed(W, meta, path, checked=True)

    def _check_packed_meta(self, meta, path) -> None:
        import json
        if int(meta.get("format", -1)) != self.PACK_FORMAT or int(meta.get("abi", -1)) != _lib.ABI_VERSION:
            raise ValueError(f"{path}: packed-weight format {meta.get('format')} / ABI {meta.get('abi')} does not match "
                             f"this build ({self.PACK_FORMAT} / {_lib.ABI_VERSION}): re-pack from the state dict")
        if json.loads(meta.get("layout", "null")) != self._pack_layout():
            raise ValueError(f"{path}: packed for kernel layout {meta.get('layout')}, this instance needs {json.dumps(self._pack_layout())} "
                             "(latent size / denoising steps / L2D_ROWGEMM* / L2D_WSGEMM* differ): re-pack from the state dict")
        if int(meta["window"]) != self.cfg.window_size or json.loads(meta["block_out_channels"]) != list(self.cfg.block_out_channels):
            raise ValueError(f"{path}: packed for window {meta['window']} / widths {meta['block_out_channels']}, "
                             f"this instance is window {self.cfg.window_size} / {list(self.cfg.block_out_channels)}")

    def _adopt_packed(self, W, meta, path, checked: bool = False) -> None:
        import json
        if not checked:
            self._check_packed_meta(meta, path)
        self.W = W
        self.temb_offsets = {k: int(v) for k, v in json.loads(meta["temb_offsets"]).items()}
        self.text_offsets = {k: int(v) for k, v in json.loads(meta["text_offsets"]).items()}
        self.n_map_blocks = int(meta["n_map_blocks"])
        self.temb_total = self.W["temb_all.w"].shape[0]
        self.text_total, self.text_kp = self.W["text_k.w"].shape

    def _pack_layout(self) -> dict:
        """what decides which packed form each layer has (besides the weights themselves)"""
        nl = self.cfg.num_levels
        return dict(ws_levels=[bool(v) for v in self.ws_levels],
                    old_levels=[((self.h >> l) * (self.w >> l)) % 32 != 0 for l in range(nl)],
                    ws_skip=sorted(ops._WS_SKIP), ws_large=sorted(ops._WS_LARGE), ws_tokens=[self.N * (self.h >> l) * (self.w >> l) if self.ws_levels[l] else 0 for l in range(nl)],
                    cconv=os.environ.get("L2D_CCONV", "1"), rowgemm=os.environ.get("L2D_ROWGEMM", "1"), rowchain=os.environ.get("L2D_ROWCHAIN", "1"), rg_plain_max_k=os.environ.get("L2D_ROWGEMM_PLAIN_MAX_K", "640"),
                    rg_ff1_max_k=os.environ.get("L2D_ROWGEMM_FF1_MAX_K", "1280"),
                    # round 6: the fallback rules decide packed forms too (which layers take the weight-streaming form at token counts the
                    # tuner never saw; from how many blocks a level packs the chain kernel's weights)
                    ws_rule=os.environ.get("L2D_WSGEMM_RULE", "1"), rowchain_min_blocks=int(ops.ROWCHAIN_MIN_BLOCKS))

    @staticmethod
    def packed_cache_name(model_name: str, few_step_model_type: str, window_size: int, lora_dict: Optional[dict] = None,
                          height: int = 0, width: int = 0, denoising_steps_num: int = 0) -> str:
        """File stem for a packed-weight cache entry, in the spirit of the reference's engine prefix (wrapper.py:300-332).  Since
        round 4 the packed forms depend on the stream shape (which levels take the weight-streaming kernel), so the LATENT size
        and the number of denoising steps are part of the name like they are in the reference's prefix; the tiny-VAE is not."""
        stem = f"{model_name}--{few_step_model_type}--"
        for k, v in (lora_dict or {}).items():
            stem += f"{os.path.splitext(os.path.basename(str(k)))[0]}-{v}--"
        shape = f"{height}x{width}x{denoising_steps_num}--" if height and width and denoising_steps_num else ""
        return stem + shape + f"L{window_size}--l2dpack{HipStreamingUNet.PACK_FORMAT}"

    # ------------------------------------------------------------------ plan construction
    def _build_plan(self, mode: str, kv_cache: List[torch.Tensor]):
        cfg, dev, W = self.cfg, self.device, self.W
        B = self.N if mode == "stream" else self.F          # frames processed as the batch axis
        Bt = self.N if mode == "stream" else 1              # rows of timestep / text inputs
        h, w = self.h, self.w
        L, G = cfg.window_size, cfg.norm_num_groups
        ar = _Arena(dev)
        pl = _lib.OpList()
        # `cond_pl`: the launches that depend on (timestep, text) only -- time-embedding MLP + every resnet's
        # time_emb_proj, and the K / V^T text projections of all 16 cross-attention layers (frame-invariant).
        # They run when the conditioning changes (first frame, update_prompt, a new warm-up row), not every frame.
        cond_pl = _lib.OpList()
        st = SimpleNamespace(mode=mode, B=B, Bt=Bt, pl=pl, cond_pl=cond_pl, cond_key=None, arena=ar, tattn_ops=[], warm=False,
                             ident={}, rg=os.environ.get("L2D_ROWGEMM", "1") != "0", ws=any(self.ws_levels),
                             chain_heads=os.environ.get("L2D_ROWCHAIN_HEADS", "1") != "0")
        cur = [cond_pl]

        def add(opk):
            op, keep = opk
            cur[0].append(op, *keep)
            return op

        def gemm(x1, wt, out, **kw):
            """igemm with the (tile, split-K) schedule chosen for its shape; the fp32 split-K workspace comes from
            the arena and is released right after (stream order makes the reuse safe)."""
            batch, taps = kw.get("batch", 1), kw.get("taps", 1)
            epi = kw.get("epi", 0)
            tile, S, variant = ops.igemm_schedule(kw["M"], kw["Nout"], taps * kw["CinP"], batch, epi, taps)
            if self.igemm_splitk_off:
                S = 1
            if variant in (6, 7) and kw["CinP"] % 128:
                variant = 1            # BK = 128 rings need K slices of 128
            if tile == 1 and variant in (7, 8, 9):
                variant = 5            # deep rings exist for the 64x64 tile only (LDS)
            if epi == 1:
                S = 1                  # GEGLU pairs value and gate in one block's registers: no split-K
            ws, cnt_kw = None, {}
            if ops.splitk_fused(S):
                n_ws, n_cnt = ops.splitk_sizes(kw["M"], kw["Nout"], S, batch, tile)
                ws = ar.alloc(n_ws, torch.float32)
                cnt_kw = dict(cnt=st.sk_cnt, cnt_off=st.sk_used)
                st.sk_used += n_cnt
            elif S > 1:
                ws = ar.alloc(batch * S * kw["M"] * round_up(kw["Nout"], 4), torch.float32)
            # XCD tile order: weight-tile major when the weight matrix outweighs the activations (L2 fills, see igemm.hip)
            wbytes = kw["Nout"] * taps * kw["CinP"]
            xbytes = kw["M"] * (kw["C1"] + kw.get("C2", 0))
            order = int(wbytes > xbytes)
            op = add(ops.igemm(x1, wt, out, splitk=S, tile=tile, ws=ws, variant=variant, order=order, **cnt_kw, **kw))
            ar.release(ws)
            return op

        # arrival counters of the split-K launches (fused reduction): zero now, every launch leaves them zero
        st.sk_cnt, st.sk_used = torch.zeros(1 << 20, dtype=torch.int32, device=dev), 0

        # ---- static inputs
        st.in_sample = torch.zeros(B, cfg.in_channels, h * w, dtype=torch.float16, device=dev)
        st.in_depth = torch.zeros_like(st.in_sample)
        st.in_t = torch.zeros(Bt, dtype=torch.int64, device=dev)
        st.in_enc = torch.zeros(Bt, TEXT_PAD, self.text_kp, dtype=torch.float16, device=dev)
        if mode == "stream":
            st.in_bias = torch.zeros(B, L, dtype=torch.float16, device=dev)
            st.in_pe_idx = torch.zeros(B, L, dtype=torch.int64, device=dev)
            st.in_upd = torch.zeros(B, dtype=torch.int64, device=dev)
        st.out_sample = torch.zeros(B, cfg.out_channels, h * w, dtype=torch.float16, device=dev)

        # ---- helpers
        def new_act(C, H_, W_):
            return _Act(ar.alloc(B * H_ * W_ * C), C, H_, W_)

        def free(a: Optional[_Act]):
            if a is not None:
                ar.release(a.buf)

        def ident_affine(C):
            if C not in st.ident:
                st.ident[C] = (torch.ones(C, dtype=torch.float16, device=dev), torch.zeros(C, dtype=torch.float16, device=dev))
            return st.ident[C]

        def gn_stats_target(x: _Act, x2: Optional[_Act], T, cpg):
            """Ask the producers of x (and x2) to accumulate this GroupNorm's statistics; returns the accumulator pointer or
            None (then nothing is left attached)."""
            ins = [(x, 0)] + ([(x2, x.C)] if x2 is not None else [])
            if not (st.gn_fuse and all(a_.producer is not None for a_, _ in ins) and st.gn_layers < st.gn_acc.shape[0]):
                return None
            acc_ptr = st.gn_acc.data_ptr() + st.gn_layers * B * G * 2 * 8
            saved = [(a_.producer, [a_.producer.p[9], a_.producer.p[10]], list(a_.producer.i[24:30])) for a_, _ in ins]
            if all(ops.gn_target(a_.producer, acc_ptr, T=T, G=G, cpg=cpg, choff=off) for a_, off in ins):
                st.gn_layers += 1
                return acc_ptr
            for op_, ps, is_ in saved:                   # undo a half-attached layer
                op_.p[9], op_.p[10] = ps
                for j, v in enumerate(is_):
                    op_.i[24 + j] = v
            return None

        def gn(x: _Act, name, eps, silu, x2: Optional[_Act] = None, affine: bool = True) -> _Act:
            """affine=False: normalise only (gamma = 1, beta = 0): the consumer's packed weights carry the affine part."""
            T = x.H * x.W
            C2 = x2.C if x2 is not None else 0
            out = new_act(x.C + C2, x.H, x.W)
            gam, bet = (W[name + ".g"], W[name + ".beta"]) if affine else ident_affine(x.C + C2)
            # Statistics from the producers: the igemm launches that wrote x (and x2) accumulate sum / sum of squares per
            # (sample, group of THIS GroupNorm) in their epilogues as fixed-point integer atomics -- no gn_stats launch, no
            # second pass over the tensor.  Falls back to the stats kernel when a producer cannot (tile straddles samples,
            # both of its target slots taken, direct epilogue forced).
            cpg = (x.C + C2) // G
            acc_ptr = gn_stats_target(x, x2, T, cpg)
            if acc_ptr is not None:
                add(ops.gn_apply(x.buf, None, gam, bet, out.buf, eps=eps, silu=silu, B=B, T=T, C1=x.C,
                                 ld1=x.C, G=G, nchunk=0, x2=(x2.buf if x2 is not None else None), C2=C2, ld2=C2,
                                 acc_ptr=acc_ptr))
                return out
            if ops.gn_self_ok(T, x.C + C2, G):
                # small tensor whose producers cannot deliver the statistics (tokens per sample are no whole number of their tiles: any
                # resolution with 12 x 12, 6 x 6, 10 x 10 ... pixel levels): statistics + apply in ONE launch, the tensor read once
                st.gn_self_launches += 1
                add(ops.gn_apply(x.buf, None, gam, bet, out.buf, eps=eps, silu=silu, B=B, T=T, C1=x.C, ld1=x.C, G=G, nchunk=0,
                                 x2=(x2.buf if x2 is not None else None), C2=C2, ld2=C2))
                return out
            nchunk = max(1, min(64, T // 16))
            partial = ar.alloc(B * nchunk * G * 2, torch.float32)
            kw = dict(B=B, T=T, C1=x.C, ld1=x.C, G=G, nchunk=nchunk, x2=(x2.buf if x2 is not None else None), C2=C2,
                      ld2=C2)
            st.gn_stats_launches += 1
            add(ops.gn_stats(x.buf, partial, **kw))
            add(ops.gn_apply(x.buf, partial, gam, bet, out.buf, eps=eps, silu=silu, **kw))
            ar.release(partial)
            return out

        def conv3(x: _Act, name, stride=1, ups=0, epi=0, res: Optional[_Act] = None, rowbias=None, x2: Optional[_Act] = None,
                  gnf=None) -> _Act:
            """x2 / gnf = (acc_ptr, gamma, beta, eps): cconv only -- the conv of silu(GroupNorm(x | x2)), normalised inside the launch"""
            if (name + ".cw") in W:
                # patch-resident activations + register-streamed weights (cconv.hip): resnet convs of the wide levels, up-samplers
                assert stride == 1 and epi == 0
                cout = W[name + ".b"].numel()
                Ho, Wo = x.H << ups, x.W << ups
                out = new_act(cout, Ho, Wo)
                cin = x.C + (x2.C if x2 is not None else 0)
                kg = ops.cconv_schedule(self.N, Ho, Wo, cout, cin)[1]          # (the packing's: decided on the stream batch)
                sched = ops.cconv_schedule(B, Ho, Wo, cout, cin, KG=kg)
                ws_buf, kw = None, {}
                if sched[3] > 1:
                    n_ws, n_cnt = ops.cconv_sizes(B, Ho, Wo, cout, sched[0], sched[3])
                    ws_buf = ar.alloc(n_ws, torch.float32)
                    kw = dict(ws=ws_buf, cnt=st.sk_cnt, cnt_off=st.sk_used)
                    st.sk_used += n_cnt
                if rowbias is not None:
                    kw.update(rowbias=st.temb_all, ldrb=self.temb_total, rows_per_bias=(Ho * Wo if mode == "stream" else B * Ho * Wo))
                if gnf is not None:
                    kw.update(gn_acc_ptr=gnf[0], gn_gamma=gnf[1], gn_beta=gnf[2], gn_G=G, gn_eps=gnf[3])
                if x2 is not None:
                    kw.update(x2=x2.buf, C2=x2.C, ldx2=x2.C)
                op_ = add(ops.cconv(x.buf, W[name + ".cw"], out.buf, B=B, H=Ho, W=Wo, C1=x.C, ldx1=x.C, Nout=cout, ldo=cout, KG=kg, ups=ups,
                                    bias=W[name + ".b"], res=(res.buf if res is not None else None), ldr=(res.C if res is not None else 0),
                                    sched=sched, **kw))
                if rowbias is not None:
                    op_.p[4] = st.temb_all.data_ptr() + 4 * rowbias
                ar.release(ws_buf)
                out.producer = op_
                return out
            assert x2 is None and gnf is None
            if use_ws(name):
                # resnet conv at a few-token level: weight-streaming GEMM over (tap, channel chunk) stages (wsgemm.hip)
                assert stride == 1 and not ups and epi == 0
                cout = W[name + ".b"].numel()
                out = new_act(cout, x.H, x.W)
                out.producer = wslin(x.buf, B * x.H * x.W, x.C, name + ".ww", out.buf, cout, T=x.H * x.W, bias=W[name + ".b"],
                                     res=(res.buf if res is not None else None), ldr=(res.C if res is not None else 0), taps=9, Bc=B, H=x.H,
                                     Wd=x.W, rowbias_off=rowbias)
                return out
            wt = W[name + ".w"]
            cout = wt.shape[0]
            cinp = wt.shape[1] // 9
            Hin, Win = x.H, x.W
            if ups:
                Ho, Wo = Hin * 2, Win * 2
            elif stride == 2:
                Ho, Wo = (Hin - 1) // 2 + 1, (Win - 1) // 2 + 1
            else:
                Ho, Wo = Hin, Win
            out = new_act(cout, Ho, Wo)
            kw = {}
            if rowbias is not None:
                off = rowbias
                kw = dict(rowbias=st.temb_all, ldrb=self.temb_total, rows_per_bias=(Ho * Wo if mode == "stream" else B * Ho * Wo))
            patch = ops.pconv_patch(B, Hin, Win, cout, x.C) if (stride == 1 and not ups and epi == 0 and cinp == x.C) else None
            if patch is not None:
                # resnet convs at the resolutions where a CU's ingest, not the matrix cores, bounds the implicit-GEMM kernel:
                # activation patch resident in LDS, fetched once per 64-channel chunk instead of once per tap (pconv.hip)
                op_ = add(ops.pconv(x.buf, wt, out.buf, B=B, H=Hin, W=Win, C1=x.C, ldx1=x.C, CinP=cinp, Nout=cout, ldo=cout, patch=patch,
                                    bias=W[name + ".b"], res=(res.buf if res is not None else None), ldr=(res.C if res is not None else 0), **kw))
                if rowbias is not None:
                    op_.p[4] = st.temb_all.data_ptr() + 4 * rowbias
                out.producer = op_
                return out
            op_ = gemm(x.buf, wt, out.buf, M=B * Ho * Wo, Nout=cout, C1=x.C, ldx1=x.C, CinP=cinp, ldo=cout,
                            bias=W[name + ".b"], res=(res.buf if res is not None else None), ldr=(res.C if res is not None else 0),
                            taps=9, B=B, Hin=Hin, Win=Win, Hout=Ho, Wout=Wo, stride=stride, ups=ups, epi=epi, **kw)
            if rowbias is not None:
                op_.p[4] = st.temb_all.data_ptr() + 4 * rowbias
            out.producer = op_
            return out

        def linear_raw(xbuf, M, K, ldx, wt, outbuf, ldo, bias=None, res=None, ldr=0, epi=0, x2=None, C2=0, ldx2=0,
                       **kw):
            nout = wt.shape[0]
            return gemm(xbuf, wt, outbuf, M=M, Nout=nout, C1=K, ldx1=ldx, CinP=wt.shape[1], ldo=ldo, bias=bias,
                        res=res, ldr=ldr, epi=epi, x2=x2, C2=C2, ldx2=ldx2, **kw)

        def rowlin(xbuf, M, K, wkey, bkey, outbuf, ldo, ldx=None, res=None, ldr=0, **kw):
            """one token-row GEMM launch (rowgemm.hip) on weights packed by ops.pack_rowgemm"""
            wt = W[wkey]
            kw.setdefault("T", M // B)          # tokens per sample: 64-token tiles only when a sample is a whole number of them
            return add(ops.rowgemm(xbuf, wt, outbuf, M=M, K=K, Nout=wt.numel() // K, ldx=(ldx or K), ldo=ldo, bias=W.get(bkey),
                                   res=res, ldr=ldr, **kw))

        def wslin(xbuf, M, C1, wkey, outbuf, ldo, *, T, bias=None, colsum=None, res=None, ldr=0, x2buf=None, C2=0, epi=0, pro=0,
                  taps=1, Bc=1, H=1, Wd=1, rowbias_off=None, out_t=None, ntr=0, ldt=0, stt=0):
            """one weight-streaming GEMM launch (wsgemm.hip) on weights packed by ops.pack_wsgemm / pack_wsgemm_conv3x3; the split-K
            slabs come from the arena (released right after: stream order makes the reuse safe), the arrival counters from the
            plan's counter block"""
            wt = W[wkey]
            Ktot = taps * (C1 + C2)
            nout = wt.numel() // Ktot
            sched = ops.wsgemm_schedule(M, Ktot, nout, ntr, epi, pro, taps)
            NW_, NT_, NL_, S_, ntw_ = sched
            ws_buf, kw = None, {}
            if S_ > 1:
                n_ws, n_cnt = ops.wsgemm_sizes(M, nout, NW_, NT_, S_)
                ws_buf = ar.alloc(n_ws, torch.float32)
                kw = dict(ws=ws_buf, cnt=st.sk_cnt, cnt_off=st.sk_used)
                st.sk_used += n_cnt
            if rowbias_off is not None:
                kw.update(rowbias=st.temb_all, ldrb=self.temb_total, rows_per_bias=(T if mode == "stream" else B * T))
            op_ = add(ops.wsgemm(xbuf, wt, outbuf, M=M, Nout=nout, C1=C1, ldx1=C1, ldo=ldo, x2=x2buf, C2=C2, ldx2=C2, bias=bias, colsum=colsum,
                                 res=res, ldr=ldr, taps=taps, B=Bc, H=H, W=Wd, epi=epi, pro=pro, eps=1e-5, T=T, out_t=out_t, ntr=ntr, ldt=ldt,
                                 st=stt, sched=sched, **kw))
            if rowbias_off is not None:
                op_.p[4] = st.temb_all.data_ptr() + 4 * rowbias_off
            ar.release(ws_buf)
            return op_

        def use_ws(key) -> bool:
            return st.ws and (key + ".ww") in W

        def use_rg(key) -> bool:
            return st.rg and (key + ".rw") in W

        def linear(x: _Act, name, bias=True, res: Optional[_Act] = None, wkey=None, x2: Optional[_Act] = None, **kw) -> _Act:
            """kw: pro / eps / T / G / gn_acc_ptr of a fused norm prologue (row GEMM only)"""
            if wkey is None and use_ws(name) and not kw:
                nout = W[name + ".ww"].numel() // (x.C + (x2.C if x2 is not None else 0))
                out = new_act(nout, x.H, x.W)
                out.producer = wslin(x.buf, B * x.H * x.W, x.C, name + ".ww", out.buf, nout, T=x.H * x.W, bias=W.get(name + ".wb"),
                                     res=(res.buf if res is not None else None), ldr=(res.C if res is not None else 0),
                                     x2buf=(x2.buf if x2 is not None else None), C2=(x2.C if x2 is not None else 0))
                return out
            if x2 is None and wkey is None and use_rg(name):
                nout = W[name + ".rw"].numel() // x.C
                out = new_act(nout, x.H, x.W)
                out.producer = rowlin(x.buf, B * x.H * x.W, x.C, name + ".rw", name + ".rb", out.buf, nout,
                                      res=(res.buf if res is not None else None), ldr=(res.C if res is not None else 0), **kw)
                return out
            assert not kw
            wt = W[wkey or (name + ".w")]
            out = new_act(wt.shape[0], x.H, x.W)
            out.producer = linear_raw(x.buf, B * x.H * x.W, x.C, x.C, wt, out.buf, wt.shape[0], bias=(W[name + ".b"] if bias else None),
                                      res=(res.buf if res is not None else None), ldr=(res.C if res is not None else 0),
                                      x2=(x2.buf if x2 is not None else None), C2=(x2.C if x2 is not None else 0),
                                      ldx2=(x2.C if x2 is not None else 0))
            return out

        def layernorm(x: _Act, name) -> _Act:
            out = new_act(x.C, x.H, x.W)
            add(ops.layernorm(x.buf, W[name + ".g"], W[name + ".beta"], out.buf, rows=B * x.H * x.W, C=x.C, ldx=x.C, ldo=x.C))
            return out

        def gn_linear(x: _Act, nname, eps, lname) -> _Act:
            """GroupNorm -> Linear.  Row GEMM path: the normalisation is the GEMM's prologue (statistics from x's producers),
            the affine part lives in the packed weights; if the statistics cannot come from the producers or a sample is not a
            whole number of 32-token tiles, a normalise-only GroupNorm launch runs in front."""
            if not use_rg(lname):
                hn = gn(x, nname, eps, False)
                y = linear(hn, lname)
                free(hn)
                return y
            T = x.H * x.W
            acc_ptr = gn_stats_target(x, None, T, x.C // G) if T % 32 == 0 else None
            if acc_ptr is not None:
                return linear(x, lname, pro=2, eps=eps, T=T, G=G, gn_acc_ptr=acc_ptr)
            hn = gn(x, nname, eps, False, affine=False)
            y = linear(hn, lname)
            free(hn)
            return y

        def ln_rowlin(x: _Act, wkey, outbuf, ldo, **kw):
            """LayerNorm -> Linear as one row GEMM launch (LayerNorm eps = 1e-5: nn.LayerNorm default, as ops.layernorm)"""
            return rowlin(x.buf, B * x.H * x.W, x.C, wkey + ".rw", wkey + ".rb", outbuf, ldo, pro=1, eps=1e-5, **kw)

        def geglu_ff(x: _Act, name, res: _Act, nname=None) -> _Act:
            """x: the un-normalised input when `nname` names the LayerNorm to fuse (row GEMM), else the normalised one"""
            if nname is not None and st.ws and (name + ".ww1") in W:
                c4 = W[name + ".ww1"].numel() // x.C // 2
                hid = new_act(c4, x.H, x.W)
                wslin(x.buf, B * x.H * x.W, x.C, name + ".ww1", hid.buf, c4, T=x.H * x.W, bias=W[name + ".wb1"], colsum=W[name + ".wcs1"], epi=1, pro=1)
                out = linear(hid, name + ".net.2", res=res)
                free(hid)
                return out
            if nname is not None:
                c4 = W[name + ".rw1"].numel() // x.C // 2
                hid = new_act(c4, x.H, x.W)
                rowlin(x.buf, B * x.H * x.W, x.C, name + ".rw1", name + ".rb1", hid.buf, c4, pro=1, eps=1e-5, epi=1)
                out = linear(hid, name + ".net.2", res=res)
                free(hid)
                return out
            return geglu_ff_old(x, name, res)

        def geglu_ff_old(x: _Act, name, res: _Act) -> _Act:
            w1 = W[name + ".w1"]
            c4 = w1.shape[0] // 2
            hid = new_act(c4, x.H, x.W)
            linear_raw(x.buf, B * x.H * x.W, x.C, x.C, w1, hid.buf, c4, bias=W[name + ".b1"], epi=1)
            out = linear(hid, name + ".net.2", res=res, **({} if (use_rg(name + ".net.2") or use_ws(name + ".net.2")) else dict(wkey=name + ".net.2.w")))
            free(hid)
            return out

        def block_tail(ao: _Act, res1: _Act, res2: _Act, to_out, ff, proj_out) -> Optional[_Act]:
            """attention output projection + residual -> LayerNorm -> GEGLU -> FF2 + residual -> proj_out + block residual as ONE
            token-resident launch (rowchain.hip) where the level's M / 32 blocks fill the chip (C = 320); None = not here."""
            T, C = ao.H * ao.W, ao.C
            keys = (to_out + ".rw", to_out + ".rb", ff + ".rw1", ff + ".rb1", ff + ".net.2.chw", ff + ".net.2.chb", proj_out + ".rw", proj_out + ".rb")
            if not (st.rg and ops.rowchain_ok(B * T, C, T) and all(k in W for k in keys)):
                return None
            out = new_act(C, ao.H, ao.W)
            out.producer = add(ops.rowchain(ao.buf, res1.buf, res2.buf, out.buf, M=B * T, C=C, w_out=W[keys[0]], b_out=W[keys[1]],
                                            w_ff1=W[keys[2]], b_ff1=W[keys[3]], w_ff2=W[keys[4]], b_ff2=W[keys[5]], w_po=W[keys[6]],
                                            b_po=W[keys[7]], eps=1e-5))
            return out

        def block_head(x: _Act, a_name, b_name, outbuf, passes, *, res: Optional[_Act] = None, gn_of: Optional[_Act] = None, out_t=None,
                       ldt=0, stt=0, ldo=None) -> Optional[_Act]:
            """Two dependent layers as one token-resident launch (rowchain.hip head segment): h = A(x) (+ res) -- or A(GroupNorm(x)) with
            the statistics from x's producers -- stored as the residual stream, then B(LayerNorm(h)) -> outbuf (q | k | v, q | k + V^T,
            or the cross-attention's query).  Returns h, or None when the segment does not run here (the caller emits the two launches)."""
            T, C = x.H * x.W, x.C
            keys = (a_name + ".rw", a_name + ".rb", b_name + ".rw")
            if not (st.rg and st.chain_heads and ops.rowchain_ok(B * T, C, T) and all(k in W for k in keys)):
                return None
            acc_ptr = None
            if gn_of is not None:
                acc_ptr = gn_stats_target(gn_of, None, T, C // G)
                if acc_ptr is None:
                    return None
            h = new_act(C, x.H, x.W)
            add(ops.rowchain_head(x.buf, h.buf, outbuf, M=B * T, C=C, wA=W[keys[0]], bA=W[keys[1]], wB=W[keys[2]], bB=W.get(b_name + ".rb"),
                                  passes=passes, resA=(res.buf if res is not None else None), gn_acc_ptr=acc_ptr, T=T, G=G,
                                  eps_gn=cfg.transformer_norm_eps, eps_ln=1e-5, out_t=out_t, ldt=ldt, st=stt, ldo=ldo))
            return h

        # L2D_CCONV_GN=1: the GroupNorm + SiLU in front of a cconv launch runs inside it (its loader waves normalise the patch in LDS).  Built,
        # parity-tested (tests/test_gpu_cconv.py) and measured in the frame (round 6, profiles/round6_f_cconv_gn_fused_ab.txt): 20 GroupNorm
        # launches and their 0.135 ms go, but every output-channel tile of a conv re-normalises its patch (10-20 x redundant arithmetic at
        # the 640 / 1280-wide levels) and the cconv family pays 0.16 ms for it: 7.96-8.01 vs 7.99-8.00 ms per frame.  Off by default.
        gn_in_conv = os.environ.get("L2D_CCONV_GN", "0") != "0"

        def gn_conv3(x: _Act, x2: Optional[_Act], nname, cname, **kw) -> _Act:
            """conv3(silu(GroupNorm(x | x2))) (reference resnet.py:233-234, 249-250).  Where the conv is a cconv launch and the statistics
            come from the producers' epilogues, the normalisation runs inside that launch (its loader waves normalise the patch in LDS):
            no GroupNorm launch, no normalised tensor; else GroupNorm launch + conv."""
            if gn_in_conv and (cname + ".cw") in W:
                C_ = x.C + (x2.C if x2 is not None else 0)
                if C_ % G == 0 and -(-(C_ // 64) // 1) <= 8 * 48:
                    acc_ptr = gn_stats_target(x, x2, x.H * x.W, C_ // G)
                    if acc_ptr is not None:
                        return conv3(x, cname, x2=x2, gnf=(acc_ptr, W[nname + ".g"], W[nname + ".beta"], cfg.norm_eps), **kw)
            hn = gn(x, nname, cfg.norm_eps, True, x2=x2)
            out_ = conv3(hn, cname, **kw)
            free(hn)
            return out_

        def resnet(x: _Act, name, skip: Optional[_Act] = None) -> _Act:
            h1 = gn_conv3(x, skip, name + ".norm1", name + ".conv1", rowbias=self.temb_offsets[name])
            if (name + ".conv_shortcut.w") in W or (name + ".conv_shortcut.ww") in W:
                sc = linear(x, name + ".conv_shortcut", x2=skip)
                out = gn_conv3(h1, None, name + ".norm2", name + ".conv2", res=sc)
                free(sc)
            else:
                assert skip is None
                out = gn_conv3(h1, None, name + ".norm2", name + ".conv2", res=x)
            free(h1)
            return out

        def spatial(x: _Act, name) -> _Act:
            T, C = x.H * x.W, x.C
            d = C // cfg.num_heads
            b = name + ".transformer_blocks.0"
            rg = use_rg(b + ".attn1.qkv") and T % 32 == 0         # (else: the implicit-GEMM path with separate norm launches)
            rg_in = use_rg(name + ".proj_in") and T % 32 == 0
            ldvt = round_up(T, 8)
            # every linear layer picks its kernel by the packed form it finds: weight-streaming (.ww, few-token levels), token-row
            # (.rw) or implicit GEMM (.w)
            lin = lambda a_, nm, **k_: linear(a_, nm, **k_) if (use_ws(nm) or use_rg(nm)) else linear(a_, nm, wkey=nm + ".w", **k_)
            y = None
            qk = vt = None
            if rg_in and rg and T % 32 == 0:
                # proj_in behind the block's GroupNorm + norm1 -> q | k | V^T as one launch (head segment of rowchain.hip)
                qk, vt = ar.alloc(B * T * 2 * C), ar.alloc(B * C * ldvt)
                y = block_head(x, name + ".proj_in", b + ".attn1.qkv", qk, 3, gn_of=x, out_t=vt, ldt=ldvt, stt=C * ldvt, ldo=2 * C)
                if y is None:
                    ar.release(qk); ar.release(vt)
                    qk = vt = None
            if y is not None:
                pass
            elif rg_in:
                y = gn_linear(x, name + ".norm", cfg.transformer_norm_eps, name + ".proj_in")
            else:
                if (name + ".proj_in.w") not in W:
                    raise ValueError(f"{name}: T = {T} tokens per sample is no multiple of 32 at this level and the packed weights "
                                     "lack the implicit-GEMM form of this block (packed-weight file written at another "
                                     "resolution): re-pack from the state dict at this resolution")
                hn = gn(x, name + ".norm", cfg.transformer_norm_eps, False)
                y = linear(hn, name + ".proj_in", wkey=name + ".proj_in.w")
                free(hn)
            # --- self attention: norm1 -> q | k | V^T
            fused_qkv = qk is not None
            if not fused_qkv:
                qk = ar.alloc(B * T * 2 * C)
                vt = ar.alloc(B * C * ldvt)
            if fused_qkv:
                pass
            elif use_ws(b + ".attn1.qkv"):
                wslin(y.buf, B * T, C, b + ".attn1.qkv.ww", qk, 2 * C, T=T, bias=W.get(b + ".attn1.qkv.wb"), colsum=W[b + ".attn1.qkv.wcs"], pro=1,
                      out_t=vt, ntr=C, ldt=ldvt, stt=C * ldvt)
            elif rg:
                ln_rowlin(y, b + ".attn1.qkv", qk, 2 * C, T=T, out_t=vt, ntr=C, ldt=ldvt, st=C * ldvt)
            else:
                n1 = layernorm(y, b + ".norm1")
                linear_raw(n1.buf, B * T, C, C, W[b + ".attn1.qk"], qk, 2 * C)
                # V^T[b] = Wv . n1[b]^T : the same GEMM with operand roles swapped (tokens act as "channels")
                wv = W[b + ".attn1.v"]
                gemm(wv, n1.buf, vt, M=C, Nout=T, C1=C, ldx1=wv.shape[1], CinP=C, ldo=ldvt, batch=B, sx1=0,
                              sw=T * C, so=C * ldvt)
                free(n1)
            ao = new_act(C, x.H, x.W)
            add(ops.flash_attn(qk, qk, vt, ao.buf, B=B, H=cfg.num_heads, d=d, Tq=T, Tk=T, ldq=2 * C, ldk=2 * C, ldvt=ldvt,
                               ldo=C, sq=T * 2 * C, sk=T * 2 * C, svt=C * ldvt, so=T * C, k_off=C))
            ar.release(qk); ar.release(vt)
            # --- attn1.to_out + residual -> norm2 -> cross-attention query: one launch where the head segment runs
            q2 = new_act(C, x.H, x.W)
            y2 = block_head(ao, b + ".attn1.to_out.0", b + ".attn2.to_q", q2.buf, 1, res=y)
            fused_q = y2 is not None
            if not fused_q:
                y2 = lin(ao, b + ".attn1.to_out.0", res=y)
            free(ao); free(y)
            # --- text cross attention (K / V^T of all 16 layers come from two batched GEMMs at plan start)
            if fused_q:
                pass
            elif use_ws(b + ".attn2.to_q"):
                wslin(y2.buf, B * T, C, b + ".attn2.to_q.ww", q2.buf, C, T=T, bias=W.get(b + ".attn2.to_q.wb"), colsum=W[b + ".attn2.to_q.wcs"], pro=1)
            elif use_rg(b + ".attn2.to_q") and T % 32 == 0:
                ln_rowlin(y2, b + ".attn2.to_q", q2.buf, C)
            else:
                free(q2)
                n2 = layernorm(y2, b + ".norm2")
                q2 = linear(n2, b + ".attn2.to_q", bias=False, wkey=b + ".attn2.to_q.w")
                free(n2)
            off = self.text_offsets[name]
            ao = new_act(C, x.H, x.W)
            add(ops.flash_attn(q2.buf, st.text_k, st.text_vt, ao.buf, B=B, H=cfg.num_heads, d=d, Tq=T, Tk=st.text_len,
                               ldq=C, ldk=self.text_total, ldvt=TEXT_PAD, ldo=C, sq=T * C,
                               sk=(TEXT_PAD * self.text_total if Bt > 1 else 0),
                               svt=(self.text_total * TEXT_PAD if Bt > 1 else 0), so=T * C, k_off=off, vt_off=off * TEXT_PAD))
            free(q2)
            tail = block_tail(ao, y2, x, b + ".attn2.to_out.0", b + ".ff", name + ".proj_out")
            if tail is not None:
                free(ao); free(y2)
                return tail
            y3 = lin(ao, b + ".attn2.to_out.0", res=y2)
            free(ao); free(y2)
            if (st.ws and (b + ".ff.ww1") in W) or (st.rg and T % 32 == 0 and (b + ".ff.rw1") in W):
                y4 = geglu_ff(y3, b + ".ff", res=y3, nname=b + ".norm3")
            else:
                n3 = layernorm(y3, b + ".norm3")
                y4 = geglu_ff_old(n3, b + ".ff", res=y3)
                free(n3)
            free(y3)
            out = lin(y4, name + ".proj_out", res=x)
            free(y4)
            return out

        def motion(x: _Act, name, idx_base: int) -> _Act:
            T, C = x.H * x.W, x.C
            t = name + ".temporal_transformer"
            b = t + ".transformer_blocks.0"
            # proj_in behind the module's GroupNorm + LayerNorm -> q | k | v of the first attention as one launch where the head segment runs
            qkv_next = ar.alloc(B * T * 3 * C)
            y = block_head(x, t + ".proj_in", b + ".attention_blocks.0.qkv", qkv_next, 3, gn_of=x)
            if y is None:
                ar.release(qkv_next)
                qkv_next = None
                y = gn_linear(x, t + ".norm", cfg.transformer_norm_eps, t + ".proj_in")
            for j in range(2):
                a = b + f".attention_blocks.{j}"
                qkv = qkv_next if qkv_next is not None else ar.alloc(B * T * 3 * C)
                if qkv_next is not None:
                    qkv_next = None
                elif use_ws(a + ".qkv"):
                    wslin(y.buf, B * T, C, a + ".qkv.ww", qkv, 3 * C, T=T, bias=W.get(a + ".qkv.wb"), colsum=W[a + ".qkv.wcs"], pro=1)
                elif use_rg(a + ".qkv"):
                    ln_rowlin(y, a + ".qkv", qkv, 3 * C)
                else:
                    nrm = layernorm(y, b + f".norms.{j}")
                    linear_raw(nrm.buf, B * T, C, C, W[a + ".qkv"], qkv, 3 * C)
                    free(nrm)
                ao = new_act(C, x.H, x.W)
                idx = idx_base + j
                cache = kv_cache[idx]
                if mode == "stream":
                    op = add(ops.tattn_stream(qkv, cache, W[a + ".q_pe"], W[a + ".k_pe"], W[a + ".v_pe"], st.in_pe_idx,
                                              st.in_upd, st.in_bias, ao.buf, N=B, T=T, C=C, L=L, H=cfg.temporal_heads,
                                              variant=self.tattn_variant))
                else:
                    op = add(ops.tattn_warmup(qkv, cache[0], W[a + ".q_pe"], W[a + ".k_pe"], W[a + ".v_pe"], ao.buf,
                                              F=B, T=T, C=C, L=L, H=cfg.temporal_heads))
                st.tattn_ops.append((op.tag, idx))
                ar.release(qkv)
                if j == 1:
                    tail = block_tail(ao, y, x, a + ".to_out.0", b + ".ff", t + ".proj_out")
                    if tail is not None:
                        free(ao); free(y)
                        return tail
                y2 = None
                if j == 0:
                    # to_out + residual -> LayerNorm -> q | k | v of the second attention
                    qkv_next = ar.alloc(B * T * 3 * C)
                    y2 = block_head(ao, a + ".to_out.0", b + ".attention_blocks.1.qkv", qkv_next, 3, res=y)
                    if y2 is None:
                        ar.release(qkv_next)
                        qkv_next = None
                if y2 is None:
                    y2 = linear(ao, a + ".to_out.0", res=y)
                free(ao); free(y)
                y = y2
            if (st.ws and (b + ".ff.ww1") in W) or use_rg(b + ".ff") or (st.rg and (b + ".ff.rw1") in W):
                y2 = geglu_ff(y, b + ".ff", res=y, nname=b + ".ff_norm")
            else:
                nrm = layernorm(y, b + ".ff_norm")
                y2 = geglu_ff_old(nrm, b + ".ff", res=y)
                free(nrm)
            free(y)
            out = linear(y2, t + ".proj_out", res=x)
            free(y2)
            return out

        # ---- time embedding: sinusoid -> MLP -> SiLU -> every resnet's time_emb_proj in ONE skinny GEMM
        c0, E = cfg.block_out_channels[0], cfg.time_embed_dim
        t_sin = torch.zeros(Bt, c0, dtype=torch.float16, device=dev)
        t_h1 = torch.zeros(Bt, E, dtype=torch.float16, device=dev)
        t_h2 = torch.zeros(Bt, E, dtype=torch.float16, device=dev)
        st.temb_all = torch.zeros(Bt, self.temb_total, dtype=torch.float32, device=dev)
        add(ops.timestep_embed(st.in_t, t_sin, N=Bt, dim=c0))
        add(ops.skinny_linear(t_sin, W["time_embedding.linear_1.w"], W["time_embedding.linear_1.b"], t_h1, M=Bt, K=c0,
                              Nout=E, silu_out=True))
        add(ops.skinny_linear(t_h1, W["time_embedding.linear_2.w"], W["time_embedding.linear_2.b"], t_h2, M=Bt, K=E,
                              Nout=E, silu_out=True))   # only silu(emb) is ever consumed (resnet.py:238)
        add(ops.skinny_linear(t_h2, W["temb_all.w"], W["temb_all.b"], st.temb_all, M=Bt, K=E, Nout=self.temb_total))

        # ---- text K / V^T for all cross-attention layers (two GEMMs)
        st.text_len = self.text_len
        st.text_k = torch.zeros(Bt * TEXT_PAD, self.text_total, dtype=torch.float16, device=dev)
        st.text_vt = torch.zeros(Bt, self.text_total, TEXT_PAD, dtype=torch.float16, device=dev)
        D = cfg.cross_attention_dim
        gemm(st.in_enc, W["text_k.w"], st.text_k, M=Bt * TEXT_PAD, Nout=self.text_total, C1=D, ldx1=self.text_kp,
                      CinP=self.text_kp, ldo=self.text_total)
        gemm(W["text_v.w"], st.in_enc, st.text_vt, M=self.text_total, Nout=TEXT_PAD, C1=D, ldx1=self.text_kp,
                      CinP=self.text_kp, ldo=TEXT_PAD, batch=Bt, sx1=0, sw=TEXT_PAD * self.text_kp,
                      so=self.text_total * TEXT_PAD)

        cur[0] = pl
        # ---- GroupNorm statistics accumulators (one [B][G][2] int64 block per fused GroupNorm), zeroed once per frame
        st.gn_fuse = os.environ.get("L2D_GN_FUSE", "1") != "0"
        st.gn_layers, st.gn_stats_launches, st.gn_self_launches = 0, 0, 0
        st.gn_acc = torch.zeros(96, B, G, 2, dtype=torch.int64, device=dev)
        st.gn_zero = torch.zeros_like(st.gn_acc)
        zero_op = add(ops.copy(st.gn_zero, st.gn_acc, st.gn_acc.numel() * 8)) if st.gn_fuse else None
        # ---- input: NCHW latents -> channels-last (padded to 8 channels), conv_in + depth mapping network
        x_in = _Act(ar.alloc(B * h * w * 8), 8, h, w)
        d_in = _Act(ar.alloc(B * h * w * 8), 8, h, w)
        add(ops.nchw_to_nhwc(st.in_sample, x_in.buf, B=B, C=cfg.in_channels, HW=h * w, Cpad=8))
        add(ops.nchw_to_nhwc(st.in_depth, d_in.buf, B=B, C=cfg.in_channels, HW=h * w, Cpad=8))
        x0 = conv3(x_in, "conv_in")
        e = conv3(d_in, "flow_conv_in.conv_in", epi=2)
        for i in range(self.n_map_blocks):
            e2 = conv3(e, f"flow_conv_in.blocks.{i}", epi=2)
            free(e)
            e = e2
        x = conv3(e, "flow_conv_in.conv_out", res=x0)     # depth embedding + conv_in(sample) (:523-526)
        free(e); free(x0); free(x_in); free(d_in)

        skips = [x]
        mm = 0
        nl = cfg.num_levels
        for i in range(nl):
            for j in range(cfg.layers_per_block):
                x2 = resnet(x, f"down_blocks.{i}.resnets.{j}")
                if x is not skips[-1]:
                    free(x)
                x = x2
                if i != nl - 1:
                    x2 = spatial(x, f"down_blocks.{i}.attentions.{j}")
                    free(x)
                    x = x2
                x2 = motion(x, f"down_blocks.{i}.motion_modules.{j}", mm)
                free(x)
                x = x2
                mm += 2
                skips.append(x)
            if i != nl - 1:
                x = conv3(x, f"down_blocks.{i}.downsamplers.0.conv", stride=2)
                skips.append(x)
        x2 = resnet(x, "mid_block.resnets.0")          # x is still referenced by skips[-1]
        x = x2
        x2 = spatial(x, "mid_block.attentions.0"); free(x); x = x2
        x2 = resnet(x, "mid_block.resnets.1"); free(x); x = x2
        for i in range(nl):
            for j in range(cfg.layers_per_block + 1):
                sk = skips.pop()
                x2 = resnet(x, f"up_blocks.{i}.resnets.{j}", skip=sk)
                free(x); free(sk)
                x = x2
                if i != 0:
                    x2 = spatial(x, f"up_blocks.{i}.attentions.{j}"); free(x); x = x2
                x2 = motion(x, f"up_blocks.{i}.motion_modules.{j}", mm); free(x); x = x2
                mm += 2
            if i != nl - 1:
                x2 = conv3(x, f"up_blocks.{i}.upsamplers.0.conv", ups=1); free(x); x = x2
        hn = gn(x, "conv_norm_out", cfg.norm_eps, True)
        free(x)
        y = conv3(hn, "conv_out")
        add(ops.nhwc_to_nchw(y.buf, st.out_sample, B=B, C=cfg.out_channels, HW=h * w, ld=cfg.out_channels))
        if zero_op is not None:
            zero_op.l[0] = max(16, st.gn_layers * B * G * 2 * 8)          # only the blocks in use
        st.kv_ptrs = [c.data_ptr() for c in kv_cache]
        st.arena_bytes = ar.nbytes()
        st.n_ops = len(pl)
        return st

    def _plan(self, mode, kv_cache):
        st = self._plans.get(mode)
        if st is None:
            st = self._build_plan(mode, kv_cache)
            self._plans[mode] = st
        return st

    def _bind_caches(self, st, kv_cache, row: Optional[int] = None):
        """Re-point the temporal-attention ops at the caller's cache tensors (they normally never change:
        the pipeline owns one `kv_cache_list` for the stream's lifetime)."""
        changed = False
        for tag, idx in st.tattn_ops:
            c = kv_cache[idx]
            assert c.dtype == torch.float16 and c.is_contiguous(), "kv_cache must be contiguous fp16 [N,2,T,L,C]"
            ptr = c.data_ptr() if row is None else c.data_ptr() + row * c.stride(0) * 2
            op = st.pl[tag]
            if op.p[1] != ptr:
                op.p[1] = ptr
                changed = True
        if changed:
            st.pl._arr = None
            self._graph.pop(st.mode, None)

    def invalidate_text_cache(self):
        """Force the conditioning launches (time embedding, text K / V^T) to re-run on the next call: for callers that
        rewrite the plan's static `in_enc` / `in_t` buffers themselves (HipStreamStep.set_prompt)."""
        for st in self._plans.values():
            st.cond_key = None

    def _load_cond(self, st, timestep, encoder_hidden_states):
        """Conditioning inputs -> static buffers + the `cond_pl` launches, only when they changed.  "Changed" is decided
        on the host without a sync: the SAME tensor objects as last call (held here, so their storage cannot be recycled
        for other data) with unchanged in-place version counters.  The reference pipeline passes `self.prompt_embeds`
        (re-bound only by update_prompt) and one `sub_timesteps_tensor` for the whole stream; a caller that builds fresh
        tensors every call simply gets the launches every call."""
        cfg = self.cfg
        key = (timestep, encoder_hidden_states, timestep._version, encoder_hidden_states._version)
        old = st.cond_key
        if (self.cond_cache and old is not None and len(old) == 4 and old[0] is key[0] and old[1] is key[1]
                and old[2:] == key[2:]):
            return
        st.in_t.copy_(timestep.reshape(-1)[:1] if st.Bt == 1 else timestep.reshape(-1).expand(st.Bt))
        st.in_enc[:, : st.text_len, : cfg.cross_attention_dim].copy_(encoder_hidden_states[: st.Bt])
        st.cond_pl.run()
        st.cond_key = key

    def _ensure_cond(self, st):
        """For callers that own the static inputs (HipStreamStep): run the conditioning launches if they are stale."""
        if st.cond_key is None:
            st.cond_pl.run()
            st.cond_key = ("external",)

    def _run(self, st):
        if self.use_graph and st.warm:      # the first call runs directly: the launchers' one-time kernel-attribute /
            g = self._graph.get(st.mode)    # device queries are not allowed inside a stream capture
            if g is None:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    g = _lib.Graph(st.pl, stream=int(side.cuda_stream))
                torch.cuda.current_stream().wait_stream(side)
                self._graph[st.mode] = g
            g.launch()
        else:
            st.pl.run()
            st.warm = True

    # ------------------------------------------------------------------ the boundary call
    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states=None, temporal_attention_mask=None, depth_sample=None,
                 kv_cache=None, pe_idx=None, update_idx=None, return_dict: bool = True, **kwargs):
        N, cfg = self.N, self.cfg
        if tuple(sample.shape) != (N, cfg.in_channels, 1, self.h, self.w):
            raise ValueError(f"sample shape {tuple(sample.shape)} != static {(N, cfg.in_channels, 1, self.h, self.w)}")
        if kv_cache is None or len(kv_cache) != len(self.mm_layout):
            raise ValueError(f"kv_cache must be the list of {len(self.mm_layout)} caches from prepare_cache()")
        st = self._plan("stream", kv_cache)
        self._bind_caches(st, kv_cache)
        st.text_len_rt = encoder_hidden_states.shape[1]
        if st.text_len_rt != st.text_len:
            raise ValueError(f"text length {st.text_len_rt} != static {st.text_len}")
        st.in_sample.copy_(sample.reshape(N, cfg.in_channels, -1))
        st.in_depth.copy_(depth_sample.reshape(N, cfg.in_channels, -1))
        self._load_cond(st, timestep, encoder_hidden_states)
        st.in_bias.copy_(temporal_attention_mask)
        st.in_pe_idx.copy_(pe_idx)
        st.in_upd.copy_(update_idx)
        self._run(st)
        if self.fresh_output:
            out = st.out_sample.clone().view(N, cfg.out_channels, 1, self.h, self.w)
            return UNetOutput(out, kv_cache) if return_dict else (out, kv_cache)
        out = st.out_sample.view(N, cfg.out_channels, 1, self.h, self.w)    # a view of the plan's static output buffer (like
        if not return_dict:                                                  # the TensorRT engine's output binding): the next call overwrites it
            return (out, kv_cache)
        return UNetOutput(out, kv_cache)

    @torch.no_grad()
    def warmup(self, sample, timestep, encoder_hidden_states=None, depth_sample=None, kv_cache=None, row: int = 0,
               return_dict: bool = True, **kwargs):
        """Warm-up UNet pass over F frames that fills cache row `row` (slots 0..F-1) of every layer.
        sample/depth [1,4,F,h,w]; timestep [1]; encoder_hidden_states [1,77,D]; kv_cache = the FULL cache list
        (the reference passes `[cache[idx] for cache in kv_cache_list]`, pipeline :326)."""
        F_, cfg = self.F, self.cfg
        if tuple(sample.shape) != (1, cfg.in_channels, F_, self.h, self.w):
            raise ValueError(f"warm-up sample shape {tuple(sample.shape)} != {(1, cfg.in_channels, F_, self.h, self.w)}")
        st = self._plan("warmup", kv_cache)
        self._bind_caches(st, kv_cache, row=row)
        st.in_sample.copy_(sample[0].transpose(0, 1).reshape(F_, cfg.in_channels, -1))
        st.in_depth.copy_(depth_sample[0].transpose(0, 1).reshape(F_, cfg.in_channels, -1))
        self._load_cond(st, timestep, encoder_hidden_states)
        self._run(st)
        out = st.out_sample.view(F_, cfg.out_channels, self.h, self.w).transpose(0, 1).unsqueeze(0)
        if not return_dict:
            return (out,)
        return UNetOutput(out, kv_cache)

    # ------------------------------------------------------------------ introspection for bench / tests
    def plan_summary(self, mode="stream"):
        st = self._plans[mode]
        kinds = {}
        for j in range(len(st.pl)):
            kinds[st.pl[j].kind] = kinds.get(st.pl[j].kind, 0) + 1
        return dict(n_ops=len(st.pl), n_cond_ops=len(st.cond_pl), gn_fused=st.gn_layers, gn_stats_launches=st.gn_stats_launches, gn_self_launches=st.gn_self_launches, kinds=kinds, arena_bytes=st.arena_bytes, weight_bytes=self.weight_bytes())
