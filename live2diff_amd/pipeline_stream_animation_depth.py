"""Host-side mirror of the reference's per-frame streaming loop, built around the HIP UNet backend.

Mirrors `live2diff/pipeline_stream_animation_depth.py:24-666` (class, attribute and method names are the
reference's public surface) but is organised around device-resident state:

  * the ring-buffer state machine (`initialize_attn_bias_pe_and_update_idx` / `update_attn_bias`, reference
    :403-438) runs on small HOST tensors and is uploaded once per frame -- the reference evaluates
    `.any()` / `.sum()` on device tensors, i.e. N host syncs per frame;
  * `WARMUP_FRAMES` / `WINDOW_SIZE` are constructor parameters (the reference hard-codes 8 / 16 at :20-21,
    BASELINE configs need 4/8 sink and 8/16/32 rolling slots);
  * N = 1 is defined (the reference raises IndexError at :412): row-0 rule only;
  * `stream.unet` is a `HipStreamingUNet`; the warm-up UNet is the same object's `.warmup` (the reference
    keeps a second CPU-resident copy of all weights and moves it to the GPU for `prepare`, :315,338).

VAE / text encoder / depth detector are the caller's objects (duck-typed exactly like the reference's
`stream.vae`, `stream.text_encoder`, `stream.depth_detector` swap points); they are outside this path.  The
near-duplicate frame gate (reference image_filter.py) defaults to this package's `frame_filter.SimilarImageFilter`;
`stream.similar_filter` may be replaced by any object with `__call__(x) -> x | None`, `set_threshold`,
`set_max_skip_frame`.
"""
import time
from typing import List, Literal, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from .frame_filter import SimilarImageFilter
from .scheduler import LCMSchedule

WARMUP_FRAMES = 8
WINDOW_SIZE = 16


# ----------------------------------------------------------------------------- ring buffer (host tensors)
def ring_buffer_init(n: int, window: int = WINDOW_SIZE, sink: int = WARMUP_FRAMES):
    """reference :403-414.  Returns (attn_bias [n,L] fp32 0/-inf, pe_idx [n,L] int64, update_idx [n] int64)."""
    mask = torch.zeros(n, window, dtype=torch.bool)
    mask[:, :sink] = True
    mask[0, sink] = True
    bias = torch.zeros(n, window, dtype=torch.float32)
    bias.masked_fill_(~mask, float("-inf"))
    pe_idx = torch.arange(window, dtype=torch.int64).unsqueeze(0).repeat(n, 1)
    update_idx = torch.full((n,), sink, dtype=torch.int64)
    if n > 1:
        update_idx[1] = sink + 1          # reference quirk kept: slot sink+1 is still masked for row 1 (:412)
    return bias, pe_idx, update_idx


def ring_buffer_update(bias, pe_idx, update_idx, window: int = WINDOW_SIZE, sink: int = WARMUP_FRAMES):
    """reference :416-438 (in place on host tensors, returned for convenience)."""
    n = bias.shape[0]
    for i in range(n):
        filled = int((bias[i] == 0).sum())
        if filled < window:                                   # some slot still masked: append
            update_idx[i] = filled
        else:                                                 # full: roll the rolling part's PE, overwrite the oldest
            pe_idx[i, sink:] = pe_idx[i, sink:].roll(shifts=1, dims=0)
            update_idx[i] = int(pe_idx[i].argmax())
        bias[i, : min(filled + 1, window)] = 0
    return bias, pe_idx, update_idx


def retrieve_latents(encoder_output, generator=None):
    if hasattr(encoder_output, "latent_dist"):
        return encoder_output.latent_dist.sample(generator)
    if hasattr(encoder_output, "latents"):
        return encoder_output.latents
    if torch.is_tensor(encoder_output):
        return encoder_output
    raise AttributeError("could not access latents of the provided encoder_output")


class _ImageProcessor:
    """the slice of diffusers' VaeImageProcessor.preprocess the stream uses (:630): -> [B,3,H,W] in [-1,1].
    Tensor inputs are resized with F.interpolate's default (nearest) mode like VaeImageProcessor.resize does for tensors
    (third-party, parity unpinned).  `assume_unit_range=True` (what the reference's callers feed: frames in [0,1])
    skips the `image.min() < 0` probe, which is a device->host sync on every frame; None keeps diffusers' probe."""

    def __init__(self, assume_unit_range: Optional[bool] = None):
        self.assume_unit_range = assume_unit_range

    def preprocess(self, image, height: int, width: int) -> torch.Tensor:
        if not torch.is_tensor(image):
            arr = np.asarray(image)
            if arr.ndim == 3:
                arr = arr[None]
            image = torch.from_numpy(arr.astype(np.float32) / (255.0 if arr.dtype == np.uint8 else 1.0)).permute(0, 3, 1, 2)
        if image.ndim == 3:
            image = image[None]
        if image.shape[-2:] != (height, width):
            image = F.interpolate(image.float(), (height, width))
        unit = self.assume_unit_range
        if unit is None:
            unit = bool(image.min() >= 0)
        if unit:
            image = 2.0 * image - 1.0
        return image


class StreamAnimateDiffusionDepth:
    def __init__(self, pipe, num_inference_steps: int, t_index_list: Optional[List[int]] = None,
                 strength: Optional[float] = None, torch_dtype: torch.dtype = torch.float16, width: int = 512,
                 height: int = 512, do_add_noise: bool = True, use_denoising_batch: bool = True,
                 frame_buffer_size: int = 1, clip_skip: int = 1,
                 cfg_type: Literal["none", "full", "self", "initialize"] = "none",
                 warmup_frames: int = WARMUP_FRAMES, window_size: int = WINDOW_SIZE, scheduler_kwargs: Optional[dict] = None):
        self.device = pipe.device
        self.dtype = torch_dtype
        self.generator = None
        self.height, self.width = height, width
        self.pipe = pipe
        self.latent_height = int(height // pipe.vae_scale_factor)
        self.latent_width = int(width // pipe.vae_scale_factor)
        self.clip_skip = clip_skip
        self.warmup_frames, self.window_size = warmup_frames, window_size

        cfg = dict(scheduler_kwargs or getattr(getattr(pipe, "scheduler", None), "config", None) or {})
        self.scheduler = LCMSchedule(**cfg)
        self.scheduler.set_timesteps(num_inference_steps, self.device)
        if strength is not None:
            t_index_list, timesteps = self.get_timesteps(num_inference_steps, strength, self.device)
            self.timesteps = timesteps
        else:
            self.timesteps = self.scheduler.timesteps.to(self.device)
        self.frame_bff_size = frame_buffer_size
        self.denoising_steps_num = len(t_index_list)
        self.strength = strength
        assert cfg_type == "none", f'cfg_type must be "none" for now, but got {cfg_type}.'   # reference :75
        assert use_denoising_batch and frame_buffer_size == 1, "the HIP backend is built for the stream-batch mode"
        self.cfg_type = cfg_type
        self.batch_size = self.denoising_steps_num * frame_buffer_size
        self.trt_unet_batch_size = self.batch_size
        self.t_list = t_index_list
        self.do_add_noise = do_add_noise
        self.use_denoising_batch = use_denoising_batch
        self.similar_image_filter = False
        self.similar_filter = getattr(pipe, "similar_filter", None) or SimilarImageFilter()   # duck-typed, replaceable
        self.prev_image_result = None
        self.image_processor = _ImageProcessor()
        self.text_encoder = getattr(pipe, "text_encoder", None)
        self.unet = pipe.unet                       # HipStreamingUNet (or anything with the same call contract)
        self.vae = getattr(pipe, "vae", None)
        self.depth_detector = getattr(pipe, "depth_model", None)
        self._depth_glue, self._glue_off = None, False      # HipDepthGlue, created on first use on the device
        self.inference_time_ema = 0
        self.depth_time_ema = 0
        self.inference_time_list = []
        self.depth_time_list = []
        self.mask_shift = 1
        self.is_tensorrt = False
        self.unet_warmup = None

    @property
    def depth_glue(self):
        return self._depth_glue

    @depth_glue.setter
    def depth_glue(self, g):
        """assign a HipDepthGlue, or None to use the reference's torch expressions (comparison runs)"""
        self._depth_glue, self._glue_off = g, g is None

    # ------------------------------------------------------------------ cache / timesteps / lora
    def prepare_cache(self, height, width, denoising_steps_num):
        if hasattr(self.pipe, "prepare_cache"):
            self.kv_cache_list = self.pipe.prepare_cache(height=height, width=width, denoising_steps_num=denoising_steps_num)
        else:
            self.kv_cache_list = self.unet.prepare_cache(denoising_steps_num)

    def get_timesteps(self, num_inference_steps, strength, device):
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        timesteps = self.scheduler.timesteps[t_start:].to(device)
        return list(range(len(timesteps))), timesteps

    def load_lora(self, pretrained_lora_model_name_or_path_or_dict, adapter_name=None, **kwargs):
        self.pipe.load_lora_weights(pretrained_lora_model_name_or_path_or_dict, adapter_name, **kwargs)

    def fuse_lora(self, fuse_unet=True, fuse_text_encoder=True, lora_scale=1.0, safe_fusing=False):
        self.pipe.fuse_lora(fuse_unet=fuse_unet, fuse_text_encoder=fuse_text_encoder, lora_scale=lora_scale,
                            safe_fusing=safe_fusing)

    def enable_similar_image_filter(self, threshold: float = 0.98, max_skip_frame: float = 10):
        """reference :112-118"""
        if self.similar_filter is None:
            self.similar_filter = SimilarImageFilter()
        self.similar_image_filter = True
        self.similar_filter.set_threshold(threshold)
        self.similar_filter.set_max_skip_frame(max_skip_frame)

    def disable_similar_image_filter(self):
        self.similar_image_filter = False

    def load_warmup_unet(self, config=None):
        """The reference builds a second (CPU) UNet with bidirectional attention (:662-666); the HIP backend's
        warm-up path shares the streaming weights, so this only records the handle."""
        self.unet_warmup = self.unet

    # ------------------------------------------------------------------ ring buffer
    def initialize_attn_bias_pe_and_update_idx(self):
        self._rb = ring_buffer_init(self.denoising_steps_num, self.window_size, self.warmup_frames)
        return self._upload_rb()

    def update_attn_bias(self, attn_bias=None, pe_idx=None, update_idx=None):
        ring_buffer_update(*self._rb, self.window_size, self.warmup_frames)
        return self._upload_rb()

    def _upload_rb(self):
        b, p, u = self._rb
        return (b.to(device=self.device, dtype=self.dtype, non_blocking=True), p.to(self.device, non_blocking=True),
                u.to(self.device, non_blocking=True))

    # ------------------------------------------------------------------ scheduler algebra
    def add_noise(self, original_samples, noise, t_index: int):
        return self.alpha_prod_t_sqrt[t_index] * original_samples + self.beta_prod_t_sqrt[t_index] * noise

    def scheduler_step_batch(self, model_pred_batch, x_t_latent_batch, idx: Optional[int] = None):
        if idx is None:
            F_theta = (x_t_latent_batch - self.beta_prod_t_sqrt * model_pred_batch) / self.alpha_prod_t_sqrt
            return self.c_out * F_theta + self.c_skip * x_t_latent_batch
        F_theta = (x_t_latent_batch - self.beta_prod_t_sqrt[idx] * model_pred_batch) / self.alpha_prod_t_sqrt[idx]
        return self.c_out[idx] * F_theta + self.c_skip[idx] * x_t_latent_batch

    # ------------------------------------------------------------------ prepare (warm-up window)
    @torch.no_grad()
    def prepare(self, warmup_frames, prompt: str = "", negative_prompt: str = "", guidance_scale: float = 1.2,
                delta: float = 1.0, generator: Optional[torch.Generator] = None, seed: int = 2,
                prompt_embeds: Optional[torch.Tensor] = None):
        """Forward the warm-up frames ([F,3,H,W] in [0,1]) and fill the KV-cache (reference :171-344)."""
        n = self.denoising_steps_num
        self._device_step = None          # a previous stream's device-side state (ring buffer, frame counter, rows) is stale
        if generator is None:
            self.generator = torch.Generator(device=self.device)
            self.generator.manual_seed(seed)
        else:
            self.generator = generator
        lat = (4, 1, self.latent_height, self.latent_width)
        if n > 1:
            self.x_t_latent_buffer = torch.zeros((n - 1) * self.frame_bff_size, *lat, dtype=self.dtype, device=self.device)
            self.depth_latent_buffer = torch.zeros_like(self.x_t_latent_buffer)
        else:
            self.x_t_latent_buffer = self.depth_latent_buffer = None
        self.attn_bias, self.pe_idx, self.update_idx = self.initialize_attn_bias_pe_and_update_idx()
        self.guidance_scale = 1.0
        self.delta = delta
        if prompt_embeds is None:
            prompt_embeds = self.pipe._encode_prompt(prompt=prompt, device=self.device, num_videos_per_prompt=1,
                                                     do_classifier_free_guidance=False, negative_prompt=negative_prompt,
                                                     clip_skip=self.clip_skip)[0]
        self.prompt_embeds = prompt_embeds.to(device=self.device, dtype=self.dtype).reshape(1, *prompt_embeds.shape[-2:]).repeat(
            self.batch_size, 1, 1)
        self.sub_timesteps = [self.timesteps[t] for t in self.t_list]
        self.sub_timesteps_tensor = torch.tensor([int(t) for t in self.sub_timesteps], dtype=torch.long, device=self.device)
        self.init_noise = torch.randn((self.batch_size, 4, self.warmup_frames, self.latent_height, self.latent_width),
                                      generator=generator).to(device=self.device, dtype=self.dtype)
        self.stock_noise = torch.zeros_like(self.init_noise)
        cs, co, al, be = [], [], [], []
        for t in self.sub_timesteps:
            c_skip, c_out = self.scheduler.get_scalings_for_boundary_condition_discrete(int(t))
            cs.append(c_skip); co.append(c_out)
            a = self.scheduler.alphas_cumprod[int(t)]
            al.append(a.sqrt()); be.append((1 - a).sqrt())
        shp = (len(self.t_list), 1, 1, 1, 1)
        to = dict(dtype=self.dtype, device=self.device)
        self.c_skip = torch.stack(cs).view(shp).to(**to)
        self.c_out = torch.stack(co).view(shp).to(**to)
        self.alpha_prod_t_sqrt = torch.stack(al).view(shp).to(**to)
        self.beta_prod_t_sqrt = torch.stack(be).view(shp).to(**to)

        xs = [self.image_processor.preprocess(f, self.height, self.width).to(**to) for f in warmup_frames]
        warmup_x = torch.cat(xs, dim=0)
        x_t_latent = self.encode_image(warmup_x).transpose(0, 1)[None]           # [1,4,F,h,w]
        depth_latent = self.encode_depth(warmup_x).transpose(0, 1)[None]
        warm = self.unet_warmup if self.unet_warmup is not None else self.unet
        for idx, t in enumerate(self.sub_timesteps_tensor):
            if hasattr(warm, "warmup"):
                pred = warm.warmup(x_t_latent, t.view(1), encoder_hidden_states=self.prompt_embeds[0:1],
                                   depth_sample=depth_latent, kv_cache=self.kv_cache_list, row=idx)["sample"]
            else:   # a reference-style warm-up module
                pred = warm(x_t_latent, t.view(1), temporal_attention_mask=None, depth_sample=depth_latent,
                            encoder_hidden_states=self.prompt_embeds[0:1], kv_cache=[c[idx] for c in self.kv_cache_list],
                            return_dict=True)["sample"]
            x_0_pred = self.scheduler_step_batch(pred, x_t_latent, idx)
            if idx < len(self.sub_timesteps_tensor) - 1:
                x_t_latent = self.alpha_prod_t_sqrt[idx + 1] * x_0_pred + self.beta_prod_t_sqrt[idx + 1] * torch.randn_like(x_0_pred)
        frames = self.decode_image(x_0_pred[0].transpose(0, 1))
        self.warmup_engine()
        return frames

    def warmup_engine(self):
        """reference :346-366 (TensorRT engine warm-up); the HIP plan needs none."""
        return

    @torch.no_grad()
    def update_prompt(self, prompt: str):
        emb = self.pipe._encode_prompt(prompt=prompt, device=self.device, num_videos_per_prompt=1,
                                       do_classifier_free_guidance=False)[0]
        self.prompt_embeds = emb.to(device=self.device, dtype=self.dtype).repeat(self.batch_size, 1, 1)
        inval = getattr(self.unet, "invalidate_text_cache", None)
        if inval is not None:             # explicit: never rely on tensor identity alone for a prompt change
            inval()
        ds = getattr(self, "_device_step", None)
        if ds is not None:                # the device step reads the prompt from the plan's static input buffer
            ds.set_prompt(self.prompt_embeds)

    # ------------------------------------------------------------------ per-frame
    def unet_step(self, x_t_latent, depth_latent, t_list, idx: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        output = self.unet(x_t_latent, t_list, depth_sample=depth_latent, encoder_hidden_states=self.prompt_embeds,
                           temporal_attention_mask=self.attn_bias, kv_cache=self.kv_cache_list, pe_idx=self.pe_idx,
                           update_idx=self.update_idx, return_dict=True)
        model_pred = output["sample"]
        self.kv_cache_list = output["kv_cache"]
        return self.scheduler_step_batch(model_pred, x_t_latent, idx), model_pred

    def encode_image(self, image_tensors):
        image_tensors = image_tensors.to(device=self.device, dtype=self.vae.dtype)
        img_latent = retrieve_latents(self.vae.encode(image_tensors), self.generator) * self.vae.config.scaling_factor
        noise = torch.randn(img_latent.shape, device=img_latent.device, dtype=img_latent.dtype, generator=self.generator)
        return self.add_noise(img_latent, noise, 0)

    def decode_image(self, x_0_pred_out):
        out = self.vae.decode(x_0_pred_out / self.vae.config.scaling_factor, return_dict=False)[0]
        return out.clip(-1, 1)

    def encode_depth(self, image_tensors):
        """reference :544-571.  On the device the arithmetic around the (caller-owned) depth detector -- the 384x384 bilinear
        resize, the min-max normalisation over the batch, x3 channels, [-1,1], the resize back -- runs as HIP ops without the
        reference's two host-visible reductions (`self.depth_glue`, vae_hip.HipDepthGlue); CPU tensors (host-logic tests)
        and `depth_glue = None` take the reference's torch expressions."""
        image_tensors = image_tensors.to(device=self.device, dtype=self.depth_detector.dtype)
        h, w = image_tensors.shape[2], image_tensors.shape[3]
        glue = self.depth_glue
        if glue is None and image_tensors.is_cuda and image_tensors.dtype == torch.float16 and not self._glue_off:
            from .vae_hip import HipDepthGlue
            glue = self.depth_glue = HipDepthGlue(self.device)
        if glue is not None:
            depth_map = self.depth_detector(glue.resize(image_tensors, 384, 384))
            dn = glue.normalize_resize(depth_map.to(torch.float16), h, w)
        else:
            images_input = F.interpolate(image_tensors, (384, 384), mode="bilinear", align_corners=False)
            depth_map = self.depth_detector(images_input)
            dn = (depth_map - depth_map.min()) / (depth_map.max() - depth_map.min())
            dn = dn[:, None].repeat(1, 3, 1, 1) * 2 - 1
            dn = F.interpolate(dn, (h, w), mode="bilinear", align_corners=False)
        return retrieve_latents(self.vae.encode(dn.to(dtype=self.vae.dtype)), self.generator) * self.vae.config.scaling_factor

    def enable_device_step(self, use_graph: bool = False, seed: int = 0):
        """Opt in to the device-side frame step (stream_step_hip.HipStreamStep): after `prepare`,
        everything `predict_x0_batch` does around the UNet (batch assembly, LCM step, buffer shift, re-noising,
        ring-buffer update) runs as ops of the UNet's own static plan.  Needs the HIP UNet backend; frame_bff_size 1."""
        from .stream_step_hip import HipStreamStep
        from .unet_hip import HipStreamingUNet
        if not isinstance(self.unet, HipStreamingUNet) or self.frame_bff_size != 1:
            raise ValueError("enable_device_step needs the HipStreamingUNet backend and frame_bff_size == 1")
        self._device_step = HipStreamStep(self.unet, self.kv_cache_list, self.sub_timesteps_tensor, self.prompt_embeds,
                                          self.alpha_prod_t_sqrt, self.beta_prod_t_sqrt, self.c_skip, self.c_out,
                                          do_add_noise=self.do_add_noise, seed=seed, use_graph=use_graph,
                                          ring_state=self._rb)      # frames may already have run on the host path
        self._device_step.load_buffers(self.x_t_latent_buffer, self.depth_latent_buffer)
        return self._device_step

    # ------------------------------------------------------------------ pipelined frames (opt-in; not in the reference)
    def enable_frame_pipelining(self):
        """Opt in to `push(frame)` / `pop()`.  The reference's `__call__` is synchronous: encode, depth path, UNet and decode of a
        frame run back to back and the call ends with a global synchronize (:643-654), although everything in front of the UNet
        -- preprocess, TAESD encode + noise, 384x384 resize, depth detector, min-max / resize, TAESD encode of the depth map --
        depends on the incoming frame only.  `push(frame)` starts that part on a second HIP stream and returns; `pop()` runs the
        UNet step and the decode of the OLDEST pushed frame on the caller's stream and returns its output.  A caller that
        pushes frame t + 1 before it pops frame t overlaps the two (the UNet step is a chain of latency-bound launches that
        leaves most CUs idle), at the price of holding one frame more in flight.  Results are those of
        `__call__` on the same frames, bit for bit: needs the device step (its re-noising draws from the plan's own
        counter-based generator, so the order of the torch.randn draws of `encode_image` is all that is left on the host
        generator and it is the same in both modes); the near-duplicate frame filter is not consulted in this mode."""
        import collections
        if getattr(self, "_device_step", None) is None:
            raise ValueError("enable_frame_pipelining needs enable_device_step() first")
        self._pre_stream = torch.cuda.Stream(device=self.device)
        self._pending = collections.deque()

    def push(self, x: Union[torch.Tensor, np.ndarray]) -> None:
        if getattr(self, "_pending", None) is None:
            raise RuntimeError("push() needs enable_frame_pipelining() first")
        cur = torch.cuda.current_stream()
        x = self.image_processor.preprocess(x, self.height, self.width).to(device=self.device, dtype=self.dtype)
        self._pre_stream.wait_stream(cur)                     # the frame is ready; earlier consumers of the side buffers are done
        with torch.cuda.stream(self._pre_stream):
            x_t_latent = self.encode_image(x)
            depth_latent = self.encode_depth(x)
            ev = torch.cuda.Event()
            ev.record(self._pre_stream)
        if x.is_cuda:
            x.record_stream(self._pre_stream)
        self._pending.append((x_t_latent, depth_latent, ev))

    def pop(self) -> torch.Tensor:
        """(The depth-path time of a popped frame is hidden under the previous frame's UNet step: `depth_time_ema` /
        `depth_time_list` are not updated in this mode; `inference_time_ema` is.)"""
        if getattr(self, "_pending", None) is None:
            raise RuntimeError("pop() needs enable_frame_pipelining() first")
        if not self._pending:
            raise RuntimeError("pop() without a pushed frame")
        x_t_latent, depth_latent, ev = self._pending.popleft()
        cur = torch.cuda.current_stream()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        cur.wait_event(ev)
        for t in (x_t_latent, depth_latent):
            if t.is_cuda:
                t.record_stream(cur)
        x_0 = self.predict_x0_batch(x_t_latent.unsqueeze(2), depth_latent.unsqueeze(2))
        x_output = self.decode_image(x_0[:, :, 0]).detach().clone()
        self.prev_image_result = x_output
        end.record()
        cur.synchronize()                                     # (this stream only: the next frame's side-stream work keeps running)
        inference_time = start.elapsed_time(end) / 1000
        self.inference_time_ema = 0.9 * self.inference_time_ema + 0.1 * inference_time
        self.inference_time_list.append(inference_time)
        return x_output

    def predict_x0_batch(self, x_t_latent, depth_latent, noise: Optional[torch.Tensor] = None):
        """reference :573-623 (stream-batch shift register). `noise` lets tests inject the re-noising tensor."""
        n = self.denoising_steps_num
        ds = getattr(self, "_device_step", None)
        if ds is not None:
            if noise is not None:
                raise ValueError("device step is enabled: its ring buffer / latent rows live on the device and the host "
                                 "path would run with stale state; inject noise through HipStreamStep(inject_noise=True)")
            x_0_pred_out = ds.step(x_t_latent, depth_latent)
            self.attn_bias, self.pe_idx, self.update_idx = ds.attn_bias, ds.pe_idx, ds.update_idx
            if n > 1:
                self.x_t_latent_buffer = ds.x_t_latent_buffer
            return x_0_pred_out
        if n > 1:
            x_t_latent = torch.cat((x_t_latent, self.x_t_latent_buffer), dim=0)
            depth_latent = torch.cat((depth_latent, self.depth_latent_buffer), dim=0)
        x_0_pred_batch, _ = self.unet_step(x_t_latent, depth_latent, self.sub_timesteps_tensor)
        self.attn_bias, self.pe_idx, self.update_idx = self.update_attn_bias()
        if n > 1:
            x_0_pred_out = x_0_pred_batch[-1].unsqueeze(0)
            if self.do_add_noise:
                nz = torch.randn_like(x_0_pred_batch[:-1]) if noise is None else noise
                self.x_t_latent_buffer = self.alpha_prod_t_sqrt[1:] * x_0_pred_batch[:-1] + self.beta_prod_t_sqrt[1:] * nz
            else:
                self.x_t_latent_buffer = self.alpha_prod_t_sqrt[1:] * x_0_pred_batch[:-1]
            self.depth_latent_buffer = depth_latent[:-1]
        else:
            x_0_pred_out = x_0_pred_batch
            self.x_t_latent_buffer = None
        return x_0_pred_out

    @torch.no_grad()
    def __call__(self, x: Union[torch.Tensor, np.ndarray]) -> torch.Tensor:
        if getattr(self, "_pending", None):
            # the side stream may still be writing the static plan buffers this call would use (TAESD encoder, depth detector
            # arena, glue scratch), and the host generator's draw order would change: finish the pushed frames first
            raise RuntimeError("__call__ while pushed frames are pending: pop() them first (push / pop and __call__ share plan buffers)")
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        x = self.image_processor.preprocess(x, self.height, self.width).to(device=self.device, dtype=self.dtype)
        if self.similar_image_filter:
            x = self.similar_filter(x)
            if x is None:
                time.sleep(self.inference_time_ema)
                return self.prev_image_result
        x_t_latent = self.encode_image(x)
        sd, ed = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sd.record()
        depth_latent = self.encode_depth(x)
        ed.record()
        x_0 = self.predict_x0_batch(x_t_latent.unsqueeze(2), depth_latent.unsqueeze(2))      # [1,4,1,h,w]
        x_output = self.decode_image(x_0[:, :, 0]).detach().clone()
        self.prev_image_result = x_output
        end.record()
        torch.cuda.synchronize()
        inference_time = start.elapsed_time(end) / 1000
        depth_time = sd.elapsed_time(ed) / 1000
        self.inference_time_ema = 0.9 * self.inference_time_ema + 0.1 * inference_time
        self.depth_time_ema = 0.9 * self.depth_time_ema + 0.1 * depth_time
        self.inference_time_list.append(inference_time)
        self.depth_time_list.append(depth_time)
        return x_output
