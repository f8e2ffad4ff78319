"""Generate the golden fixtures in this directory from the REFERENCE's own classes.

Run in the build container only (needs /root/reference):   python tests/golden/gen_golden.py
The reference source never leaves that container: only inputs/outputs (fp32 .npz, KB-sized) and
this script are committed.  Weights are never stored -- both sides rebuild them from the key-hashed
generator `live2diff_amd.weights._fill` (seed = crc32(key)).

What is reference-pinned vs stub-pinned is stated in diffusers_stub.py / SURVEY.md section 8c.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_loader  # noqa: E402
from live2diff_amd.config import tiny_config  # noqa: E402
from live2diff_amd.weights import _fill, unet_param_spec  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)
R = ref_loader.load()


def rnd(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def fill_module(mod, prefix):
    """key-hashed fill of a reference module; keys are prefixed so that tests can regenerate them."""
    sd = mod.state_dict()
    new = {}
    for k, v in sd.items():
        if "pos_encoder" in k or k.endswith("_pe"):
            new[k] = v
        else:
            new[k] = _fill(prefix + k, tuple(v.shape), 1.0)
    mod.load_state_dict(new)
    return mod


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


def mm_kwargs(cfg):
    return dict(num_attention_heads=cfg.temporal_heads, num_transformer_block=1,
                attention_block_types=["Temporal_Self", "Temporal_Self"], temporal_position_encoding=True,
                temporal_position_encoding_max_len=cfg.temporal_max_len, temporal_attention_dim_div=1,
                zero_initialize=True, attention_class_name="stream",
                attention_kwargs=dict(window_size=cfg.window_size, sink_size=cfg.sink_size))


def unet_kwargs(cfg, streaming=True):
    kw = dict(cond_mapping=True, use_inflated_groupnorm=True, use_motion_module=True,
              motion_module_resolutions=(1, 2, 4, 8), unet_use_cross_frame_attention=False,
              unet_use_temporal_attention=False, block_out_channels=cfg.block_out_channels,
              cross_attention_dim=cfg.cross_attention_dim)
    mk = mm_kwargs(cfg)
    if streaming:
        kw.update(motion_module_type="Streaming", motion_module_kwargs=mk)
    else:  # reference unet_depth_warmup.py:611-614
        mk = dict(mk, attention_class_name="versatile", attention_kwargs={})
        kw.update(motion_module_type="Vanilla", motion_module_kwargs=mk)
    return kw


# ----------------------------------------------------------------------------- 1. PE table
def gen_pe():
    pe = R.positional_encoding.PositionalEncoding(64, max_len=40).pe[0]
    save("pe_table", pe=pe)


# ----------------------------------------------------------------------------- 2. ring buffer + LCM step
def gen_state_machine():
    P = ref_loader.load_pipeline_class()
    cls = P.StreamAnimateDiffusionDepth
    torch.Tensor.cuda = lambda self, *a, **k: self   # reference hard-codes .cuda() (:410)
    out = {}
    for n in (2, 3, 4):
        fake = type("S", (), {})()
        fake.denoising_steps_num = n
        fake.device = "cpu"
        fake.dtype = torch.float32
        bias, pe_idx, upd = cls.initialize_attn_bias_pe_and_update_idx(fake)
        bs, ps, us = [bias.clone()], [pe_idx.clone()], [upd.clone()]
        for _ in range(40):
            bias, pe_idx, upd = cls.update_attn_bias(fake, bias, pe_idx, upd)
            bs.append(bias.clone()); ps.append(pe_idx.clone()); us.append(upd.clone())
        out[f"bias_n{n}"] = torch.stack(bs)
        out[f"pe_idx_n{n}"] = torch.stack(ps)
        out[f"update_idx_n{n}"] = torch.stack(us)
    # LCM step (scheduler_step_batch :387-401) with explicit scalars
    fake = type("S", (), {})()
    n = 3
    fake.alpha_prod_t_sqrt = torch.tensor([0.3, 0.6, 0.9]).view(n, 1, 1, 1, 1)
    fake.beta_prod_t_sqrt = (1 - fake.alpha_prod_t_sqrt ** 2).sqrt()
    fake.c_skip = torch.tensor([0.01, 0.05, 0.2]).view(n, 1, 1, 1, 1)
    fake.c_out = torch.tensor([0.99, 0.97, 0.8]).view(n, 1, 1, 1, 1)
    x = rnd(n, 4, 1, 4, 4, seed=11)
    eps = rnd(n, 4, 1, 4, 4, seed=12)
    out["lcm_x"], out["lcm_eps"] = x, eps
    out["lcm_alpha"], out["lcm_beta"] = fake.alpha_prod_t_sqrt, fake.beta_prod_t_sqrt
    out["lcm_c_skip"], out["lcm_c_out"] = fake.c_skip, fake.c_out
    out["lcm_x0"] = cls.scheduler_step_batch(fake, eps, x)
    out["lcm_x0_idx1"] = cls.scheduler_step_batch(fake, eps[1:2], x[1:2], 1)
    out["add_noise_1"] = cls.add_noise(fake, x[1:2], eps[1:2], 1)
    save("state_machine", **out)


# ----------------------------------------------------------------------------- 3. StreamTemporalAttention
def gen_stream_attn():
    cases = []
    for ci, (C, T, L, S, N) in enumerate([(64, 16, 16, 8, 2), (64, 16, 12, 4, 1), (128, 8, 24, 8, 4), (64, 4, 40, 8, 2)]):
        att = R.stream_motion_module.StreamTemporalAttention(
            attention_mode="Temporal", temporal_position_encoding=True, temporal_position_encoding_max_len=max(24, L),
            window_size=L, sink_size=S, query_dim=C, heads=8, dim_head=C // 8, cross_attention_dim=None, bias=False)
        fill_module(att, f"sta{ci}.")
        att.set_info(T, 1)
        cache = att.set_cache(N)
        att.prepare_pe_buffer()
        cache.copy_(rnd(*cache.shape, seed=100 + ci))
        g = torch.Generator().manual_seed(200 + ci)
        pe_idx = torch.stack([torch.cat([torch.arange(S), S + torch.randperm(L - S, generator=g)]) for _ in range(N)])
        upd = torch.randint(S, L, (N,), generator=g)
        bias = torch.zeros(N, L)
        for n in range(N):
            if n % 2 == 1:  # partially masked row (ramp-up phase)
                bias[n, S + 2:] = float("-inf")
        x = rnd(N, T, C, seed=300 + ci)    # "(b f) d c" with f=1 : [N, hw, C]
        cache_in = cache.clone()
        out = att(x, video_length=1, temporal_attention_mask=bias, kv_cache=cache, pe_idx=pe_idx, update_idx=upd)
        cases.append((C, T, L, S, N))
        save(f"stream_attn_{ci}", x=x, cache_in=cache_in, cache_out=cache, pe_idx=pe_idx, update_idx=upd, bias=bias,
             out=out, meta=np.array([C, T, L, S, N]))


# ----------------------------------------------------------------------------- 4./5. motion module (stream + warm-up)
def gen_motion_module():
    cfg = tiny_config()
    C, H, W, N = 64, 4, 4, 2
    mm = R.motion_module.get_motion_module(C, "Streaming", mm_kwargs(cfg))
    fill_module(mm, "mm.")
    caches = []
    for j, a in enumerate(mm.temporal_transformer.transformer_blocks[0].attention_blocks):
        a.set_info(H, W)
        a.set_index(j)
        c = a.set_cache(N)
        a.prepare_pe_buffer()
        c.copy_(rnd(*c.shape, seed=400 + j))
        caches.append(c)
    x = rnd(N, C, 1, H, W, seed=410)
    pe_idx = torch.arange(16).repeat(N, 1)
    pe_idx[0, 8:] = torch.roll(pe_idx[0, 8:], 3)
    upd = torch.tensor([pe_idx[0].argmax().item(), 10])
    bias = torch.zeros(N, 16)
    bias[1, 11:] = float("-inf")
    cin = [c.clone() for c in caches]
    out = mm(x, None, None, temporal_attention_mask=bias, kv_cache=caches, pe_idx=pe_idx, update_idx=upd)
    save("motion_module_stream", x=x, pe_idx=pe_idx, update_idx=upd, bias=bias, out=out,
         cache_in0=cin[0], cache_in1=cin[1], cache_out0=caches[0], cache_out1=caches[1])

    # warm-up twin (VersatileAttention), same weights
    mk = dict(mm_kwargs(cfg), attention_class_name="versatile", attention_kwargs={})
    mw = R.motion_module.get_motion_module(C, "Vanilla", mk)
    fill_module(mw, "mm.")
    F_ = 8
    rows = [torch.zeros(2, H * W, 16, C) for _ in range(2)]
    for j, a in enumerate(mw.temporal_transformer.transformer_blocks[0].attention_blocks):
        a.set_info(H, W)
        a.set_index(j)
    xw = rnd(1, C, F_, H, W, seed=420)
    outw = mw(xw, None, None, temporal_attention_mask=None, kv_cache=rows)
    save("motion_module_warmup", x=xw, out=outw, cache_out0=rows[0], cache_out1=rows[1])


# ----------------------------------------------------------------------------- 6. resnet family
def gen_resnet_family():
    for name, cin, cout in (("resnet_same", 64, 64), ("resnet_proj", 96, 64)):
        r = R.resnet.ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=128, eps=1e-5, groups=32,
                                   use_inflated_groupnorm=True)
        fill_module(r, name + ".")
        x = rnd(2, cin, 1, 6, 5, seed=500)
        temb = rnd(2, 128, seed=501)
        save(name, x=x, temb=temb, out=r(x, temb))
    d = R.resnet.Downsample3D(64, use_conv=True, out_channels=64, padding=1, name="op")
    fill_module(d, "down.")
    x = rnd(2, 64, 1, 6, 8, seed=510)
    save("downsample", x=x, out=d(x))
    u = R.resnet.Upsample3D(64, use_conv=True, out_channels=64)
    fill_module(u, "up.")
    x = rnd(2, 64, 1, 3, 4, seed=520)
    save("upsample", x=x, out=u(x))
    m = R.resnet.MappingNetwork(conditioning_embedding_channels=64, conditioning_channels=4)
    fill_module(m, "map.")
    x = rnd(2, 4, 1, 6, 5, seed=530)
    save("mapping", x=x, out=m(x))


# ----------------------------------------------------------------------------- 7. spatial transformer (stub-pinned)
def gen_spatial():
    t = R.attention.Transformer3DModel(8, 8, in_channels=64, num_layers=1, cross_attention_dim=96, norm_num_groups=32,
                                       unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
    fill_module(t, "sp.")
    x = rnd(2, 64, 1, 5, 4, seed=600)
    enc = rnd(2, 7, 96, seed=601)
    save("spatial_transformer", x=x, enc=enc, out=t(x, encoder_hidden_states=enc).sample)


# ----------------------------------------------------------------------------- 8. tiny full UNet rollout
def gen_unet_rollout():
    P = ref_loader.load_pipeline_class()
    cls = P.StreamAnimateDiffusionDepth
    torch.Tensor.cuda = lambda self, *a, **k: self
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    spec = unet_param_spec(cfg)
    sd = {k: _fill(k, shp, 1.0) for k, shp in spec.items()}
    us = R.unet_depth_streaming.UNet3DConditionStreamingModel(**unet_kwargs(cfg, True))
    uw = R.unet_depth_warmup.UNet3DConditionWarmupModel(**unet_kwargs(cfg, False))
    m1, u1 = us.load_state_dict(sd, strict=False)
    m2, u2 = uw.load_state_dict(sd, strict=False)
    assert not u1 and not u2 and all("pos_encoder" in k for k in m1 + m2), (m1, u1, m2, u2)
    h = w = 16     # deepest level 2x2: a 1x1 level makes GroupNorm (2 values/group) amplify fp32 noise
    N, FR = 2, 12
    us.set_info_for_attn(h, w)
    uw.set_info_for_attn(h, w)
    kv = us.prepare_cache(N)
    enc = rnd(1, 5, cfg.cross_attention_dim, seed=700)
    tsteps = torch.tensor([399, 199])
    # warm-up: N passes of 8 frames, each filling cache row idx (pipeline :317-328)
    wx = rnd(N, 4, 8, h, w, seed=701)
    wd = rnd(1, 4, 8, h, w, seed=702)
    wout = []
    for idx in range(N):
        o = uw(wx[idx:idx + 1], tsteps[idx:idx + 1], temporal_attention_mask=None, depth_sample=wd,
               encoder_hidden_states=enc, kv_cache=[c[idx] for c in kv], return_dict=True)["sample"]
        wout.append(o)
    fake = type("S", (), {})()
    fake.denoising_steps_num = N
    fake.device = "cpu"
    fake.dtype = torch.float32
    bias, pe_idx, upd = cls.initialize_attn_bias_pe_and_update_idx(fake)
    xs = rnd(FR, N, 4, 1, h, w, seed=703)
    ds = rnd(FR, N, 4, 1, h, w, seed=704)
    outs = []
    for f in range(FR):
        o = us(xs[f], tsteps, encoder_hidden_states=enc.repeat(N, 1, 1), temporal_attention_mask=bias,
               depth_sample=ds[f], kv_cache=kv, pe_idx=pe_idx, update_idx=upd)
        assert o["kv_cache"] is kv
        outs.append(o["sample"].clone())
        bias, pe_idx, upd = cls.update_attn_bias(fake, bias, pe_idx, upd)
    cache_sum = torch.stack([c.double().sum() for c in kv])
    cache_sq = torch.stack([(c.double() ** 2).sum() for c in kv])
    cache_slice = torch.stack([c[:, :, :1, :, :8] for c in kv])
    save("unet_rollout", enc=enc, tsteps=tsteps, warm_x=wx, warm_depth=wd, warm_out=torch.stack(wout),
         xs=xs, ds=ds, outs=torch.stack(outs), cache_sum=cache_sum, cache_sq=cache_sq, cache_slice=cache_slice,
         meta=np.array([h, w, N, FR]))
    with open(os.path.join(HERE, "param_spec_tiny.json"), "w") as f:
        ref_spec = {k: list(v.shape) for k, v in us.state_dict().items() if "pos_encoder" not in k and not k.endswith("_pe")}
        json.dump(ref_spec, f, indent=0, sort_keys=True)


# ----------------------------------------------------------------------------- 8. prepare() + per-frame __call__ chain
def gen_pipeline_chain():
    """StreamAnimateDiffusionDepth.prepare (:171-344, incl. the x0 -> re-noise chain between warm-up passes) followed by six
    `__call__`s (:625-666: encode_image, encode_depth, predict_x0_batch, decode_image) of the REFERENCE class, run unbound on a
    fake `self` with the deterministic mocks of tests/pipeline_mocks.py.  Captures outputs, buffers, ring-buffer state and the
    caches after every step; the RNG contract (global seed 123, generator seed 2) is part of what is pinned."""
    sys.path.insert(0, os.path.dirname(HERE))
    import pipeline_mocks as M
    P = ref_loader.load_pipeline_class()
    cls = P.StreamAnimateDiffusionDepth
    P.retrieve_latents = M.retrieve_latents
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.Event = M.NoCudaEvent
    torch.cuda.synchronize = lambda *a, **k: None
    out = {}
    for n, t_index in ((2, [30, 40]), (3, [20, 30, 45])):
        sch = M.MockScheduler()
        fake = type("S", (), {})()
        for name in ("prepare", "initialize_attn_bias_pe_and_update_idx", "update_attn_bias", "scheduler_step_batch", "add_noise",
                     "encode_image", "decode_image", "encode_depth", "predict_x0_batch", "unet_step", "warmup_engine"):
            setattr(fake, name, getattr(cls, name).__get__(fake))
        fake.device, fake.dtype = torch.device("cpu"), torch.float32
        fake.height, fake.width, fake.latent_height, fake.latent_width = M.H, M.W, M.H // 8, M.W // 8
        fake.denoising_steps_num, fake.frame_bff_size, fake.batch_size = n, 1, n
        fake.cfg_type, fake.use_denoising_batch, fake.do_add_noise, fake.clip_skip = "none", True, True, 1
        fake.t_list, fake.scheduler, fake.timesteps = t_index, sch, sch.timesteps
        fake.pipe, fake.image_processor, fake.vae, fake.depth_detector = M.MockPipe(), M.MockImageProcessor(), M.MockVAE(), M.MockDepth()
        fake.unet, fake.unet_warmup = M.MockStreamUNet(), M.MockWarmupUNet()
        fake.kv_cache_list = M.make_caches(n)
        fake.similar_image_filter, fake.is_tensorrt = False, False
        fake.inference_time_ema = fake.depth_time_ema = 0
        fake.inference_time_list, fake.depth_time_list = [], []
        torch.manual_seed(123)
        warm = fake.prepare(M.frames(8, seed=7), "a prompt", seed=2)
        k = f"n{n}_"
        out[k + "prepare_out"] = warm
        out[k + "prepare_caches"] = torch.stack(fake.kv_cache_list)
        out[k + "sub_timesteps"] = fake.sub_timesteps_tensor
        out[k + "c_skip"], out[k + "c_out"] = fake.c_skip, fake.c_out
        out[k + "alpha"], out[k + "beta"] = fake.alpha_prod_t_sqrt, fake.beta_prod_t_sqrt
        out[k + "init_noise"] = fake.init_noise
        outs, bufs, dbufs = [], [], []
        for i, img in enumerate(M.frames(6, seed=11)):
            outs.append(cls.__call__(fake, img))
            bufs.append(fake.x_t_latent_buffer.clone())
            dbufs.append(fake.depth_latent_buffer.clone())
        out[k + "frame_out"] = torch.stack(outs)
        out[k + "x_t_buffer"], out[k + "depth_buffer"] = torch.stack(bufs), torch.stack(dbufs)
        out[k + "caches"] = torch.stack(fake.kv_cache_list)
        out[k + "bias"], out[k + "pe_idx"], out[k + "update_idx"] = fake.attn_bias, fake.pe_idx, fake.update_idx
        out[k + "unet_t"] = torch.stack([c["t"] for c in fake.unet.log])
        out[k + "unet_update_idx"] = torch.stack([c["update_idx"] for c in fake.unet.log])
    save("pipeline_chain", **out)


if __name__ == "__main__":
    gen_pe()
    gen_state_machine()
    gen_stream_attn()
    gen_motion_module()
    gen_resnet_family()
    gen_spatial()
    gen_unet_rollout()
    gen_pipeline_chain()
    print("done")
