"""-m gpu: the device-side per-frame pipeline glue (csrc/stream_glue.hip, stream_step_hip.py).

Integer / byte work (the ring-buffer state machine) is checked BIT-EXACTLY against the 40-frame trace captured from
the reference (tests/golden/state_machine.npz) and against the host restatement for the other window sizes; the LCM
step + shift register is checked bit-exactly against the reference-pinned host tensor expression evaluated in fp16
(same rounding points); the noise generator has no reference stream to match (torch.randn is backend specific), so it
is checked for reproducibility and for its distribution.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n", [2, 3, 4])
def test_ring_update_matches_reference_trace(golden, n):
    from live2diff_amd import ops
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init
    g = golden("state_machine")
    b, p, u = ring_buffer_init(n)
    bd, pd, ud = b.half().to(DEV), p.to(DEV), u.to(DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    for f in range(41):
        assert torch.equal(bd.float().cpu(), torch.from_numpy(g[f"bias_n{n}"][f])), f
        assert torch.equal(pd.cpu(), torch.from_numpy(g[f"pe_idx_n{n}"][f])), f
        assert torch.equal(ud.cpu(), torch.from_numpy(g[f"update_idx_n{n}"][f])), f
        ops.run(ops.ring_update(bd, pd, ud, N=n, L=16, sink=8, frame_ctr=ctr))
        torch.cuda.synchronize()
    assert int(ctr) == 41


@pytest.mark.parametrize("n,L,S", [(1, 12, 4), (2, 24, 8), (2, 40, 8), (4, 16, 8), (8, 16, 8)])
def test_ring_update_other_windows_match_host(n, L, S):
    from live2diff_amd import ops
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init, ring_buffer_update
    b, p, u = ring_buffer_init(n, L, S)
    bd, pd, ud = b.half().to(DEV), p.to(DEV), u.to(DEV)
    for f in range(3 * L):
        ring_buffer_update(b, p, u, L, S)
        ops.run(ops.ring_update(bd, pd, ud, N=n, L=L, sink=S))
        torch.cuda.synchronize()
        assert torch.equal(bd.float().cpu(), b) and torch.equal(pd.cpu(), p) and torch.equal(ud.cpu(), u), f


@pytest.mark.parametrize("N,noise_on,depth_on", [(2, True, True), (4, True, True), (3, False, True), (1, True, False), (2, True, False)])
def test_stream_shift_bit_exact(golden, N, noise_on, depth_on):
    """x0 / next-buffer arithmetic with the reference's fp16 rounding points: bit-identical to the reference-pinned host
    expressions (scheduler_step_batch + the buffer update of predict_x0_batch) evaluated with fp16 tensors."""
    from live2diff_amd import ops
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth as S
    from live2diff_amd.scheduler import LCMSchedule
    sch = LCMSchedule()
    sch.set_timesteps(50)
    ts = [399, 299, 199, 99][:N]
    al = torch.tensor([float(sch.alphas_cumprod[t]) ** 0.5 for t in ts])
    be = torch.tensor([(1 - float(sch.alphas_cumprod[t])) ** 0.5 for t in ts])
    cs, co = zip(*[sch.get_scalings_for_boundary_condition_discrete(t) for t in ts])
    shp = (N, 1, 1, 1, 1)
    fake = S.__new__(S)
    to = dict(device=DEV, dtype=torch.float16)
    fake.alpha_prod_t_sqrt, fake.beta_prod_t_sqrt = al.view(shp).to(**to), be.view(shp).to(**to)
    fake.c_skip = torch.tensor([float(c) for c in cs]).view(shp).to(**to)
    fake.c_out = torch.tensor([float(c) for c in co]).view(shp).to(**to)
    g = torch.Generator(device=DEV).manual_seed(3)
    h = w = 16
    per = 4 * h * w
    x = torch.randn(N, 4, 1, h, w, generator=g, **to)
    eps = torch.randn(N, 4, 1, h, w, generator=g, **to)
    nz = torch.randn(max(N - 1, 1), 4, 1, h, w, generator=g, **to)
    dep = torch.randn(N, 4, 1, h, w, generator=g, **to)
    # reference-pinned host expressions (fp16 tensors on the device)
    x0 = S.scheduler_step_batch(fake, eps, x)
    want_out = x0[-1]
    if N > 1:
        want_buf = fake.alpha_prod_t_sqrt[1:] * x0[:-1]
        if noise_on:
            want_buf = want_buf + fake.beta_prod_t_sqrt[1:] * nz[: N - 1]
    scal = torch.stack([fake.alpha_prod_t_sqrt.reshape(-1).float(), fake.beta_prod_t_sqrt.reshape(-1).float(),
                        fake.c_skip.reshape(-1).float(), fake.c_out.reshape(-1).float()], 1).contiguous()
    xd, dd = x.clone(), dep.clone()
    out = torch.zeros(per, **to)
    ops.run(ops.stream_shift(xd, eps, scal, out, N=N, per=per, noise=(nz if noise_on else None), depth=(dd if depth_on else None)))
    torch.cuda.synchronize()
    assert torch.equal(out.view_as(want_out), want_out)
    assert torch.equal(xd[0], x[0])                       # row 0 is the caller's (next frame's latent goes there)
    if N > 1:
        assert torch.equal(xd[1:], want_buf)
        if depth_on:
            assert torch.equal(dd[1:], dep[:-1]) and torch.equal(dd[0], dep[0])


def test_randn_reproducible_and_normal():
    from live2diff_amd import ops
    n = 1 << 20
    a = torch.empty(n, dtype=torch.float16, device=DEV)
    b = torch.empty(n, dtype=torch.float16, device=DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.run(ops.randn(a, seed=7, frame_ctr=ctr))
    ops.run(ops.randn(b, seed=7, frame_ctr=ctr))
    torch.cuda.synchronize()
    assert torch.equal(a, b)                              # same (seed, stream position) -> same tensor
    ctr += 1
    ops.run(ops.randn(b, seed=7, frame_ctr=ctr))
    c = torch.empty(n - 3, dtype=torch.float16, device=DEV)
    ops.run(ops.randn(c, seed=8))
    torch.cuda.synchronize()
    assert not torch.equal(a, b) and abs(float((a.float() * b.float()).mean())) < 5e-3      # next frame: fresh, uncorrelated
    for t in (a, b, c):
        f = t.double().cpu()
        assert torch.isfinite(f).all()
        m, v = f.mean().item(), f.var().item()
        k = ((f - m) ** 4).mean().item() / v ** 2
        assert abs(m) < 5e-3 and abs(v - 1) < 1e-2 and abs(k - 3) < 5e-2, (m, v, k)
        assert 0.30 < (f.abs() > 1).double().mean().item() < 0.335                          # P(|z| > 1) = 0.3173


@pytest.mark.parametrize("N,use_graph", [(2, False), (3, True)])
def test_device_step_equals_host_pipeline(golden, N, use_graph):
    """HipStreamStep (everything between two frames on the device, one static plan / hipGraph) against the host-driven
    path -- UNet boundary call + torch tensor expressions + host ring buffer -- with the re-noising tensor injected
    into both: identical x0 outputs, x_t buffers, ring-buffer states and KV caches over warm ring-buffer wrap-around."""
    from live2diff_amd.config import tiny_config
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth as S
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init, ring_buffer_update
    from live2diff_amd.scheduler import LCMSchedule
    from live2diff_amd.stream_step_hip import HipStreamStep
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    h = w = 16
    sd = {k: v.to(DEV) for k, v in random_state_dict(cfg, dtype=torch.float16).items()}
    ua, ub = HipStreamingUNet(sd, cfg, h, w, N), HipStreamingUNet(sd, cfg, h, w, N)
    g = torch.Generator(device=DEV).manual_seed(11)
    to = dict(device=DEV, dtype=torch.float16)
    rn = lambda *s: torch.randn(*s, generator=g, **to)
    kva, kvb = ua.prepare_cache(N), ub.prepare_cache(N)
    for a, b in zip(kva, kvb):
        a.normal_(generator=g)
        b.copy_(a)
    sch = LCMSchedule()
    sch.set_timesteps(50)
    ts_list = [399, 299, 199][:N]
    shp = (N, 1, 1, 1, 1)
    fake = S.__new__(S)
    fake.alpha_prod_t_sqrt = torch.tensor([float(sch.alphas_cumprod[t]) ** 0.5 for t in ts_list]).view(shp).to(**to)
    fake.beta_prod_t_sqrt = torch.tensor([(1 - float(sch.alphas_cumprod[t])) ** 0.5 for t in ts_list]).view(shp).to(**to)
    sc = [sch.get_scalings_for_boundary_condition_discrete(t) for t in ts_list]
    fake.c_skip = torch.tensor([float(c[0]) for c in sc]).view(shp).to(**to)
    fake.c_out = torch.tensor([float(c[1]) for c in sc]).view(shp).to(**to)
    ts = torch.tensor(ts_list, device=DEV)
    enc = rn(N, 77, cfg.cross_attention_dim)
    xbuf, dbuf = rn(N - 1, 4, 1, h, w), rn(N - 1, 4, 1, h, w)
    step = HipStreamStep(ub, kvb, ts, enc, fake.alpha_prod_t_sqrt, fake.beta_prod_t_sqrt, fake.c_skip, fake.c_out,
                         inject_noise=True, use_graph=use_graph)
    step.load_buffers(xbuf, dbuf)
    rb = ring_buffer_init(N, cfg.window_size, cfg.sink_size)
    for f in range(2 * cfg.window_size + 3):
        x_new, d_new = rn(1, 4, 1, h, w), rn(1, 4, 1, h, w)
        nz = rn(N - 1, 4, 1, h, w)
        # host-driven path (predict_x0_batch, reference :573-623)
        xt, dt = torch.cat((x_new, xbuf), 0), torch.cat((d_new, dbuf), 0)
        o = ua(xt, ts, encoder_hidden_states=enc, temporal_attention_mask=rb[0].half().to(DEV), depth_sample=dt, kv_cache=kva,
               pe_idx=rb[1].to(DEV), update_idx=rb[2].to(DEV))
        x0 = S.scheduler_step_batch(fake, o["sample"], xt)
        ring_buffer_update(*rb, cfg.window_size, cfg.sink_size)
        want_out = x0[-1:].clone()
        xbuf = fake.alpha_prod_t_sqrt[1:] * x0[:-1] + fake.beta_prod_t_sqrt[1:] * nz
        dbuf = dt[:-1].clone()
        # device path
        step.noise.copy_(nz.reshape(-1))
        got = step.step(x_new, d_new)
        torch.cuda.synchronize()
        assert torch.equal(got, want_out), f
        assert torch.equal(step.x_t_latent_buffer, xbuf), f
        assert torch.equal(step.attn_bias.float().cpu(), rb[0]) and torch.equal(step.pe_idx.cpu(), rb[1]), f
        assert torch.equal(step.update_idx.cpu(), rb[2]), f
    for a, b in zip(kva, kvb):
        assert torch.equal(a, b)


class _StubVAE:
    """Caller-owned duck-typed VAE (reference swap point `stream.vae`, engine.py:71-109): 8x average-pool 'encoder' to 4
    channels and nearest-upsample 'decoder' -- enough to drive the pipeline's per-frame path end to end."""
    dtype = torch.float16

    class config:
        scaling_factor = 0.5

    def encode(self, x):
        lat = torch.nn.functional.avg_pool2d(x.float(), 8)
        lat = torch.cat([lat, lat.mean(1, keepdim=True)], 1).to(torch.float16)      # 3 -> 4 channels
        return type("Out", (), {"latents": lat})()

    def decode(self, z, return_dict=False):
        return (torch.nn.functional.interpolate(z[:, :3].float(), scale_factor=8, mode="nearest").to(torch.float16),)


class _StubDepth:
    dtype = torch.float16

    def __call__(self, images):
        return images.float().mean(1).to(torch.float16) + 1.0                          # [B,384,384]


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipeline_frames_device_step_vs_host_path(golden, use_graph):
    """StreamAnimateDiffusionDepth end to end on the HIP UNet (prepare = N warm-up passes, then frames through
    __call__): the opt-in device-side step gives bit-identical frames to the host-driven predict_x0_batch
    (do_add_noise=False: the two paths draw their re-noising tensors from different generators)."""
    from types import SimpleNamespace

    from live2diff_amd.config import tiny_config
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    H = W = 128
    sd = {k: v.to(DEV) for k, v in random_state_dict(cfg, dtype=torch.float16).items()}
    g = torch.Generator().manual_seed(5)
    warm = [torch.rand(3, H, W, generator=g) for _ in range(cfg.sink_size)]
    frames = [torch.rand(1, 3, H, W, generator=g) for _ in range(2 * cfg.window_size)]
    emb = torch.randn(1, 77, 64, generator=g)
    outs = []
    for device_step in (False, True):
        torch.manual_seed(0)        # `prepare` re-noises the warm-up latents from the global generator, like the reference (:337)
        pipe = SimpleNamespace(device=torch.device(DEV), vae_scale_factor=8, unet=HipStreamingUNet(sd, cfg, H // 8, W // 8, 2),
                               vae=_StubVAE(), depth_model=_StubDepth(), scheduler=None)
        s = StreamAnimateDiffusionDepth(pipe, num_inference_steps=50, t_index_list=[30, 40], width=W, height=H, do_add_noise=False,
                                        warmup_frames=cfg.sink_size, window_size=cfg.window_size)
        s.prepare_cache(H, W, 2)
        first = s.prepare(warm, prompt_embeds=emb, seed=3)
        assert torch.isfinite(first).all()
        if device_step:
            s.enable_device_step(use_graph=use_graph)
        res = [s(f.to(DEV)).clone() for f in frames]
        assert all(torch.isfinite(r).all() for r in res)
        outs.append((first, res, [c.clone() for c in s.kv_cache_list], s.update_idx.clone().cpu(), s.pe_idx.clone().cpu()))
    (fa, ra, ka, ua, pa), (fb, rb_, kb, ub, pb) = outs
    assert torch.equal(fa, fb)
    for i, (a, b) in enumerate(zip(ra, rb_)):
        assert torch.equal(a, b), i
    assert all(torch.equal(a, b) for a, b in zip(ka, kb)) and torch.equal(ua, ub) and torch.equal(pa, pb)


def test_device_step_late_enable_prompt_update_and_reprepare(golden):
    """Round-1 advisor findings on the device step: (a) enable_device_step() AFTER frames already ran on the host path must
    continue from the pipeline's current ring-buffer state; (b) update_prompt() must reach the device step's static text
    input; (c) prepare() again must drop the old device step (stale ring state / frame counter); (d) once the device step
    is on, the host path with injected noise is refused instead of running on stale state.  Each case is compared with a
    pipeline that stays on the host path (do_add_noise=False, so both paths are deterministic)."""
    from types import SimpleNamespace

    from live2diff_amd.config import tiny_config
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    H = W = 128
    sd = {k: v.to(DEV) for k, v in random_state_dict(cfg, dtype=torch.float16).items()}
    g = torch.Generator().manual_seed(6)
    warm = [torch.rand(3, H, W, generator=g) for _ in range(cfg.sink_size)]
    frames = [torch.rand(1, 3, H, W, generator=g) for _ in range(30)]
    emb1, emb2 = torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g)

    def make():
        pipe = SimpleNamespace(device=torch.device(DEV), vae_scale_factor=8, unet=HipStreamingUNet(sd, cfg, H // 8, W // 8, 2),
                               vae=_StubVAE(), depth_model=_StubDepth(), scheduler=None)
        pipe._encode_prompt = lambda prompt, **k: ((emb2 if prompt == "two" else emb1).to(DEV),)
        s = StreamAnimateDiffusionDepth(pipe, num_inference_steps=50, t_index_list=[30, 40], width=W, height=H, do_add_noise=False,
                                        warmup_frames=cfg.sink_size, window_size=cfg.window_size)
        s.prepare_cache(H, W, 2)
        return s

    outs = []
    for device in (False, True):
        torch.manual_seed(0)
        s = make()
        s.prepare(warm, prompt_embeds=emb1, seed=3)
        res = [s(f.to(DEV)).clone() for f in frames[:11]]            # 11 frames on the host path (rolling part mid-cycle)
        if device:
            s.enable_device_step()                                    # (a) late enable
        res += [s(f.to(DEV)).clone() for f in frames[11:18]]
        s.update_prompt("two")                                        # (b)
        res += [s(f.to(DEV)).clone() for f in frames[18:24]]
        if device:
            with pytest.raises(ValueError):                           # (d)
                s.predict_x0_batch(torch.zeros(1, 4, 1, H // 8, W // 8, device=DEV, dtype=torch.float16),
                                   torch.zeros(1, 4, 1, H // 8, W // 8, device=DEV, dtype=torch.float16),
                                   noise=torch.zeros(1, 4, 1, H // 8, W // 8, device=DEV, dtype=torch.float16))
        torch.manual_seed(0)
        s.prepare_cache(H, W, 2)
        s.prepare(warm, prompt_embeds=emb1, seed=3)                   # (c) a new stream on the same object
        assert getattr(s, "_device_step", None) is None
        if device:
            s.enable_device_step()
        res += [s(f.to(DEV)).clone() for f in frames[24:30]]
        outs.append(res)
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), f"frame {i}: device-step pipeline diverged from the host-path pipeline"
    assert not torch.equal(outs[0][17], outs[0][18])
