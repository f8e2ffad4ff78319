"""-m gpu: the depth detector (DPT-Hybrid) through the C ABI against the fp32 oracle
(oracle/midas_ref.py; third-party topology: parity unpinned) and torch references of the single ops it added.

Tolerances (fp16 storage, fp32 accumulate): single ops rel-L2 <= 2e-3 (bit-exact for pooling / subsampling).  For the
122 M-parameter network the allowance is tied to the noise floor of fp16 storage itself: with random weights the ResNetV2
stages amplify rounding (GroupNorm over few elements), so each stage tap and the final inverse depth must be within
max(1e-2, 2 x d16), where d16 = rel-L2 between the fp32 oracle and the same oracle with conv / GroupNorm outputs rounded to
fp16 (the op-by-op fp16 graph the reference runs); cosine of the depth map >= 0.999."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def cos(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm())).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.fixture(scope="module")
def L():
    from live2diff_amd import _lib, ops
    print("device:", _lib.device_name())
    return ops


@pytest.mark.parametrize("B,H,W", [(1, 64, 64), (2, 96, 160), (1, 384, 384)])
def test_stem7x7(L, B, H, W):
    from oracle.midas_ref import same_pad
    x = rnd(B, 3, H, W, seed=1)
    w = rnd(64, 3, 7, 7, seed=2, scale=0.1)
    pt, pb = same_pad(H, 7, 2)
    pl, pr = same_pad(W, 7, 2)
    ref = F.conv2d(F.pad(x.float(), (pl, pr, pt, pb)), w.float(), stride=2)
    out = torch.zeros(B, H // 2, W // 2, 64, dtype=torch.float16, device=DEV)
    L.run(L.stem7x7(x.to(DEV), w.to(DEV), out, B=B, H=H, W=W))
    torch.cuda.synchronize()
    assert rel(out.permute(0, 3, 1, 2), ref) <= 2e-3


@pytest.mark.parametrize("B,H,W,C", [(1, 32, 32, 64), (2, 24, 40, 256), (1, 96, 96, 256)])
def test_resample_modes(L, B, H, W, C):
    x = rnd(B, C, H, W, seed=3)
    xn = nhwc(x).to(DEV)
    # max pool 3x3 stride 2, TF-SAME (pad 0 low / 1 high for even sizes, -inf padding)
    o = torch.zeros(B, H // 2, W // 2, C, dtype=torch.float16, device=DEV)
    L.run(L.resample_nhwc(xn, o, B=B, H=H, W=W, C=C, mode=L.RS_MAXPOOL))
    torch.cuda.synchronize()
    ref = F.max_pool2d(F.pad(x.float(), (0, 1, 0, 1), value=float("-inf")), 3, 2)
    assert torch.equal(o.permute(0, 3, 1, 2).float().cpu(), ref)
    # stride-2 subsample (1x1 stride-2 SAME conv = subsample then 1x1)
    L.run(L.resample_nhwc(xn, o, B=B, H=H, W=W, C=C, mode=L.RS_SUBSAMPLE))
    torch.cuda.synchronize()
    assert torch.equal(o.permute(0, 3, 1, 2).cpu(), x[:, :, ::2, ::2])
    # bilinear x2, align_corners=True
    o2 = torch.zeros(B, 2 * H, 2 * W, C, dtype=torch.float16, device=DEV)
    L.run(L.resample_nhwc(xn, o2, B=B, H=H, W=W, C=C, mode=L.RS_UP2X))
    torch.cuda.synchronize()
    ref = F.interpolate(x.float(), scale_factor=2, mode="bilinear", align_corners=True)
    assert rel(o2.permute(0, 3, 1, 2), ref) <= 1e-3


def test_elementwise_add_relu(L):
    n = 3 * 1000 * 8
    a, b = rnd(n, seed=1).to(DEV), rnd(n, seed=2).to(DEV)
    s, r = torch.zeros_like(a), torch.zeros_like(a)
    L.run(L.ew(a, b, s, r, n=n))
    torch.cuda.synchronize()
    assert torch.equal(s, a + b) and torch.equal(r, F.relu(a + b))
    r2 = torch.zeros_like(a)
    L.run(L.ew(a, None, None, r2, n=n))
    torch.cuda.synchronize()
    assert torch.equal(r2, F.relu(a))


@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 32, 32, 64, 64), (2, 24, 40, 128, 128), (1, 96, 96, 128, 128)])
def test_igemm_stride2_same_padding(L, B, H, W, cin, cout):
    """the ResNetV2 stride-2 3x3 convs pad (0, 1), not (1, 1)"""
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), stride=2)
    xn = nhwc(x).reshape(-1, cin).to(DEV)
    wp = L.pack_conv3x3(w.to(DEV))
    Ho, Wo = H // 2, W // 2
    M = B * Ho * Wo
    out = torch.zeros(M, cout, dtype=torch.float16, device=DEV)
    tile, S, variant = L.igemm_schedule(M, cout, wp.shape[1], 1, 0, 9)
    ws = torch.zeros(S * M * cout, dtype=torch.float32, device=DEV) if S > 1 else None
    L.run(L.igemm(xn, wp, out, M=M, Nout=cout, C1=cin, ldx1=cin, CinP=wp.shape[1] // 9, ldo=cout, taps=9, B=B, Hin=H, Win=W, Hout=Ho,
                  Wout=Wo, stride=2, tile=tile, splitk=S, ws=ws, variant=variant, pad_same=True))
    torch.cuda.synchronize()
    assert rel(out.reshape(B, Ho, Wo, cout).permute(0, 3, 1, 2), ref) <= 2e-3


@pytest.mark.parametrize("M,K,N,rowbias", [(577, 768, 3072, False), (576, 768, 768, True), (100, 64, 4, False)])
def test_igemm_gelu_epilogue(L, M, K, N, rowbias):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3).float()
    rb = rnd(N, seed=4).float()
    ref = F.gelu(F.linear(x.float(), w.float(), b) + (rb if rowbias else 0))
    wp = L.pack_linear(w.to(DEV))
    out = torch.zeros(M, max(4, N), dtype=torch.float16, device=DEV)
    tile, S, variant = L.igemm_schedule(M, N, wp.shape[1], 1, 5, 1)
    ws = torch.zeros(S * M * N, dtype=torch.float32, device=DEV) if S > 1 else None
    L.run(L.igemm(x.to(DEV), wp, out, M=M, Nout=N, C1=K, ldx1=K, CinP=wp.shape[1], ldo=max(4, N), bias=b.to(DEV), epi=5, tile=tile, splitk=S,
                  ws=ws, variant=variant, rowbias=(rb.to(DEV) if rowbias else None), ldrb=N, rows_per_bias=(M if rowbias else 0)))
    torch.cuda.synchronize()
    assert rel(out[:, :N], ref) <= 2e-3


@pytest.mark.parametrize("act", [0, 2, 3])
def test_gn_apply_relu_and_residual(L, act):
    B, T, C, G = 2, 2304, 256, 32
    x, r = rnd(B, T, C, seed=1, scale=3.0), rnd(B, T, C, seed=2)
    g, be = rnd(C, seed=3) * 0.1 + 1, rnd(C, seed=4) * 0.1
    ref = F.group_norm(x.float().transpose(1, 2), G, g.float(), be.float(), 1e-5).transpose(1, 2)
    if act == 2:
        ref = F.relu(ref)
    if act == 3:
        ref = F.relu(ref.half().float() + r.float())
    nchunk = 64
    partial = torch.zeros(B * nchunk * G * 2, dtype=torch.float32, device=DEV)
    out = torch.zeros(B, T, C, dtype=torch.float16, device=DEV)
    kw = dict(B=B, T=T, C1=C, ld1=C, G=G, nchunk=nchunk)
    xd = x.to(DEV)
    L.run(L.gn_stats(xd, partial, **kw))
    L.run(L.gn_apply(xd, partial, g.to(DEV), be.to(DEV), out, eps=1e-5, silu=act, res=(r.to(DEV) if act == 3 else None), **kw))
    torch.cuda.synchronize()
    assert rel(out, ref) <= 2e-3


def test_skinny_linear_strided_rows(L):
    """the class-token rows of a [B, T, C] token buffer (row stride T*C) as the A operand"""
    B, T, C, N = 3, 577, 768, 768
    tok = rnd(B, T, C, seed=1)
    w, b = rnd(N, C, seed=2, scale=C ** -0.5), rnd(N, seed=3).float()
    ref = F.linear(tok[:, 0].float(), w.float(), b)
    out = torch.zeros(B, N, dtype=torch.float32, device=DEV)
    L.run(L.skinny_linear(tok.to(DEV), w.to(DEV), b.to(DEV), out, M=B, K=C, Nout=N, lda=T * C))
    torch.cuda.synchronize()
    assert rel(out, ref) <= 2e-3


@pytest.mark.parametrize("B,T", [(1, 577), (2, 577), (1, 65)])
def test_flash_attn_vit_heads(L, B, T):
    """12 heads of 64 over 1 + 24 x 24 tokens (ragged T: neither a multiple of the key tile nor of 8)"""
    H, d = 12, 64
    C = H * d
    qk = rnd(B, T, 2 * C, seed=1)
    v = rnd(B, T, C, seed=2)
    q, k = qk[..., :C].float().reshape(B, T, H, d).transpose(1, 2), qk[..., C:].float().reshape(B, T, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(q, k, v.float().reshape(B, T, H, d).transpose(1, 2)).transpose(1, 2).reshape(B, T, C)
    ldvt = L.round_up(T, 8)
    vt = torch.zeros(B, C, ldvt, dtype=torch.float16, device=DEV)
    vt[:, :, :T] = v.transpose(1, 2).to(DEV)
    out = torch.zeros(B, T, C, dtype=torch.float16, device=DEV)
    L.run(L.flash_attn(qk.to(DEV), qk.to(DEV), vt, out, B=B, H=H, d=d, Tq=T, Tk=T, ldq=2 * C, ldk=2 * C, ldvt=ldvt, ldo=C, sq=T * 2 * C,
                       sk=T * 2 * C, svt=C * ldvt, so=T * C, k_off=C))
    torch.cuda.synchronize()
    assert rel(out, ref) <= 2e-3


# ----------------------------------------------------------------------------- the network
def _compare(img, B, seed=0):
    from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict
    from oracle import midas_ref as M
    sd = random_midas_state_dict(dtype=torch.float16, img=img)           # both sides see the fp16-rounded weights
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, img, img, generator=g).clamp(-2.5, 2.5).to(torch.float16)
    taps = {}
    sd32 = {k: v.float() for k, v in sd.items()}
    ref = M.midas_forward(x.float(), sd32, taps)
    taps16 = {}
    with M.fp16_activations():
        taps16["out"] = M.midas_forward(x.float(), sd32, taps16)
    taps["out"] = ref
    floor = {k: rel(taps16[k], taps[k]) for k in taps}
    m = HipMidas(sd, device=DEV, img=img, debug_taps=True)
    got = m(x.to(DEV)).clone()
    got2 = m(x.to(DEV))                                                   # plan replay: same buffers, same result
    torch.cuda.synchronize()
    assert torch.equal(got, got2)
    st = m._plans[(B, img, img)]
    report = {}
    for name, t in st.taps.items():
        r = taps[name]
        r = r if name.startswith("vit") else r.permute(0, 2, 3, 1)
        report[name] = rel(t, r)
    report["out"] = rel(got, ref)
    print({k: f"{v:.2e} (fp16 floor {floor[k]:.2e})" for k, v in report.items()})
    bad = {k: (v, floor[k]) for k, v in report.items() if v > max(1e-2, 2 * floor[k])}
    assert not bad, (bad, report)
    assert cos(got, ref) >= 0.999 and (got >= 0).all()
    return m


def test_midas_small_input_against_oracle():
    """128 x 128 (8 x 8 patch grid): every stage tap and the inverse depth, batch 2"""
    _compare(128, 2)


def test_midas_reference_size_against_oracle():
    """the reference's 384 x 384 call (pipeline_stream_animation_depth.py:553-558)"""
    m = _compare(384, 1, seed=1)
    s = m.plan_summary()[(1, 384, 384)]
    assert s["gn_fused"] >= 42


def test_pipeline_encode_depth_with_hip_detector():
    """encode_depth with HipMidas in the `stream.depth_detector` slot equals the same glue fed by the oracle network"""
    from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict
    from live2diff_amd.vae_hip import HipDepthGlue
    from oracle import midas_ref as M
    sd = random_midas_state_dict(dtype=torch.float16)
    det = HipMidas(sd, device=DEV)
    glue = HipDepthGlue(DEV)
    g = torch.Generator().manual_seed(5)
    frames = (torch.rand(1, 3, 512, 512, generator=g) * 2 - 1).to(torch.float16).to(DEV)
    x384 = glue.resize(frames, 384, 384)
    depth = det(x384)
    out = glue.normalize_resize(depth, 512, 512)
    sd32 = {k: v.float() for k, v in sd.items()}

    def glue_ref(d):
        lo, hi = d.amin(), d.amax()
        return F.interpolate(((d - lo) / (hi - lo))[:, None].repeat(1, 3, 1, 1) * 2 - 1, size=(512, 512), mode="bilinear", align_corners=False)
    ref = glue_ref(M.midas_forward(x384.float().cpu(), sd32))
    with M.fp16_activations():
        floor = rel(glue_ref(M.midas_forward(x384.float().cpu(), sd32)), ref)
    torch.cuda.synchronize()
    assert out.shape == (1, 3, 512, 512) and rel(out, ref) <= max(1e-2, 2 * floor), (rel(out, ref), floor)


def test_whole_pipeline_on_hip_backends():
    """Every per-frame model slot of StreamAnimateDiffusionDepth filled by this repo (`stream.unet` HipStreamingUNet,
    `stream.vae` HipTinyVAE, `stream.depth_detector` HipMidas, HIP depth glue), host-driven frame step and device step:
    `prepare` over the warm-up window + frames run, stay finite, and the two stepping modes agree bit for bit (no re-noising,
    so neither path draws random numbers after `prepare`)."""
    from types import SimpleNamespace

    from live2diff_amd.config import tiny_config
    from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.vae_hip import HipTinyVAE, random_taesd_state_dict
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    H = W = 128
    sd = {k: v.to(DEV) for k, v in random_state_dict(cfg, dtype=torch.float16).items()}
    vae = HipTinyVAE(random_taesd_state_dict(), device=DEV)
    det = HipMidas(random_midas_state_dict(), device=DEV)
    g = torch.Generator().manual_seed(8)
    warm = [torch.rand(3, H, W, generator=g) for _ in range(cfg.sink_size)]
    frames = [torch.rand(1, 3, H, W, generator=g) for _ in range(5)]
    emb = torch.randn(1, 77, 64, generator=g)
    outs = []
    for device_step in (False, True):
        torch.manual_seed(0)
        pipe = SimpleNamespace(device=torch.device(DEV), vae_scale_factor=8, unet=HipStreamingUNet(sd, cfg, H // 8, W // 8, 2), vae=vae,
                               depth_model=det, scheduler=None)
        s = StreamAnimateDiffusionDepth(pipe, num_inference_steps=50, t_index_list=[30, 40], width=W, height=H, do_add_noise=False,
                                        warmup_frames=cfg.sink_size, window_size=cfg.window_size)
        s.prepare_cache(H, W, 2)
        first = s.prepare(warm, prompt_embeds=emb, seed=3)
        if device_step:
            s.enable_device_step()
        res = [s(f.to(DEV)).clone() for f in frames]
        assert first.shape == (cfg.sink_size, 3, H, W) and torch.isfinite(first).all()
        assert all(r.shape == (1, 3, H, W) and torch.isfinite(r).all() for r in res)
        assert float(torch.stack(res).std()) > 1e-3                           # the frames are not a constant image
        outs.append([first] + res)
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), f"frame {i}: host-driven step and device step differ"
    assert det.plan_summary().keys() >= {(1, 384, 384), (cfg.sink_size, 384, 384)}      # per-frame call and the warm-up batch


def test_pipelined_push_pop_equals_call():
    """Opt-in frame pipelining (`enable_frame_pipelining`, push / pop): the work in front of the UNet runs on a second HIP stream,
    one frame ahead of the UNet step and decode on the caller's stream.  Same frames, same seeds: the outputs of `pop()` are
    bit-identical to those of `__call__` (re-noising on, so the order of the random draws matters too), whatever the
    push-ahead depth."""
    from types import SimpleNamespace

    from live2diff_amd.config import tiny_config
    from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.vae_hip import HipTinyVAE, random_taesd_state_dict
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    H = W = 128
    sd = {k: v.to(DEV) for k, v in random_state_dict(cfg, dtype=torch.float16).items()}
    vae = HipTinyVAE(random_taesd_state_dict(), device=DEV)
    det = HipMidas(random_midas_state_dict(), device=DEV)
    g = torch.Generator().manual_seed(9)
    warm = [torch.rand(3, H, W, generator=g) for _ in range(cfg.sink_size)]
    frames = [torch.rand(1, 3, H, W, generator=g).to(DEV) for _ in range(7)]
    emb = torch.randn(1, 77, 64, generator=g)

    def build():
        torch.manual_seed(0)
        pipe = SimpleNamespace(device=torch.device(DEV), vae_scale_factor=8, unet=HipStreamingUNet(sd, cfg, H // 8, W // 8, 2), vae=vae,
                               depth_model=det, scheduler=None)
        s = StreamAnimateDiffusionDepth(pipe, num_inference_steps=50, t_index_list=[30, 40], width=W, height=H, do_add_noise=True,
                                        warmup_frames=cfg.sink_size, window_size=cfg.window_size)
        s.image_processor.assume_unit_range = True
        s.prepare_cache(H, W, 2)
        s.prepare(warm, prompt_embeds=emb, seed=3)
        s.enable_device_step(seed=5)
        return s

    s = build()
    want = [s(f).clone() for f in frames]
    assert all(torch.isfinite(r).all() for r in want) and not torch.equal(want[1], want[2])
    for ahead in (1, 2):
        s = build()
        with pytest.raises(ValueError):
            StreamAnimateDiffusionDepth.enable_frame_pipelining(SimpleNamespace(_device_step=None))
        s.enable_frame_pipelining()
        got = []
        for i in range(min(ahead, len(frames))):
            s.push(frames[i])
        for i in range(len(frames)):
            if i + ahead < len(frames):
                s.push(frames[i + ahead])
            got.append(s.pop().clone())
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), f"push-ahead {ahead}, frame {i}: pipelined output differs from __call__ ({rel(a, b):.3e})"
    with pytest.raises(RuntimeError):
        s.pop()
