"""-m gpu: the tiny VAE and the depth-path glue (row F2, the part around the detector) through the C ABI,
against the fp32 oracle (oracle/taesd_ref.py; third-party topology: parity unpinned) and torch references of single ops.

Tolerances (fp16 storage, fp32 accumulate): single ops rel-L2 <= 2e-3; the 37-conv encoder / decoder rel-L2 <= 1e-2,
cosine >= 0.9995."""
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


@pytest.fixture(scope="module")
def L():
    from live2diff_amd import _lib, ops
    print("device:", _lib.device_name())
    return ops


@pytest.mark.parametrize("B,H,W,cin,cout,epi", [(1, 32, 32, 64, 64, 3), (1, 32, 32, 64, 64, 4), (2, 24, 40, 64, 64, 4),
                                                (1, 32, 32, 64, 3, 0), (1, 16, 16, 8, 64, 3), (1, 128, 128, 64, 64, 4)])
def test_igemm_conv_relu_epilogues(L, B, H, W, cin, cout, epi):
    """epi 3: relu(conv + b); epi 4: relu(conv + b + skip) (the TAESD block); the 3-channel image conv takes the direct
    (register -> global) epilogue with rows stored 4 wide, the 64-channel ones the LDS-staged one."""
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    b = rnd(cout, seed=3).float()
    r = rnd(B, cout, H, W, seed=4)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    if epi == 3:
        ref = F.relu(ref)
    if epi == 4:
        ref = F.relu(ref.half().float() + r.float())
    xn = x.permute(0, 2, 3, 1).reshape(B * H * W, cin).contiguous().to(DEV)
    rn_ = r.permute(0, 2, 3, 1).reshape(B * H * W, cout).contiguous().to(DEV)
    wp = L.pack_conv3x3(w.to(DEV))
    ldo = max(4, cout)
    out = torch.zeros(B * H * W, ldo, dtype=torch.float16, device=DEV)
    tile, S, variant = L.igemm_schedule(B * H * W, cout, wp.shape[1], 1, epi, 9)
    L.run(L.igemm(xn, wp, out, M=B * H * W, Nout=cout, C1=cin, ldx1=cin, CinP=wp.shape[1] // 9, ldo=ldo, bias=b.to(DEV),
                  res=(rn_ if epi == 4 else None), ldr=(cout if epi == 4 else 0), taps=9, B=B, Hin=H, Win=W, Hout=H, Wout=W,
                  epi=epi, tile=tile, variant=variant))
    torch.cuda.synchronize()
    got = out[:, :cout].reshape(B, H, W, cout).permute(0, 3, 1, 2)
    assert rel(got, ref) <= 2e-3, rel(got, ref)


def test_layout_maps(L):
    x = rnd(2, 3, 50, seed=1)
    for mode, a, b, fn in [(L.MAP_ADD_SCALE, 0.5, 1.0, lambda t: (t + 1) / 2), (L.MAP_TANH3, 1.0, 0.0, lambda t: torch.tanh(t / 3) * 3),
                           (L.MAP_SCALE_ADD, 2.0, -1.0, lambda t: t * 2 - 1)]:
        o = torch.empty(2, 50, 8, dtype=torch.float16, device=DEV)
        L.run(L.nchw_to_nhwc(x.to(DEV), o, B=2, C=3, HW=50, Cpad=8, mode=mode, a=a, b=b))
        torch.cuda.synchronize()
        ref = fn(x.float()).transpose(1, 2)
        assert (o[:, :, :3].float().cpu() - ref).abs().max() <= 4e-3 and (o[:, :, 3:] == 0).all()
        y = rnd(2, 50, 4, seed=2)
        o2 = torch.empty(2, 3, 50, dtype=torch.float16, device=DEV)
        L.run(L.nhwc_to_nchw(y.to(DEV), o2, B=2, C=3, HW=50, ld=4, mode=mode, a=a, b=b))
        torch.cuda.synchronize()
        assert (o2.float().cpu() - fn(y[:, :, :3].float()).transpose(1, 2)).abs().max() <= 4e-3


@pytest.mark.parametrize("B,C,Hin,Win,Hout,Wout", [(1, 3, 512, 512, 384, 384), (2, 3, 384, 384, 512, 768), (1, 1, 100, 60, 37, 91),
                                                   (1, 2, 64, 64, 64, 64), (1, 3, 576, 1024, 384, 384), (1, 1, 5, 7, 40, 3)])
def test_resize_bilinear(B, C, Hin, Win, Hout, Wout):
    from live2diff_amd.vae_hip import HipDepthGlue
    x = rnd(B, C, Hin, Win, seed=5)
    got = HipDepthGlue(DEV).resize(x.to(DEV), Hout, Wout)
    torch.cuda.synchronize()
    ref = F.interpolate(x.float(), (Hout, Wout), mode="bilinear", align_corners=False)
    assert got.shape == ref.shape and rel(got, ref) <= 1e-3, rel(got, ref)
    assert (got.float().cpu() - ref).abs().max() <= 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,Hd,Wd,H,W", [(1, 384, 384, 512, 512), (8, 384, 384, 64, 96), (1, 384, 384, 576, 1024), (2, 33, 47, 20, 70)])
def test_depth_normalize_resize(B, Hd, Wd, H, W):
    """reference pipeline_stream_animation_depth.py:560-567 (min-max over the WHOLE batch, 3 channels, [-1,1], bilinear)."""
    from live2diff_amd.vae_hip import HipDepthGlue
    from oracle.taesd_ref import depth_glue
    d = (rnd(B, Hd, Wd, seed=6).float() * 3 + 10).half()           # MiDaS-like inverse depth: positive, wide range
    got = HipDepthGlue(DEV).normalize_resize(d.to(DEV), H, W)
    torch.cuda.synchronize()
    ref = depth_glue(d.float(), H, W)
    assert got.shape == (B, 3, H, W)
    assert torch.equal(got[:, 0], got[:, 1]) and torch.equal(got[:, 0], got[:, 2])
    assert (got.float().cpu() - ref).abs().max() <= 6e-3, (got.float().cpu() - ref).abs().max()     # fp16 steps of a [-1,1] quantity
    assert got.min() >= -1.001 and got.max() <= 1.001


@pytest.fixture(scope="module")
def vae():
    from live2diff_amd.vae_hip import HipTinyVAE, random_taesd_state_dict
    sd = random_taesd_state_dict()
    return HipTinyVAE(sd, device=DEV), {k: v.float() for k, v in sd.items()}


@pytest.mark.parametrize("B,H,W", [(1, 64, 64), (2, 128, 96), (8, 64, 64), (1, 512, 512)])
def test_taesd_encode_decode_vs_oracle(vae, B, H, W):
    """encode and decode against the fp32 oracle on the same fp16-rounded weights and inputs, incl. the frame size of
    BASELINE configs[1] (512x512) and the 8-frame warm-up batch of `prepare`."""
    from oracle import taesd_ref as T
    v, sd32 = vae
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).half()
    lat = v.encode(x.to(DEV)).latents.clone()
    torch.cuda.synchronize()
    ref = T.taesd_encode(x.float(), sd32)
    assert lat.shape == ref.shape == (B, 4, H // 8, W // 8)
    print(f"encode {B}x{H}x{W}: rel-L2 {rel(lat, ref):.3e} cos {cos(lat, ref):.6f}")
    assert torch.isfinite(lat).all() and rel(lat, ref) <= 1e-2 and cos(lat, ref) >= 0.9995
    z = (torch.randn(B, 4, H // 8, W // 8, generator=g) * 1.5).half()
    img = v.decode(z.to(DEV), return_dict=False)[0].clone()
    torch.cuda.synchronize()
    ref = T.taesd_decode(z.float(), sd32)
    assert img.shape == ref.shape == (B, 3, H, W)
    print(f"decode {B}x{H}x{W}: rel-L2 {rel(img, ref):.3e} cos {cos(img, ref):.6f}")
    assert torch.isfinite(img).all() and rel(img, ref) <= 1e-2 and cos(img, ref) >= 0.9995
    # same call again: static plan, bit-identical
    assert torch.equal(v.decode(z.to(DEV), return_dict=False)[0], img)


def test_pipeline_runs_on_hip_vae_and_glue(golden):
    """StreamAnimateDiffusionDepth with `stream.vae` = HipTinyVAE and the HIP depth glue (caller-owned stub depth detector):
    `prepare` + frames end to end; the same pipeline with the oracle VAE / torch glue in the vae / glue slots gives the same
    frames within the VAE tolerance (the UNet in between is the same HIP object)."""
    from types import SimpleNamespace

    from live2diff_amd.config import tiny_config
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.vae_hip import HipTinyVAE, random_taesd_state_dict
    from live2diff_amd.weights import random_state_dict
    from oracle import taesd_ref as T
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    H = W = 128
    sd = {k: v.to(DEV) for k, v in random_state_dict(cfg, dtype=torch.float16).items()}
    vsd = random_taesd_state_dict()
    vsd32 = {k: v.float() for k, v in vsd.items()}

    class OracleVAE:                    # test infrastructure (CPU fp32) in the caller-owned `stream.vae` slot
        dtype = torch.float16
        config = SimpleNamespace(scaling_factor=1.0)

        def encode(self, x):
            return SimpleNamespace(latents=T.taesd_encode(x.float().cpu(), vsd32).half().to(DEV))

        def decode(self, z, return_dict=False):
            return (T.taesd_decode(z.float().cpu(), vsd32).half().to(DEV),)

    class StubDepth:
        dtype = torch.float16

        def __call__(self, images):
            return (images.float().mean(1) * 4 + 9).to(torch.float16)

    g = torch.Generator().manual_seed(8)
    warm = [torch.rand(3, H, W, generator=g) for _ in range(cfg.sink_size)]
    frames = [torch.rand(1, 3, H, W, generator=g) for _ in range(6)]
    emb = torch.randn(1, 77, 64, generator=g)
    outs = []
    for hip in (True, False):
        torch.manual_seed(0)
        pipe = SimpleNamespace(device=torch.device(DEV), vae_scale_factor=8, unet=HipStreamingUNet(sd, cfg, H // 8, W // 8, 2),
                               vae=(HipTinyVAE(vsd, device=DEV) if hip else OracleVAE()), depth_model=StubDepth(), scheduler=None)
        s = StreamAnimateDiffusionDepth(pipe, num_inference_steps=50, t_index_list=[30, 40], width=W, height=H, do_add_noise=False,
                                        warmup_frames=cfg.sink_size, window_size=cfg.window_size)
        if not hip:
            s.depth_glue = None         # torch glue (the reference's own ops) on the oracle side
        s.prepare_cache(H, W, 2)
        first = s.prepare(warm, prompt_embeds=emb, seed=3)
        res = [s(f.to(DEV)).clone() for f in frames]
        assert torch.isfinite(first).all() and all(torch.isfinite(r).all() for r in res)
        outs.append([first] + res)
    for i, (a, b) in enumerate(zip(*outs)):
        assert a.shape == b.shape
        print(f"frame {i}: rel-L2 {rel(a, b):.3e}")
        assert rel(a, b) <= 3e-2 and cos(a, b) >= 0.999, (i, rel(a, b))
