"""-m gpu: the patch-resident / register-streamed 3x3 convolution (csrc/cconv.hip, L2D_OP_CCONV) through the C ABI against F.conv2d
in fp32 on the same fp16-rounded inputs: every block geometry (channel tiles x K groups x loader waves), split-K with the slabs
NaN-poisoned, channel concat of two inputs, zero padding at all four image borders, the nearest-x2 up-sampling of Upsample3D folded
into the gather (reference resnet.py:94-127), bias + per-sample time-embedding bias + residual, GroupNorm statistics of the
output, bit-repeatability.  Tolerance: rel-L2 <= 2e-3 (fp16 storage, fp32 accumulate), SURVEY.md section 8c."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16)


@pytest.fixture(scope="module")
def L():
    from live2diff_amd import _lib, ops
    print("device:", _lib.device_name())
    return ops


def _bufs(L, B, H, W, N, CG, S):
    ws = cnt = None
    if S > 1:
        n_ws, n_cnt = L.cconv_sizes(B, H, W, N, CG, S)
        ws = torch.full((n_ws,), float("nan"), dtype=torch.float32, device=DEV)      # a slab read too early / from another launch is NaN
        cnt = torch.zeros(n_cnt + 3, dtype=torch.int32, device=DEV)
    return ws, cnt


@pytest.mark.parametrize("B,H,W,C1,C2,N,sched", [
    (2, 32, 32, 640, 0, 640, (2, 2, 1, 3)),       # level 1 resnet conv (cfg-2): 80 tiles x 3 slices
    (2, 32, 32, 640, 0, 640, (2, 2, 2, 1)),
    (2, 32, 32, 640, 320, 640, (2, 2, 1, 3)),     # two inputs (channel concat)
    (2, 16, 16, 1280, 0, 1280, (1, 4, 2, 3)),     # level 2
    (2, 16, 16, 1280, 0, 1280, (1, 4, 1, 5)),
    (2, 16, 16, 1280, 0, 1280, (2, 2, 1, 6)),     # ragged slices: 20 chunks over 6
    (2, 64, 64, 320, 0, 320, (1, 4, 2, 1)),       # level 0
    (2, 16, 16, 256, 0, 256, (2, 2, 4, 2)),       # four loader waves
    (2, 16, 16, 1280, 0, 1280, (1, 4, 4, 3)),
    (2, 32, 32, 64, 0, 128, (2, 2, 4, 1)),        # one chunk: the loaders' short path
    (2, 32, 32, 128, 0, 128, (1, 4, 2, 1)),       # two chunks
    (1, 8, 16, 64, 0, 64, (1, 4, 1, 1)),          # one patch: every pixel is a border pixel somewhere
    (3, 8, 16, 64, 64, 128, (2, 2, 1, 2)),
    (8, 16, 16, 128, 0, 128, (2, 2, 2, 1)),       # warm-up style batch
    (2, 64, 96, 320, 0, 320, (1, 4, 2, 1)),       # non-square (cfg-3 aspect)
])
def test_cconv_matches_conv2d(L, B, H, W, C1, C2, N, sched):
    CG, KG, NLD, S = sched
    C = C1 + C2
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, 3, 3, seed=2, scale=(9 * C) ** -0.5)
    b = rnd(N, seed=3).float()
    temb = rnd(B, 2 * N, seed=4).float()               # per-sample row bias, this conv's columns start at offset N
    r = rnd(B * H * W, N, seed=5)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1) + temb[:, N:, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, N).half().float() + r.float()
    wp = L.pack_cconv(w.to(DEV), KG)
    assert torch.equal(L.unpack_cconv(wp, N, C, KG).cpu(), w.permute(0, 2, 3, 1).reshape(N, 9, C))
    xd = x.to(DEV)
    x1 = xd[..., :C1].contiguous()
    x2 = xd[..., C1:].contiguous() if C2 else None
    out = torch.zeros(B * H * W, N, dtype=torch.float16, device=DEV)
    tb = temb.to(DEV)
    ws, cnt = _bufs(L, B, H, W, N, CG, S)
    outs = []
    for rep in range(2):
        out.zero_()
        if ws is not None:
            ws.fill_(float("nan"))
        op, keep = L.cconv(x1, wp, out, B=B, H=H, W=W, C1=C1, ldx1=C1, Nout=N, ldo=N, KG=KG, x2=x2, C2=C2, ldx2=C2, bias=b.to(DEV),
                           rowbias=tb, ldrb=2 * N, rows_per_bias=H * W, res=r.to(DEV), ldr=N, sched=sched, ws=ws, cnt=cnt, cnt_off=3)
        op.p[4] = tb.data_ptr() + 4 * N
        L.run((op, keep))
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        e = relerr(out, ref)
        assert e <= 2e-3, f"cconv {B}x{H}x{W} C{C1}+{C2}->{N} sched {sched}: rel-L2 {e:.3e}"
        if cnt is not None:
            assert int(cnt.abs().sum()) == 0            # the arrival counters are back at zero
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1])                # fixed summation order: bit-repeatable
    # one bias row for all samples (warm-up pass), no residual, no bias, default schedule
    KG2 = L.cconv_schedule(B, H, W, N, C)[1]
    wp2 = wp if KG2 == KG else L.pack_cconv(w.to(DEV), KG2)
    sch2 = L.cconv_schedule(B, H, W, N, C)
    ws2, cnt2 = _bufs(L, B, H, W, N, sch2[0], sch2[3])
    out2 = torch.zeros_like(out)
    L.run(L.cconv(x1, wp2, out2, B=B, H=H, W=W, C1=C1, ldx1=C1, Nout=N, ldo=N, KG=KG2, x2=x2, C2=C2, ldx2=C2,
                  rowbias=tb, ldrb=2 * N, rows_per_bias=B * H * W, ws=ws2, cnt=cnt2))
    torch.cuda.synchronize()
    ref2 = (F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, padding=1) + temb[:1, :N, None, None]).permute(0, 2, 3, 1).reshape(B * H * W, N)
    assert relerr(out2, ref2) <= 2e-3


@pytest.mark.parametrize("B,H,W,C,N,sched", [
    (2, 64, 64, 640, 640, (2, 2, 1, 1)),          # up-sampler of level 1 -> 0 (cfg-2): input 32 x 32
    (2, 32, 32, 1280, 1280, (2, 2, 1, 3)),        # level 2 -> 1
    (1, 16, 32, 64, 128, (1, 4, 2, 1)),
])
def test_cconv_upsample_matches_interpolate_conv2d(L, B, H, W, C, N, sched):
    """Upsample3D: F.interpolate(scale 2, nearest) on (h, w), then the 3x3 conv (reference resnet.py:112-124) -- here one launch
    that reads the low-resolution tensor."""
    CG, KG, NLD, S = sched
    x = rnd(B, H // 2, W // 2, C, seed=31)
    w = rnd(N, C, 3, 3, seed=32, scale=(9 * C) ** -0.5)
    b = rnd(N, seed=33).float()
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    ref = F.conv2d(up, w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(B * H * W, N)
    out = torch.zeros(B * H * W, N, dtype=torch.float16, device=DEV)
    ws, cnt = _bufs(L, B, H, W, N, CG, S)
    L.run(L.cconv(x.to(DEV), L.pack_cconv(w.to(DEV), KG), out, B=B, H=H, W=W, C1=C, ldx1=C, Nout=N, ldo=N, KG=KG, ups=1, bias=b.to(DEV),
                  sched=sched, ws=ws, cnt=cnt, cnt_off=3))
    torch.cuda.synchronize()
    e = relerr(out, ref)
    assert e <= 2e-3, f"cconv upsample {B}x{H}x{W} C{C}->{N} sched {sched}: rel-L2 {e:.3e}"


def test_cconv_agrees_with_igemm_and_feeds_groupnorm(L):
    """Same operands through cconv and the implicit-GEMM kernel; the output's GroupNorm statistics (two consumers) from cconv's
    epilogue equal the sums over the stored tensor and drive gn_apply; repeated launches are bit-identical."""
    B, H, W, C, N, G = 2, 32, 32, 640, 640, 32
    T, M = H * W, B * H * W
    x, w, b, r = rnd(B, H, W, C, seed=11), rnd(N, C, 3, 3, seed=12, scale=(9 * C) ** -0.5), rnd(N, seed=13).float(), rnd(M, N, seed=14)
    o_ig = torch.empty(M, N, dtype=torch.float16, device=DEV)
    L.run(L.igemm(x.to(DEV), L.pack_conv3x3(w.to(DEV)), o_ig, M=M, Nout=N, C1=C, ldx1=C, CinP=C, ldo=N, bias=b.to(DEV), res=r.to(DEV), ldr=N,
                  taps=9, B=B, Hin=H, Win=W, Hout=H, Wout=W, tile=2, variant=1))
    for sched in ((2, 2, 1, 3), (1, 4, 2, 1)):
        CG, KG, NLD, S = sched
        wp = L.pack_cconv(w.to(DEV), KG)
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        acc = torch.zeros(2, B, G, 2, dtype=torch.int64, device=DEV)
        ws, cnt = _bufs(L, B, H, W, N, CG, S)
        accs, outs = [], []
        for rep in range(2):
            acc.zero_()
            op, keep = L.cconv(x.to(DEV), wp, out, B=B, H=H, W=W, C1=C, ldx1=C, Nout=N, ldo=N, KG=KG, bias=b.to(DEV), res=r.to(DEV), ldr=N,
                               sched=sched, ws=ws, cnt=cnt, cnt_off=3)
            assert L.gn_target(op, acc[0].data_ptr(), T=T, G=G, cpg=N // G, choff=0)
            assert L.gn_target(op, acc[1].data_ptr(), T=T, G=G, cpg=2 * N // G, choff=N)
            assert not L.gn_target(op, acc[1].data_ptr(), T=T, G=G, cpg=N // G, choff=0)
            L.run((op, keep + (acc,)))
            torch.cuda.synchronize()
            accs.append(acc.clone()); outs.append(out.clone())
        assert torch.equal(accs[0], accs[1]) and torch.equal(outs[0], outs[1])
        assert relerr(out, o_ig) <= 1e-3
        o = out.float().cpu().view(B, T, N)
        a0 = accs[0].cpu().double()
        cpg = N // G
        s1, s2 = o.double().view(B, T, G, cpg).sum((1, 3)), (o.double() ** 2).view(B, T, G, cpg).sum((1, 3))
        assert (a0[0, :, :, 0] / 2 ** 20 - s1).abs().max() <= 1e-3 * max(1.0, s1.abs().max().item())
        assert (a0[0, :, :, 1] / 2 ** 12 - s2).abs().max() <= 1e-3 * s2.abs().max().item()
        full = torch.zeros(B, T, 2 * N, dtype=torch.float64)
        full[:, :, N:] = o.double()
        t1, t2 = full.view(B, T, G, 2 * cpg).sum((1, 3)), (full ** 2).view(B, T, G, 2 * cpg).sum((1, 3))
        assert (a0[1, :, :, 0] / 2 ** 20 - t1).abs().max() <= 1e-3 * max(1.0, t1.abs().max().item())
        assert (a0[1, :, :, 1] / 2 ** 12 - t2).abs().max() <= 1e-3 * t2.abs().max().item()
        gm, bt = (1 + 0.1 * rnd(N, seed=15).float()).half(), (0.1 * rnd(N, seed=16).float()).half()
        y = torch.empty(M, N, dtype=torch.float16, device=DEV)
        L.run(L.gn_apply(out, None, gm.to(DEV), bt.to(DEV), y, B=B, T=T, C1=N, ld1=N, G=G, nchunk=0, eps=1e-5, silu=True, acc_ptr=acc[0].data_ptr()))
        torch.cuda.synchronize()
        gref = F.silu(F.group_norm(o.permute(0, 2, 1), G, gm.float(), bt.float(), 1e-5)).permute(0, 2, 1).reshape(M, N)
        assert relerr(y, gref) <= 2e-3


@pytest.mark.parametrize("B,H,W,C1,C2,N,G,sched", [
    (2, 32, 32, 640, 0, 640, 32, (1, 4, 4, 1)),       # level 1 resnet conv behind its GroupNorm
    (2, 32, 32, 640, 320, 640, 32, (2, 2, 4, 3)),     # concat input (up block): groups straddle nothing, chunks stay inside one input
    (2, 16, 16, 1280, 0, 1280, 32, (1, 4, 4, 3)),
    (2, 16, 16, 1280, 1280, 1280, 32, (1, 4, 2, 3)),
    (3, 8, 16, 64, 0, 64, 4, (1, 4, 1, 1)),           # one patch per sample: every halo pixel is padding
    (2, 16, 32, 128, 64, 128, 8, (2, 2, 2, 2)),
])
def test_cconv_fused_groupnorm_silu(L, B, H, W, C1, C2, N, G, sched):
    """conv(silu(GroupNorm(x))) (reference resnet.py:233-234, 249-250) as ONE launch: statistics as the producers leave them (fixed-point
    int64 sums), normalisation in the loader waves, zero padding applied to the NORMALISED tensor.  Against fp32 torch, and against the
    two-launch path (gn_apply, then cconv) it replaces."""
    CG, KG, NLD, S = sched
    C, T, M = C1 + C2, H * W, B * H * W
    x = rnd(B, H, W, C, seed=41) * 1.5 + rnd(1, 1, 1, C, seed=42)           # per-channel offsets: the mean matters
    w = rnd(N, C, 3, 3, seed=43, scale=(9 * C) ** -0.5)
    b = rnd(N, seed=44).float()
    gm, bt = (1 + 0.2 * rnd(C, seed=45).float()).half(), (0.2 * rnd(C, seed=46).float()).half()
    eps = 1e-5
    xn = F.silu(F.group_norm(x.float().permute(0, 3, 1, 2), G, gm.float(), bt.float(), eps))
    ref = F.conv2d(xn.half().float(), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(M, N)
    # the statistics exactly as a producer's epilogue accumulates them
    cpg = C // G
    xs = x.double().view(B, T, G, cpg)
    acc = torch.stack([(xs.sum((1, 3)) * 2 ** 20).round(), ((xs ** 2).sum((1, 3)) * 2 ** 12).round()], -1).to(torch.int64).to(DEV)
    xd = x.to(DEV)
    x1 = xd[..., :C1].contiguous()
    x2 = xd[..., C1:].contiguous() if C2 else None
    wp = L.pack_cconv(w.to(DEV), KG)
    out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    ws, cnt = _bufs(L, B, H, W, N, CG, S)
    L.run(L.cconv(x1, wp, out, B=B, H=H, W=W, C1=C1, ldx1=C1, Nout=N, ldo=N, KG=KG, x2=x2, C2=C2, ldx2=C2, bias=b.to(DEV), sched=sched, ws=ws,
                  cnt=cnt, cnt_off=3, gn_acc_ptr=acc.data_ptr(), gn_gamma=gm.to(DEV), gn_beta=bt.to(DEV), gn_G=G, gn_eps=eps))
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    e = relerr(out, ref)
    assert e <= 2e-3, f"cconv + GroupNorm {B}x{H}x{W} C{C1}+{C2}->{N} G{G} sched {sched}: rel-L2 {e:.3e}"
    # the two launches it replaces
    y = torch.empty(M, C, dtype=torch.float16, device=DEV)
    L.run(L.gn_apply(x1, None, gm.to(DEV), bt.to(DEV), y, B=B, T=T, C1=C1, ld1=C1, G=G, nchunk=0, eps=eps, silu=True, x2=x2, C2=C2, ld2=C2,
                     acc_ptr=acc.data_ptr()))
    out2 = torch.zeros_like(out)
    if ws is not None:
        ws.fill_(float("nan"))
    L.run(L.cconv(y, wp, out2, B=B, H=H, W=W, C1=C, ldx1=C, Nout=N, ldo=N, KG=KG, bias=b.to(DEV), sched=sched, ws=ws, cnt=cnt, cnt_off=3))
    torch.cuda.synchronize()
    assert relerr(out, out2) <= 3e-4, relerr(out, out2)


def test_cconv_rejects_bad_arguments(L):
    from live2diff_amd import _lib
    x, w = rnd(1, 8, 16, 64).to(DEV), L.pack_cconv(rnd(64, 64, 3, 3).to(DEV), 4)
    out = torch.empty(128, 64, dtype=torch.float16, device=DEV)
    with pytest.raises(_lib.L2DError):
        L.run(L.cconv(x, w, out, B=1, H=8, W=16, C1=64, ldx1=64, Nout=64, ldo=64, KG=4, sched=(1, 4, 3, 1)))      # no such loader count
    with pytest.raises(_lib.L2DError):
        L.run(L.cconv(x, w, out, B=1, H=8, W=16, C1=64, ldx1=64, Nout=64, ldo=60, KG=4, sched=(1, 4, 1, 1)))      # ldo % 8
    with pytest.raises(_lib.L2DError):
        L.run(L.cconv(x, w, out, B=1, H=8, W=8, C1=64, ldx1=64, Nout=64, ldo=64, KG=4, sched=(1, 4, 1, 1)))       # W % 16
    ws, cnt = torch.zeros(1 << 16, dtype=torch.float32, device=DEV), torch.zeros(16, dtype=torch.int32, device=DEV)
    with pytest.raises(_lib.L2DError):
        L.run(L.cconv(x, w, out, B=1, H=8, W=16, C1=64, ldx1=64, Nout=64, ldo=64, KG=4, sched=(1, 4, 1, 2), ws=ws, cnt=cnt))      # more slices than chunks
