"""-m gpu: the patch-resident 3x3 convolution (csrc/pconv.hip, L2D_OP_PCONV) through the C ABI against F.conv2d in fp32 on the
same fp16-rounded inputs, and against the implicit-GEMM kernel it replaces: every patch geometry, channel concat of two inputs,
zero padding at all four image borders, bias + per-sample time-embedding bias + residual, GroupNorm statistics of the output.
Tolerance: rel-L2 <= 2e-3 (fp16 storage, fp32 accumulate), SURVEY.md section 8c."""
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


@pytest.mark.parametrize("B,H,W,C1,C2,N,patch", [
    (2, 64, 64, 320, 0, 320, (8, 16)),        # level 0 resnet conv
    (2, 64, 64, 320, 0, 320, (8, 8)),
    (2, 64, 64, 320, 0, 320, (4, 8)),
    (2, 64, 64, 320, 320, 320, (8, 16)),      # two inputs (channel concat), K = 5760
    (2, 32, 32, 640, 0, 640, (8, 8)),         # level 1
    (2, 32, 32, 640, 320, 640, (8, 16)),
    (2, 16, 16, 1280, 0, 1280, (4, 8)),       # level 2
    (1, 8, 16, 64, 0, 64, (8, 16)),           # one patch: every pixel is a border pixel somewhere
    (3, 8, 8, 64, 64, 128, (4, 8)),
    (8, 16, 16, 128, 0, 64, (8, 8)),          # warm-up style batch
    (2, 64, 96, 320, 0, 320, (8, 16)),        # non-square (cfg-3 aspect)
])
def test_pconv_matches_conv2d(L, B, H, W, C1, C2, N, patch):
    C = C1 + C2
    x = rnd(B, H, W, C, seed=1)
    w = rnd(N, C, 3, 3, seed=2, scale=(9 * C) ** -0.5)
    b = rnd(N, seed=3).float()
    temb = rnd(B, 2 * N, seed=4).float()               # per-sample row bias, this conv's columns start at offset N
    r = rnd(B * H * W, N, seed=5)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1) + temb[:, N:, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, N).half().float() + r.float()
    wp = L.pack_conv3x3(w.to(DEV))
    xd = x.to(DEV)
    x1 = xd[..., :C1].contiguous()
    x2 = xd[..., C1:].contiguous() if C2 else None
    out = torch.zeros(B * H * W, N, dtype=torch.float16, device=DEV)
    tb = temb.to(DEV)
    for order in (0, 1):
        out.zero_()
        op, keep = L.pconv(x1, wp, out, B=B, H=H, W=W, C1=C1, ldx1=C1, CinP=C, Nout=N, ldo=N, patch=patch, x2=x2, C2=C2, ldx2=C2,
                           bias=b.to(DEV), rowbias=tb, ldrb=2 * N, rows_per_bias=H * W, res=r.to(DEV), ldr=N, order=order)
        op.p[4] = tb.data_ptr() + 4 * N
        L.run((op, keep))
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        e = relerr(out, ref)
        assert e <= 2e-3, f"pconv {B}x{H}x{W} C{C1}+{C2}->{N} patch {patch} order {order}: rel-L2 {e:.3e}"
    # one bias row for all samples (warm-up pass), no residual, no bias
    out2 = torch.zeros_like(out)
    L.run(L.pconv(x1, wp, out2, B=B, H=H, W=W, C1=C1, ldx1=C1, CinP=C, Nout=N, ldo=N, patch=patch, x2=x2, C2=C2, ldx2=C2,
                  rowbias=tb, ldrb=2 * N, rows_per_bias=B * H * W))
    torch.cuda.synchronize()
    ref2 = (F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, padding=1) + temb[:1, :N, None, None]).permute(0, 2, 3, 1).reshape(B * H * W, N)
    assert relerr(out2, ref2) <= 2e-3


def test_pconv_agrees_with_igemm_and_feeds_groupnorm(L):
    """Same operands through both conv kernels; the output's GroupNorm statistics (two consumers) from pconv's epilogue equal the
    sums over the stored tensor and drive gn_apply; repeated launches are bit-identical."""
    B, H, W, C, N, G = 2, 32, 32, 640, 640, 32
    T, M = H * W, B * H * W
    x, w, b, r = rnd(B, H, W, C, seed=11), rnd(N, C, 3, 3, seed=12, scale=(9 * C) ** -0.5), rnd(N, seed=13).float(), rnd(M, N, seed=14)
    wp = L.pack_conv3x3(w.to(DEV))
    o_ig = torch.empty(M, N, dtype=torch.float16, device=DEV)
    L.run(L.igemm(x.to(DEV), wp, o_ig, M=M, Nout=N, C1=C, ldx1=C, CinP=C, ldo=N, bias=b.to(DEV), res=r.to(DEV), ldr=N, taps=9, B=B, Hin=H,
                  Win=W, Hout=H, Wout=W, tile=2, variant=1))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    acc = torch.zeros(2, B, G, 2, dtype=torch.int64, device=DEV)
    accs, outs = [], []
    for rep in range(2):
        acc.zero_()
        op, keep = L.pconv(x.to(DEV), wp, out, B=B, H=H, W=W, C1=C, ldx1=C, CinP=C, Nout=N, ldo=N, patch=(8, 8), bias=b.to(DEV), res=r.to(DEV), ldr=N)
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


def test_pconv_rejects_bad_arguments(L):
    from live2diff_amd import _lib
    x, w = rnd(1, 8, 16, 64).to(DEV), L.pack_conv3x3(rnd(64, 64, 3, 3).to(DEV))
    out = torch.empty(128, 64, dtype=torch.float16, device=DEV)
    with pytest.raises(_lib.L2DError):
        L.run(L.pconv(x, w, out, B=1, H=8, W=16, C1=64, ldx1=64, CinP=64, Nout=64, ldo=64, patch=(8, 12)))      # no such patch
    with pytest.raises(_lib.L2DError):
        L.run(L.pconv(x, w, out, B=1, H=8, W=16, C1=64, ldx1=64, CinP=64, Nout=64, ldo=60, patch=(8, 16)))      # ldo % 8


@pytest.mark.parametrize("B,H,W,C,N,patch", [(1, 128, 128, 64, 64, (8, 16)), (2, 48, 64, 256, 128, (8, 8)), (8, 32, 32, 64, 64, (4, 8))])
def test_pconv_activation_epilogues(L, B, H, W, C, N, patch):
    """The epilogue modes the TAESD / DPT decoder convs use (as igemm): ReLU, SiLU, GELU in fp32 before the fp16 rounding;
    relu(conv(x) + skip) with the add in fp16 after the conv output is rounded (the reference's fp16 graph)."""
    x = rnd(B, H, W, C, seed=21)
    w = rnd(N, C, 3, 3, seed=22, scale=(9 * C) ** -0.5)
    b = rnd(N, seed=23).float()
    r = rnd(B * H * W, N, seed=24)
    conv = (F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1)).permute(0, 2, 3, 1).reshape(B * H * W, N)
    refs = {0: conv, 3: F.relu(conv), 2: F.silu(conv), 5: F.gelu(conv), 4: F.relu(conv.half().float() + r.float())}
    wp = L.pack_conv3x3(w.to(DEV))
    xd = x.to(DEV).reshape(B * H * W, C)
    for epi, ref in refs.items():
        out = torch.zeros(B * H * W, N, dtype=torch.float16, device=DEV)
        L.run(L.pconv(xd, wp, out, B=B, H=H, W=W, C1=C, ldx1=C, CinP=C, Nout=N, ldo=N, patch=patch, bias=b.to(DEV),
                      res=(r.to(DEV) if epi == 4 else None), ldr=(N if epi == 4 else 0), epi=epi))
        torch.cuda.synchronize()
        e = relerr(out, ref)
        assert torch.isfinite(out.float()).all() and e <= 2e-3, f"pconv epi {epi}: rel-L2 {e:.3e}"
        # the implicit-GEMM kernel with the same epilogue mode gives the same tensor up to summation order
        out_i = torch.zeros_like(out)
        L.run(L.igemm(xd, wp, out_i, M=B * H * W, Nout=N, C1=C, ldx1=C, CinP=C, ldo=N, bias=b.to(DEV), taps=9, B=B, Hin=H, Win=W,
                      Hout=H, Wout=W, epi=epi, res=(r.to(DEV) if epi == 4 else None), ldr=(N if epi == 4 else 0), tile=2, variant=1))
        torch.cuda.synchronize()
        assert relerr(out, out_i) <= 1e-3, f"pconv vs igemm, epi {epi}"
    from live2diff_amd import _lib
    with pytest.raises(_lib.L2DError):               # relu-after-add needs the skip tensor
        L.run(L.pconv(xd, wp, out, B=B, H=H, W=W, C1=C, ldx1=C, CinP=C, Nout=N, ldo=N, patch=patch, epi=4))
