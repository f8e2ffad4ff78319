"""-m gpu: the token-row GEMM (csrc/rowgemm.hip, L2D_OP_ROWGEMM) through the C ABI against fp32 torch references on the same
fp16-rounded inputs: plain / bias / residual, LayerNorm and GroupNorm prologues (affine folded into the packed weights), GEGLU,
the transposed (V^T) part, GroupNorm statistics of the output, every (NW, NT, MT) geometry and both k-loop forms.

Tolerance: per-op rel-L2 <= 2e-3 (3e-3 behind a fused norm: the normalised activations are rounded to fp16 before the affine
map instead of after it), SURVEY.md section 8c."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def check(a, b, tol=2e-3, what=""):
    assert torch.isfinite(a.float()).all(), f"{what}: non-finite output"
    e = relerr(a, b)
    assert e <= tol, f"{what}: rel-L2 {e:.3e} > {tol:.1e}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16)


@pytest.fixture(scope="module")
def L():
    from live2diff_amd import _lib, ops
    print("device:", _lib.device_name())
    return ops


def _geoms(tiles, ntr_tiles=0, mt_ok=True, mt4=False):
    """mt4: also the 128-token tiles (K = 320 only)"""
    out = []
    for mt in (1, 2, 4):
        if (mt == 2 and not mt_ok) or (mt == 4 and not mt4):
            continue
        for nt in (1, 2, 3, 4):
            if mt >= 2 and nt > 2:
                continue
            for nw in range(1, (5 if nt >= 3 else 8) + 1):
                if tiles % (nw * nt) == 0 and ntr_tiles % (nw * nt) == 0:
                    out.append((nw, nt, mt))
    return out


@pytest.mark.parametrize("M,K,N", [(8192, 320, 320), (2048, 640, 640), (512, 1280, 1280), (128, 1280, 1280), (300, 64, 96),
                                   (77, 128, 32), (8192, 1280, 320), (1000, 192, 256), (64, 2048, 64)])
def test_rowgemm_linear_bias_residual(L, M, K, N):
    """out = x W^T + b + r for the frame's shapes and odd ones (ragged M, generic k loop), default schedule"""
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3).float(), rnd(M, N, seed=4)
    ref = (x.float() @ w.float().t() + b).half().float() + r.float()
    wp, bp = L.pack_rowgemm(w.to(DEV), b.to(DEV))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    L.run(L.rowgemm(x.to(DEV), wp, out, M=M, K=K, Nout=N, ldx=K, ldo=N, bias=bp, res=r.to(DEV), ldr=N))
    torch.cuda.synchronize()
    check(out, ref, what=f"rowgemm {M}x{K}x{N}")
    out2 = torch.empty_like(out)                      # no bias, no residual
    L.run(L.rowgemm(x.to(DEV), wp, out2, M=M, K=K, Nout=N, ldx=K, ldo=N))
    torch.cuda.synchronize()
    check(out2, x.float() @ w.float().t(), what="no bias")


@pytest.mark.parametrize("K", [320, 128])          # straight-line k loop (K = 320) and the generic one
def test_rowgemm_every_geometry(L, K):
    """All (NW, NT, MT) block geometries give the same matrix (N = 960: 30 weight tiles; ragged M)"""
    M, N = 1000, 960
    x, w, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=K ** -0.5), rnd(N, seed=13).float()
    ref = x.float() @ w.float().t() + b
    wp, bp = L.pack_rowgemm(w.to(DEV), b.to(DEV))
    seen = 0
    for sched in _geoms(N // 32, mt4=(K == 320)):
        out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        for order in (0, 1):
            L.run(L.rowgemm(x.to(DEV), wp, out, M=M, K=K, Nout=N, ldx=K, ldo=N, bias=bp, sched=sched, order=order))
            torch.cuda.synchronize()
            check(out, ref, what=f"geometry {sched} order {order}")
        seen += 1
    assert seen >= 12


@pytest.mark.parametrize("M,C,N", [(8192, 320, 960), (2048, 640, 1920), (512, 1280, 3840), (128, 1280, 1280), (100, 64, 64), (96, 256, 128)])
def test_rowgemm_layernorm_prologue(L, M, C, N):
    """LayerNorm(x) W^T with gamma / beta folded into the packed weight / bias, vs F.layer_norm + linear in fp32"""
    x = (rnd(M, C, seed=21).float() * 1.5 + 0.3).half()
    w = rnd(N, C, seed=22, scale=C ** -0.5)
    gm, bt = (1 + 0.2 * rnd(C, seed=23).float()).half(), (0.2 * rnd(C, seed=24).float()).half()
    ref = F.layer_norm(x.float(), (C,), gm.float(), bt.float(), 1e-5) @ w.float().t()
    wp, bp = L.pack_rowgemm(w.to(DEV), None, gm.to(DEV), bt.to(DEV))
    assert bp is not None
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    L.run(L.rowgemm(x.to(DEV), wp, out, M=M, K=C, Nout=N, ldx=C, ldo=N, bias=bp, pro=1, eps=1e-5))
    torch.cuda.synchronize()
    check(out, ref, tol=3e-3, what=f"LN + linear {M}x{C}x{N}")
    if C >= 128:                                      # the 16-lanes-per-row load path (512-thread blocks) and narrow blocks
        for sched in ((8, 1, 1), (2, 1, 1), (1, 2, 1)):
            if (N // 32) % (sched[0] * sched[1]):
                continue
            L.run(L.rowgemm(x.to(DEV), wp, out, M=M, K=C, Nout=N, ldx=C, ldo=N, bias=bp, pro=1, eps=1e-5, sched=sched))
            torch.cuda.synchronize()
            check(out, ref, tol=3e-3, what=f"LN + linear, geometry {sched}")


@pytest.mark.parametrize("B,T,C", [(2, 4096, 320), (2, 1024, 640), (2, 256, 1280), (2, 64, 1280), (8, 64, 320), (1, 64, 64)])
def test_rowgemm_groupnorm_prologue(L, B, T, C):
    """GroupNorm(32)(x) W^T + b: statistics arrive as the producers' fixed-point accumulators (filled here by an igemm launch
    that writes x, as in the plan), affine folded into the packed weights; vs F.group_norm + linear."""
    G, M = 32, B * T
    x0, wid = rnd(M, C, seed=31), torch.eye(C).half()
    x = torch.empty(M, C, dtype=torch.float16, device=DEV)
    acc = torch.zeros(B, G, 2, dtype=torch.int64, device=DEV)
    wi = L.pack_linear(wid.to(DEV))
    op, keep = L.igemm(x0.to(DEV), wi, x, M=M, Nout=C, C1=C, ldx1=C, CinP=wi.shape[1], ldo=C, tile=2, variant=1)
    assert L.gn_target(op, acc.data_ptr(), T=T, G=G, cpg=C // G, choff=0)
    L.run((op, keep + (acc,)))
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), x0)
    w, b = rnd(C, C, seed=32, scale=C ** -0.5), rnd(C, seed=33).float()
    gm, bt = (1 + 0.2 * rnd(C, seed=34).float()).half(), (0.2 * rnd(C, seed=35).float()).half()
    eps = 1e-6
    xn = F.group_norm(x0.float().view(B, T, C).permute(0, 2, 1), G, gm.float(), bt.float(), eps).permute(0, 2, 1).reshape(M, C)
    ref = xn @ w.float().t() + b
    wp, bp = L.pack_rowgemm(w.to(DEV), b.to(DEV), gm.to(DEV), bt.to(DEV))
    out = torch.empty(M, C, dtype=torch.float16, device=DEV)
    L.run(L.rowgemm(x, wp, out, M=M, K=C, Nout=C, ldx=C, ldo=C, bias=bp, pro=2, eps=eps, T=T, G=G, gn_acc_ptr=acc.data_ptr()))
    torch.cuda.synchronize()
    check(out, ref, tol=3e-3, what=f"GN + linear B{B} T{T} C{C}")


@pytest.mark.parametrize("M,C", [(8192, 320), (2048, 640), (512, 1280), (128, 1280), (200, 64)])
def test_rowgemm_geglu_with_layernorm(L, M, C):
    """LayerNorm -> Linear(C, 8C) -> value * gelu(gate) (diffusers GEGLU, exact-erf GELU) in one launch"""
    x = rnd(M, C, seed=41)
    w, b = rnd(8 * C, C, seed=42, scale=C ** -0.5), rnd(8 * C, seed=43).float()
    gm, bt = (1 + 0.2 * rnd(C, seed=44).float()).half(), (0.2 * rnd(C, seed=45).float()).half()
    h = F.layer_norm(x.float(), (C,), gm.float(), bt.float(), 1e-5) @ w.float().t() + b
    ref = h[:, :4 * C] * F.gelu(h[:, 4 * C:])
    wp, bp = L.pack_rowgemm(w.to(DEV), b.to(DEV), gm.to(DEV), bt.to(DEV), geglu=True)
    out = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    L.run(L.rowgemm(x.to(DEV), wp, out, M=M, K=C, Nout=8 * C, ldx=C, ldo=4 * C, bias=bp, pro=1, eps=1e-5, epi=1))
    torch.cuda.synchronize()
    check(out, ref, tol=3e-3, what=f"LN + GEGLU {M}x{C}")
    if C == 320:
        for sched in ((8, 2, 2), (4, 2, 2), (5, 4, 1), (8, 1, 1), (4, 1, 2), (5, 2, 4), (8, 1, 4), (4, 2, 4), (5, 1, 4)):
            L.run(L.rowgemm(x.to(DEV), wp, out, M=M, K=C, Nout=8 * C, ldx=C, ldo=4 * C, bias=bp, pro=1, eps=1e-5, epi=1, sched=sched))
            torch.cuda.synchronize()
            check(out, ref, tol=3e-3, what=f"LN + GEGLU geometry {sched}")


@pytest.mark.parametrize("B,T,C", [(2, 4096, 320), (2, 1024, 640), (2, 256, 1280), (2, 64, 1280), (3, 96, 64), (8, 64, 320)])
def test_rowgemm_qkv_with_transposed_v(L, B, T, C):
    """norm1 -> q | k | v in one launch: q | k as [M][2C] rows, V as V^T[sample][channel][ldvt] (what the flash kernel reads)"""
    M, ldvt = B * T, (T + 7) // 8 * 8 + 8
    x = rnd(M, C, seed=51)
    w = rnd(3 * C, C, seed=52, scale=C ** -0.5)
    gm, bt = (1 + 0.2 * rnd(C, seed=53).float()).half(), (0.2 * rnd(C, seed=54).float()).half()
    ref = F.layer_norm(x.float(), (C,), gm.float(), bt.float(), 1e-5) @ w.float().t()
    wp, bp = L.pack_rowgemm(w.to(DEV), None, gm.to(DEV), bt.to(DEV))
    qk = torch.empty(M, 2 * C, dtype=torch.float16, device=DEV)
    vt = torch.full((B, C, ldvt), 7.0, dtype=torch.float16, device=DEV)
    geoms = [None] + [g for g in _geoms(3 * C // 32, C // 32, mt_ok=(T % 64 == 0))][:10]
    if C == 320 and T % 128 == 0:
        geoms += [(5, 1, 4), (5, 2, 4), (2, 1, 4), (1, 2, 4)]
    for sched in geoms:
        qk.zero_(); vt.fill_(7.0)
        L.run(L.rowgemm(x.to(DEV), wp, qk, M=M, K=C, Nout=3 * C, ldx=C, ldo=2 * C, bias=bp, pro=1, eps=1e-5, T=T, out_t=vt, ntr=C,
                        ldt=ldvt, st=C * ldvt, sched=sched))
        torch.cuda.synchronize()
        check(qk, ref[:, :2 * C], tol=3e-3, what=f"q|k {sched}")
        vref = ref[:, 2 * C:].view(B, T, C).permute(0, 2, 1)
        check(vt[:, :, :T], vref, tol=3e-3, what=f"V^T {sched}")
        assert (vt[:, :, T:] == 7.0).all(), "columns beyond T must not be written"


@pytest.mark.parametrize("B,T,K,C,choff2,Ccat", [(2, 4096, 320, 320, 0, 640), (2, 1024, 640, 640, 640, 1280), (2, 256, 1280, 1280, 0, 2560),
                                                 (2, 64, 1280, 1280, 1280, 2560), (8, 64, 320, 320, 0, 640), (2, 4096, 1280, 320, 320, 640)])
def test_rowgemm_groupnorm_statistics_of_the_output(L, B, T, K, C, choff2, Ccat):
    """proj_out-type launches accumulate sum / sum of squares of what they store for up to two consumer GroupNorms (same
    fixed-point protocol as igemm); gn_apply from those accumulators equals GroupNorm of the stored tensor; repeats are bit-equal."""
    G, M = 32, B * T
    x, w, b, r = rnd(M, K, seed=61), rnd(C, K, seed=62, scale=K ** -0.5), rnd(C, seed=63).float(), rnd(M, C, seed=64)
    wp, bp = L.pack_rowgemm(w.to(DEV), b.to(DEV))
    out = torch.empty(M, C, dtype=torch.float16, device=DEV)
    acc = torch.zeros(2, B, G, 2, dtype=torch.int64, device=DEV)
    cpg1, cpg2 = C // G, Ccat // G
    accs = []
    for rep in range(2):
        acc.zero_()
        op, keep = L.rowgemm(x.to(DEV), wp, out, M=M, K=K, Nout=C, ldx=K, ldo=C, bias=bp, res=r.to(DEV), ldr=C)
        assert L.gn_target(op, acc[0].data_ptr(), T=T, G=G, cpg=cpg1, choff=0)
        assert L.gn_target(op, acc[1].data_ptr(), T=T, G=G, cpg=cpg2, choff=choff2)
        assert not L.gn_target(op, acc[1].data_ptr(), T=T, G=G, cpg=cpg2, choff=0)
        L.run((op, keep + (acc,)))
        torch.cuda.synchronize()
        accs.append(acc.clone())
    assert torch.equal(accs[0], accs[1])
    check(out, (x.float() @ w.float().t() + b).half().float() + r.float(), what="gemm output")
    o = out.float().cpu().view(B, T, C)
    a0 = accs[0].cpu().double()
    s1, s2 = o.double().view(B, T, G, cpg1).sum((1, 3)), (o.double() ** 2).view(B, T, G, cpg1).sum((1, 3))
    assert (a0[0, :, :, 0] / 2 ** 20 - s1).abs().max() <= 1e-3 * max(1.0, s1.abs().max().item())
    assert (a0[0, :, :, 1] / 2 ** 12 - s2).abs().max() <= 1e-3 * s2.abs().max().item()
    full = torch.zeros(B, T, Ccat, dtype=torch.float64)
    full[:, :, choff2:choff2 + C] = o.double()
    t1, t2 = full.view(B, T, G, cpg2).sum((1, 3)), (full ** 2).view(B, T, G, cpg2).sum((1, 3))
    assert (a0[1, :, :, 0] / 2 ** 20 - t1).abs().max() <= 1e-3 * max(1.0, t1.abs().max().item())
    assert (a0[1, :, :, 1] / 2 ** 12 - t2).abs().max() <= 1e-3 * t2.abs().max().item()
    gm, bt = (1 + 0.1 * rnd(C, seed=65).float()).half(), (0.1 * rnd(C, seed=66).float()).half()
    y = torch.empty(M, C, dtype=torch.float16, device=DEV)
    L.run(L.gn_apply(out, None, gm.to(DEV), bt.to(DEV), y, B=B, T=T, C1=C, ld1=C, G=G, nchunk=0, eps=1e-5, silu=True, acc_ptr=acc[0].data_ptr()))
    torch.cuda.synchronize()
    gref = F.silu(F.group_norm(o.permute(0, 2, 1), G, gm.float(), bt.float(), 1e-5)).permute(0, 2, 1).reshape(M, C)
    check(y, gref, what="gn_apply from rowgemm statistics")


def test_rowgemm_128_token_tiles(L):
    """MT = 4 (K = 320): GroupNorm prologue, bias, residual (fetched in the epilogue here) and output statistics; ragged M for the
    plain form; the same results as the default geometry."""
    B, T, C, G = 2, 4096, 320, 32
    M = B * T
    x0 = rnd(M, C, seed=71)
    x = torch.empty(M, C, dtype=torch.float16, device=DEV)
    acc = torch.zeros(B, G, 2, dtype=torch.int64, device=DEV)
    wi = L.pack_linear(torch.eye(C).half().to(DEV))
    op, keep = L.igemm(x0.to(DEV), wi, x, M=M, Nout=C, C1=C, ldx1=C, CinP=wi.shape[1], ldo=C, tile=2, variant=1)
    assert L.gn_target(op, acc.data_ptr(), T=T, G=G, cpg=C // G, choff=0)
    L.run((op, keep + (acc,)))
    w, b, r = rnd(C, C, seed=72, scale=C ** -0.5), rnd(C, seed=73).float(), rnd(M, C, seed=74)
    gm, bt = (1 + 0.2 * rnd(C, seed=75).float()).half(), (0.2 * rnd(C, seed=76).float()).half()
    xn = F.group_norm(x0.float().view(B, T, C).permute(0, 2, 1), G, gm.float(), bt.float(), 1e-6).permute(0, 2, 1).reshape(M, C)
    ref = (xn @ w.float().t() + b).half().float() + r.float()
    wp, bp = L.pack_rowgemm(w.to(DEV), b.to(DEV), gm.to(DEV), bt.to(DEV))
    outs = []
    for sched in ((5, 1, 1), (5, 1, 4), (5, 2, 4), (2, 1, 4)):
        out = torch.zeros(M, C, dtype=torch.float16, device=DEV)
        acc2 = torch.zeros(B, G, 2, dtype=torch.int64, device=DEV)
        op, keep = L.rowgemm(x, wp, out, M=M, K=C, Nout=C, ldx=C, ldo=C, bias=bp, res=r.to(DEV), ldr=C, pro=2, eps=1e-6, T=T, G=G,
                             gn_acc_ptr=acc.data_ptr(), sched=sched)
        assert L.gn_target(op, acc2.data_ptr(), T=T, G=G, cpg=C // G, choff=0)
        L.run((op, keep + (acc2,)))
        torch.cuda.synchronize()
        check(out, ref, tol=3e-3, what=f"GN + linear + residual, geometry {sched}")
        outs.append((out.clone(), acc2.clone()))
    for o, a2 in outs[1:]:
        assert torch.equal(o, outs[0][0]), "128-token tiles must reproduce the default geometry bit for bit"
        # (the statistics are summed in fp32 per block before the fixed-point atomics: another tile shape, another rounding)
        d = (a2 - outs[0][1]).abs().double() / outs[0][1].abs().double().clamp_min(1.0)
        assert d.max().item() < 1e-4
    # ragged M (not a multiple of 128), plain linear
    Mr = 1000
    xr, wr = rnd(Mr, C, seed=77), rnd(960, C, seed=78, scale=C ** -0.5)
    wpr, _ = L.pack_rowgemm(wr.to(DEV))
    o = torch.zeros(Mr, 960, dtype=torch.float16, device=DEV)
    L.run(L.rowgemm(xr.to(DEV), wpr, o, M=Mr, K=C, Nout=960, ldx=C, ldo=960, sched=(5, 2, 4)))
    torch.cuda.synchronize()
    check(o, xr.float() @ wr.float().t(), what="ragged M, 128-token tiles")


def test_rowgemm_rejects_bad_arguments(L):
    from live2diff_amd import _lib
    x, w = rnd(64, 64).to(DEV), rnd(64, 64).to(DEV)
    wp, _ = L.pack_rowgemm(w)
    out = torch.empty(64, 64, dtype=torch.float16, device=DEV)
    with pytest.raises(_lib.L2DError):
        L.run(L.rowgemm(x, wp, out, M=64, K=64, Nout=64, ldx=64, ldo=64, sched=(3, 1, 1)))      # 2 tiles do not split over 3 waves
    with pytest.raises(_lib.L2DError):
        L.run(L.rowgemm(x, wp, out, M=64, K=64, Nout=64, ldx=64, ldo=60))                        # ldo % 8
    with pytest.raises(_lib.L2DError):
        L.run(L.rowgemm(x, wp, out, M=64, K=64, Nout=64, ldx=64, ldo=64, pro=2, T=48, G=32, gn_acc_ptr=out.data_ptr()))   # T % 32
    with pytest.raises(_lib.L2DError):
        L.run(L.rowgemm(x, wp, out, M=64, K=64, Nout=64, ldx=64, ldo=64, sched=(2, 1, 4)))      # 128-token tiles: K = 320 only
