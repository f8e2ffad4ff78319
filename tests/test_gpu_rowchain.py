"""-m gpu: the token-resident block tail (csrc/rowchain.hip, L2D_OP_ROWCHAIN) through the C ABI against (a) the same chain in fp32
torch on the same fp16 inputs (reference semantics: attention.py:243-270,125-133; motion_module.py:401-435,290-297 -- to_out +
residual, LayerNorm, GEGLU feed-forward + residual, proj_out + block residual) and (b) the four separate launches it replaces.

Tolerance: rel-L2 <= 3e-3 against fp32 (four chained fp16 GEMMs with their fp16 rounding points); BIT-IDENTICAL to the unfused HIP
path (same rounding points, the same fp32 sums in the same k order: measured 0.00e+00 on every case, profiles/round5_n_rowchain_ab.txt,
asserted with torch.equal since round 6)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
C = 320


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


@pytest.fixture(scope="module")
def layers(L):
    w = dict(wo=rnd(C, C, seed=1, scale=C ** -0.5), bo=rnd(C, seed=2, scale=0.1).float(),
             gm=(1 + 0.2 * rnd(C, seed=3).float()).half(), bt=(0.2 * rnd(C, seed=4).float()).half(),
             w1=rnd(8 * C, C, seed=5, scale=C ** -0.5), b1=rnd(8 * C, seed=6, scale=0.1).float(),
             w2=rnd(C, 4 * C, seed=7, scale=(4 * C) ** -0.5), b2=rnd(C, seed=8, scale=0.1).float(),
             wp=rnd(C, C, seed=9, scale=C ** -0.5), bp=rnd(C, seed=10, scale=0.1).float())
    d = {k: v.to(DEV) for k, v in w.items()}
    packed = dict(zip(("w_out", "b_out"), L.pack_rowgemm(d["wo"], d["bo"])))
    packed.update(zip(("w_ff1", "b_ff1"), L.pack_rowgemm(d["w1"], d["b1"], d["gm"], d["bt"], geglu=True)))
    packed.update(zip(("w_ff2", "b_ff2"), L.pack_rowgemm(d["w2"], d["b2"])))
    packed.update(zip(("w_po", "b_po"), L.pack_rowgemm(d["wp"], d["bp"])))
    return w, d, packed


def reference(w, a, r1, r2):
    f = lambda t: t.float()
    h2 = (f(a) @ f(w["wo"]).t() + w["bo"]).half().float() + f(r1)
    h2 = h2.half().float()
    n = F.layer_norm(h2, (C,), f(w["gm"]), f(w["bt"]), 1e-5)
    v, g = (n @ f(w["w1"]).t() + w["b1"]).chunk(2, dim=-1)
    hid = (v * F.gelu(g)).half().float()
    h3 = ((hid @ f(w["w2"]).t() + w["b2"]).half().float() + h2).half().float()
    return (h3 @ f(w["wp"]).t() + w["bp"]).half().float() + f(r2)


@pytest.mark.parametrize("B,T", [(2, 4096), (1, 6144), (3, 2048), (8, 1024)])
def test_rowchain_against_fp32_and_the_unfused_path(L, layers, B, T):
    from live2diff_amd import _lib
    w, d, pk = layers
    M, G = B * T, 32
    a, r1, r2 = rnd(M, C, seed=11), rnd(M, C, seed=12), rnd(M, C, seed=13)
    ad, r1d, r2d = a.to(DEV), r1.to(DEV), r2.to(DEV)
    ref = reference(w, a, r1, r2)
    out = torch.zeros(M, C, dtype=torch.float16, device=DEV)
    acc = torch.zeros(2, B, G, 2, dtype=torch.int64, device=DEV)
    op, keep = L.rowchain(ad, r1d, r2d, out, M=M, C=C, eps=1e-5, **pk)
    assert L.gn_target(op, acc[0].data_ptr(), T=T, G=G, cpg=C // G, choff=0)
    assert L.gn_target(op, acc[1].data_ptr(), T=T, G=G, cpg=2 * C // G, choff=C)        # second half of a 2C-wide concat GroupNorm
    assert not L.gn_target(op, acc[1].data_ptr(), T=T, G=G, cpg=C // G, choff=0)        # both slots taken
    pl = _lib.OpList(); pl.append(op, *keep, acc)
    pl.run(); torch.cuda.synchronize()
    e = relerr(out, ref)
    assert torch.isfinite(out.float()).all() and e <= 3e-3, f"rowchain vs fp32: {e:.3e}"
    # bit-repeatable (no atomics on the data path; the statistics are integer atomics)
    a1, o1 = acc.clone(), out.clone()
    acc.zero_(); out.zero_()
    pl.run(); torch.cuda.synchronize()
    assert torch.equal(out, o1) and torch.equal(acc, a1)
    # GroupNorm statistics of what was stored
    o = out.float().cpu().view(B, T, C).double()
    s1, s2 = o.view(B, T, G, C // G).sum((1, 3)), (o ** 2).view(B, T, G, C // G).sum((1, 3))
    a0 = a1.cpu().double()
    assert (a0[0, :, :, 0] / 2 ** 20 - s1).abs().max() <= 1e-3 * max(1.0, s1.abs().max().item())
    assert (a0[0, :, :, 1] / 2 ** 12 - s2).abs().max() <= 1e-3 * s2.abs().max().item()
    full = torch.zeros(B, T, 2 * C, dtype=torch.float64); full[:, :, C:] = o
    t1, t2 = full.view(B, T, G, 2 * C // G).sum((1, 3)), (full ** 2).view(B, T, G, 2 * C // G).sum((1, 3))
    assert (a0[1, :, :, 0] / 2 ** 20 - t1).abs().max() <= 1e-3 * max(1.0, t1.abs().max().item())
    assert (a0[1, :, :, 1] / 2 ** 12 - t2).abs().max() <= 1e-3 * max(1.0, t2.abs().max().item())
    # the four launches it replaces: row GEMM (+ residual), row GEMM with LayerNorm prologue + GEGLU, implicit GEMM (+ residual),
    # row GEMM (+ residual)
    h2 = torch.empty(M, C, dtype=torch.float16, device=DEV); hid = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    h3 = torch.empty(M, C, dtype=torch.float16, device=DEV); o4 = torch.empty(M, C, dtype=torch.float16, device=DEV)
    w2l = L.pack_linear(d["w2"])
    un = _lib.OpList()
    for opk in (L.rowgemm(ad, pk["w_out"], h2, M=M, K=C, Nout=C, ldx=C, ldo=C, bias=pk["b_out"], res=r1d, ldr=C),
                L.rowgemm(h2, pk["w_ff1"], hid, M=M, K=C, Nout=8 * C, ldx=C, ldo=4 * C, bias=pk["b_ff1"], pro=1, eps=1e-5, epi=1),
                L.igemm(hid, w2l, h3, M=M, Nout=C, C1=4 * C, ldx1=4 * C, CinP=w2l.shape[1], ldo=C, bias=d["b2"], res=h2, ldr=C, tile=2, variant=1),
                L.rowgemm(h3, pk["w_po"], o4, M=M, K=C, Nout=C, ldx=C, ldo=C, bias=pk["b_po"], res=r2d, ldr=C)):
        un.append(opk[0], *opk[1])
    un.run(); torch.cuda.synchronize()
    e2 = relerr(out, o4)
    assert torch.equal(out, o4), f"rowchain vs the four separate launches: not bit-identical (rel-L2 {e2:.3e})"
    t_chain, t_un = pl.time_ms(20), un.time_ms(20)
    print(f"B {B} T {T}: rowchain {1e3 * t_chain:.1f} us, four launches {1e3 * t_un:.1f} us (warm replay); vs fp32 {e:.2e}, vs unfused {e2:.2e}")


def test_rowchain_strided_operands_and_rejects(L, layers):
    """operands that are column slices of wider buffers (ld > C); argument validation"""
    from live2diff_amd import _lib
    w, d, pk = layers
    M = 256
    A, R1, R2 = rnd(M, 3 * C, seed=21).to(DEV), rnd(M, 2 * C, seed=22).to(DEV), rnd(M, C + 64, seed=23).to(DEV)
    O = torch.zeros(M, 2 * C, dtype=torch.float16, device=DEV)
    a, r1, r2 = A[:, C:2 * C], R1[:, :C], R2[:, :C]
    op, keep = L.rowchain(A, R1, R2, O, M=M, C=C, lda=3 * C, ldr1=2 * C, ldr2=C + 64, ldo=2 * C, **pk)
    op.p[0] = A.data_ptr() + 2 * C               # (column offset C of the 3C-wide buffer)
    op.p[3] = O.data_ptr() + 2 * C
    L.run((op, keep)); torch.cuda.synchronize()
    ref = reference(w, a.cpu(), r1.cpu(), r2.cpu())
    assert relerr(O[:, C:], ref) <= 3e-3 and float(O[:, :C].abs().max()) == 0.0
    out = torch.empty(M, C, dtype=torch.float16, device=DEV)
    x = rnd(M, C, seed=24).to(DEV)
    with pytest.raises(_lib.L2DError):
        L.run(L.rowchain(x, x, x, out, M=M - 16, C=C, **pk))                                   # M % 32
    with pytest.raises(_lib.L2DError):
        L.run(L.rowchain(x, x, x, out, M=M, C=C, ldo=C + 4, **pk))                             # ldo % 8
    op, keep = L.rowchain(x, x, x, out, M=M, C=C, **pk)
    assert not L.gn_target(op, out.data_ptr(), T=48, G=32, cpg=10, choff=0)                    # T % 32
    op.i[1] = 640
    with pytest.raises(_lib.L2DError):
        L.run((op, keep))                                                                      # C = 320 only


# ----------------------------------------------------------------------------------------------- head segments (two layers per launch)
@pytest.mark.parametrize("kind", ["gn_qkv", "res_qkv", "gn_qkvT", "res_q"])
def test_rowchain_head_segments(L, kind):
    """h = A(norm?(x)) + bA (+ res) -> stored; out = B(LayerNorm(h)) (+ bB): the four shapes of the plan -- proj_in behind the block's
    GroupNorm + q | k | v (motion) / q | k + V^T (spatial), to_out + residual + q | k | v (motion) / + the cross-attention query
    (spatial) -- against fp32 torch and against the two row-GEMM launches each one replaces (the same rounding points)."""
    from live2diff_amd import _lib
    B, T, G = 2, 4096, 32
    M = B * T
    gnp, passes, trl = kind.startswith("gn"), (1 if kind == "res_q" else 3), kind == "gn_qkvT"
    x0, res = rnd(M, C, seed=31), rnd(M, C, seed=32)
    wa, ba = rnd(C, C, seed=33, scale=C ** -0.5), rnd(C, seed=34, scale=0.1).float()
    ga, bta = (1 + 0.2 * rnd(C, seed=35).float()).half(), (0.2 * rnd(C, seed=36).float()).half()      # the GroupNorm's affine (folded into A)
    wb = rnd(passes * C, C, seed=37, scale=C ** -0.5)
    gl, btl = (1 + 0.2 * rnd(C, seed=38).float()).half(), (0.2 * rnd(C, seed=39).float()).half()      # the LayerNorm's affine (folded into B)
    f = lambda t: t.float()
    # x arrives through an identity igemm that accumulates its GroupNorm statistics (as every producer in the plan does)
    x = torch.empty(M, C, dtype=torch.float16, device=DEV)
    acc = torch.zeros(B, G, 2, dtype=torch.int64, device=DEV)
    wi = L.pack_linear(torch.eye(C).half().to(DEV))
    op, keep = L.igemm(x0.to(DEV), wi, x, M=M, Nout=C, C1=C, ldx1=C, CinP=wi.shape[1], ldo=C, tile=2, variant=1)
    assert L.gn_target(op, acc.data_ptr(), T=T, G=G, cpg=C // G, choff=0)
    L.run((op, keep + (acc,)))
    if gnp:
        xn = F.group_norm(f(x0).view(B, T, C).permute(0, 2, 1), G, f(ga), f(bta), 1e-6).permute(0, 2, 1).reshape(M, C)
        h = (xn @ f(wa).t() + ba).half().float()
        wpa, bpa = L.pack_rowgemm(wa.to(DEV), ba.to(DEV), ga.to(DEV), bta.to(DEV))
    else:
        h = ((f(x0) @ f(wa).t() + ba).half().float() + f(res)).half().float()
        wpa, bpa = L.pack_rowgemm(wa.to(DEV), ba.to(DEV))
    ref = F.layer_norm(h, (C,), f(gl), f(btl), 1e-5) @ f(wb).t()
    wpb, bpb = L.pack_rowgemm(wb.to(DEV), None, gl.to(DEV), btl.to(DEV))
    ldvt = T
    hout = torch.zeros(M, C, dtype=torch.float16, device=DEV)
    ncol = (passes - (1 if trl else 0)) * C
    out = torch.zeros(M, ncol, dtype=torch.float16, device=DEV)
    vt = torch.zeros(B, C, ldvt, dtype=torch.float16, device=DEV) if trl else None
    kw = dict(M=M, C=C, wA=wpa, bA=bpa, wB=wpb, bB=bpb, passes=passes, T=T, G=G, eps_gn=1e-6, eps_ln=1e-5, ldo=ncol)
    if gnp:
        kw.update(gn_acc_ptr=acc.data_ptr())
    else:
        kw.update(resA=res.to(DEV))
    if trl:
        kw.update(out_t=vt, ldt=ldvt, st=C * ldvt)
    pl = _lib.OpList(); pl.append(*L.rowchain_head(x, hout, out, **kw))
    pl.run(); torch.cuda.synchronize()
    eh, eo = relerr(hout, h), relerr(out, ref[:, :ncol])
    assert eh <= 2e-3 and eo <= 3e-3, (kind, eh, eo)
    if trl:
        ev = relerr(vt.permute(0, 2, 1).reshape(M, C), ref[:, 2 * C:])
        assert ev <= 3e-3, (kind, ev)
    # the two launches it replaces
    h2 = torch.zeros(M, C, dtype=torch.float16, device=DEV)
    o2 = torch.zeros(M, ncol, dtype=torch.float16, device=DEV)
    vt2 = torch.zeros(B, C, ldvt, dtype=torch.float16, device=DEV) if trl else None
    un = _lib.OpList()
    if gnp:
        un.append(*L.rowgemm(x, wpa, h2, M=M, K=C, Nout=C, ldx=C, ldo=C, bias=bpa, pro=2, eps=1e-6, T=T, G=G, gn_acc_ptr=acc.data_ptr()))
    else:
        un.append(*L.rowgemm(x, wpa, h2, M=M, K=C, Nout=C, ldx=C, ldo=C, bias=bpa, res=res.to(DEV), ldr=C))
    un.append(*L.rowgemm(h2, wpb, o2, M=M, K=C, Nout=passes * C, ldx=C, ldo=ncol, bias=bpb, pro=1, eps=1e-5, T=T,
                         **(dict(out_t=vt2, ntr=C, ldt=ldvt, st=C * ldvt) if trl else {})))
    un.run(); torch.cuda.synchronize()
    assert torch.equal(hout, h2) and torch.equal(out, o2), (kind, relerr(hout, h2), relerr(out, o2))     # bit-identical to the two launches
    if trl:
        assert torch.equal(vt, vt2)
    t_c, t_u = pl.time_ms(20), un.time_ms(20)
    print(f"{kind}: head segment {1e3 * t_c:.1f} us, two launches {1e3 * t_u:.1f} us (warm replay); h {eh:.2e} out {eo:.2e}; vs unfused h {relerr(hout, h2):.1e} out {relerr(out, o2):.1e}")
