#!/usr/bin/env python
"""Runs a handful of representative hot-path launches (cfg-2 shapes) a few times each -- the target of the
rocprofv3 PMC passes (`rocprofv3 --pmc ... -- python tools/prof_ops.py`).  Each op uses distinct buffers per
repetition for the HBM-bound kernel so that the 256 MB Infinity Cache cannot serve the KV-cache reads."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from live2diff_amd import _lib, ops  # noqa: E402

DEV = "cuda"
REPS = int(os.environ.get("L2D_PROF_REPS", "6"))
VARIANT = int(os.environ.get("L2D_IGEMM_VARIANT", "5"))


def run(opk, variant=None):
    op, keep = opk
    if variant is not None:
        op.i[23] = variant
    pl = _lib.OpList()
    pl.append(op, *keep)
    pl.run()


ONLY = os.environ.get("L2D_PROF_ONLY", "")      # e.g. "flash": profile only the flash-attention launches


def main():
    if ONLY == "flash":
        return flash_only()
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float16)
    # --- conv 64x64 320->320 (level-0 resnet conv), 64x64 tile
    B, H, W, cin, cout = 2, 64, 64, 320, 320
    x = rn(B * H * W, cin)
    wp = ops.pack_conv3x3(rn(cout, cin, 3, 3) * (9 * cin) ** -0.5)
    out = torch.empty(B * H * W, cout, dtype=torch.float16, device=DEV)
    bias = torch.zeros(cout, device=DEV)
    xres = rn(B * H * W, cout)
    for tile in (2, 1):
        for _ in range(REPS):
            run(ops.igemm(x, wp, out, M=B * H * W, Nout=cout, C1=cin, ldx1=cin, CinP=wp.shape[1] // 9, ldo=cout, bias=bias, taps=9,
                          B=B, Hin=H, Win=W, Hout=H, Wout=W, tile=tile), VARIANT)
    # --- GEGLU projection M8192 N2560 K320, 128x128 tile
    w1, b1 = ops.pack_geglu(rn(2560, 320) * 320 ** -0.5, torch.zeros(2560, device=DEV))
    o1 = torch.empty(8192, 1280, dtype=torch.float16, device=DEV)
    for _ in range(REPS):
        run(ops.igemm(x, w1, o1, M=8192, Nout=2560, C1=320, ldx1=320, CinP=320, ldo=1280, bias=b1, epi=1, tile=1), VARIANT)
    # --- patch-resident 3x3 conv (pconv.hip): level-0 320 -> 320 and the concat conv 640 -> 320, 8x16 patches
    for cin_ in (320, 640):
        xx = rn(B * H * W, cin_)
        wq = ops.pack_conv3x3(rn(cout, cin_, 3, 3) * (9 * cin_) ** -0.5)
        for _ in range(REPS):
            run(ops.pconv(xx, wq, out, B=B, H=H, W=W, C1=cin_, ldx1=cin_, CinP=cin_, Nout=cout, ldo=cout, patch=(8, 16), bias=bias))
    # --- token-row GEMM (rowgemm.hip), level 0: LayerNorm + GEGLU (M8192 N2560 K320), LayerNorm + q|k|V^T (N960), plain C x C
    gam, bet = torch.ones(320, device=DEV), torch.zeros(320, device=DEV)
    rw1, rb1 = ops.pack_rowgemm(rn(2560, 320) * 320 ** -0.5, torch.zeros(2560, device=DEV), gam, bet, geglu=True)
    for _ in range(REPS):
        run(ops.rowgemm(x, rw1, o1, M=8192, K=320, Nout=2560, ldx=320, ldo=1280, bias=rb1, epi=1, pro=1))
    rwq, rbq = ops.pack_rowgemm(rn(960, 320) * 320 ** -0.5, None, gam, bet)
    oqk = torch.empty(8192, 640, dtype=torch.float16, device=DEV)
    ovt = torch.empty(2, 320, 4096, dtype=torch.float16, device=DEV)
    for _ in range(REPS):
        run(ops.rowgemm(x, rwq, oqk, M=8192, K=320, Nout=960, ldx=320, ldo=640, bias=rbq, pro=1, T=4096, out_t=ovt, ntr=320, ldt=4096,
                        st=320 * 4096))
    rwc, rbc = ops.pack_rowgemm(rn(320, 320) * 320 ** -0.5, torch.zeros(320, device=DEV))
    for _ in range(REPS):
        run(ops.rowgemm(x, rwc, out, M=8192, K=320, Nout=320, ldx=320, ldo=320, bias=rbc, res=xres, ldr=320))
    # --- streaming temporal attention, level 0: N=2, T=4096, C=320, L=16; a fresh 168 MB cache per repetition
    N, T, C, L = 2, 4096, 320, 16
    qkv = rn(N * T, 3 * C)
    tabs = [rn(L, C) for _ in range(3)]
    pe_idx = torch.arange(L, device=DEV).repeat(N, 1)
    upd = torch.tensor([12, 13], device=DEV)
    tbias = torch.zeros(N, L, dtype=torch.float16, device=DEV)
    caches = [rn(N, 2, T, L, C) for _ in range(REPS)]
    ao = torch.empty(N * T, C, dtype=torch.float16, device=DEV)
    for c in caches:
        run(ops.tattn_stream(qkv, c, tabs[0], tabs[1], tabs[2], pe_idx, upd, tbias, ao, N=N, T=T, C=C, L=L, H=8))
    # --- flash attention level 0: d=40, T=4096
    d, Hh = 40, 8
    qk = rn(N * T, 2 * C)
    vt = rn(N, C, T)
    fo = torch.empty(N * T, C, dtype=torch.float16, device=DEV)
    for _ in range(REPS):
        run(ops.flash_attn(qk, qk, vt, fo, B=N, H=Hh, d=d, Tq=T, Tk=T, ldq=2 * C, ldk=2 * C, ldvt=T, ldo=C, sq=T * 2 * C,
                           sk=T * 2 * C, svt=C * T, so=T * C, k_off=C))
    # --- flash attention, the other frame shapes: d=80 T=1024, d=160 T=256, text cross-attention d=40 4096x77
    for (dd, TT, Tk) in ((80, 1024, 1024), (160, 256, 256), (40, 4096, 77)):
        CC = 8 * dd
        q_ = rn(N * TT, CC)
        k_ = rn(N * Tk, CC)
        ld = (Tk + 7) // 8 * 8
        vt_ = rn(N, CC, ld)
        o_ = torch.empty(N * TT, CC, dtype=torch.float16, device=DEV)
        for _ in range(REPS):
            run(ops.flash_attn(q_, k_, vt_, o_, B=N, H=8, d=dd, Tq=TT, Tk=Tk, ldq=CC, ldk=CC, ldvt=ld, ldo=CC, sq=TT * CC, sk=Tk * CC,
                               svt=CC * ld, so=TT * CC))
    # --- GroupNorm level 0
    gm, bt = torch.ones(C, dtype=torch.float16, device=DEV), torch.zeros(C, dtype=torch.float16, device=DEV)
    part = torch.empty(2 * 128 * 32 * 2, dtype=torch.float32, device=DEV)
    go = torch.empty(N * T, C, dtype=torch.float16, device=DEV)
    kw = dict(B=N, T=T, C1=C, ld1=C, G=32, nchunk=128)
    for _ in range(REPS):
        run(ops.gn_stats(x, part, **kw))
        run(ops.gn_apply(x, part, gm, bt, go, eps=1e-5, silu=True, **kw))
    torch.cuda.synchronize()
    print("prof_ops done")


def flash_only():
    """self-attention of the three levels + the level-0 text cross-attention, one kernel name per shape where possible"""
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float16)
    N = 2
    for (dd, TT, Tk) in ((40, 4096, 4096), (80, 1024, 1024), (160, 256, 256)):
        CC = 8 * dd
        q_, k_ = rn(N * TT, CC), rn(N * Tk, CC)
        ld = (Tk + 7) // 8 * 8
        vt_ = rn(N, CC, ld)
        o_ = torch.empty(N * TT, CC, dtype=torch.float16, device=DEV)
        for _ in range(REPS):
            run(ops.flash_attn(q_, k_, vt_, o_, B=N, H=8, d=dd, Tq=TT, Tk=Tk, ldq=CC, ldk=CC, ldvt=ld, ldo=CC, sq=TT * CC, sk=Tk * CC,
                               svt=CC * ld, so=TT * CC))
    torch.cuda.synchronize()
    print("prof_ops flash done")


if __name__ == "__main__":
    main()
