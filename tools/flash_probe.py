"""Where does a flash-attention wave spend a key tile?  Analysis build only (make PROBES=1, see tools/igemm_probe.py):
    L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/flash_probe.py
Stamps of key tile 8 (steady state), wave by wave: loop top -> own DMA share landed -> barrier passed -> refill issued ->
QK^T issued -> row maxima known -> exp + PV issued -> next loop top.  Median cycles over all waves of the launch."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops                                             # noqa: E402

DEV = "cuda"
_lib.lib.l2d_flash_set_probe.argtypes = [ctypes.c_void_p]
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float16)
N = 2
CASES = ((40, 4096, 2), (40, 4096, 4), (40, 2048, 2), (40, 2048, 4), (40, 4096, 3), (80, 1024, 3))   # T = 2048: one block per CU, one wave per SIMD
for (dd, TT, variant) in CASES:
    CC = 8 * dd
    q_, k_ = rn(N * TT, CC), rn(N * TT, CC)
    vt_ = rn(N, CC, TT)
    o_ = torch.empty(N * TT, CC, dtype=torch.float16, device=DEV)
    pl = _lib.OpList()
    pl.append(*ops.flash_attn(q_, k_, vt_, o_, B=N, H=8, d=dd, Tq=TT, Tk=TT, ldq=CC, ldk=CC, ldvt=TT, ldo=CC, sq=TT * CC, sk=TT * CC,
                              svt=CC * TT, so=TT * CC, variant=variant))
    for _ in range(3):
        pl.run()
    torch.cuda.synchronize()
    us = pl.time_ms(reps=20) * 1e3
    rows = 32 if variant in (2, 4) else 16
    nblk = ((TT + 4 * rows - 1) // (4 * rows)) * 8 * N
    probe = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=DEV)
    _lib.lib.l2d_flash_set_probe(ctypes.c_void_p(probe.data_ptr()))
    pl.run()
    torch.cuda.synchronize()
    _lib.lib.l2d_flash_set_probe(None)
    p = probe.view(nblk, 8, 8)[:, :4].reshape(-1, 8).cpu()
    p = p[(p > 0).all(1)]
    # order of stamps in time: 4 (top), 5 (landed), 6 (barrier), 0 (refill issued), 1 (QK issued), 2 (max known), 3 (exp+PV issued), 7 (next top)
    order = [4, 5, 6, 0, 1, 2, 3, 7]
    names = ["wait own DMA", "barrier", "issue refill", "K frags + QK^T issue", "row max", "rescale test + exp + PV issue", "loop back"]
    if variant == 4:   # pipelined loop: 4 top, 5 landed, 6 barrier, 0 refill issued, 2 K fragment reads issued, 1 phase A issued, 3 phase B + test, 7 next top
        order = [4, 5, 6, 0, 2, 1, 3, 7]
        names = ["wait own DMA", "barrier", "issue refill", "K fragment reads issued", "2 exp pairs + V^T reads + QK^T(t+1) beside exp(t)",
                 "PV(t) beside lane maxima of S(t+1) + test", "loop back"]
    t = p[:, order].double()
    d = t[:, 1:] - t[:, :-1]
    med = d.median(0).values
    print(f"d{dd} T{TT} variant {variant}: {us:.1f} us, {p.shape[0]} waves, tile period median {int((t[:, -1] - t[:, 0]).median())} cycles")
    for n_, m_, q90 in zip(names, med, d.quantile(0.9, dim=0)):
        print(f"    {n_:52s} median {int(m_):6d}   p90 {int(q90):6d}")
