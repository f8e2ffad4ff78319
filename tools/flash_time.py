"""HIP-event timing of the flash-attention geometries on the frame's shapes (B = 2 x 8 heads):  python tools/flash_time.py 2 6"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops                                             # noqa: E402

DEV = "cuda"
variants = [int(v) for v in sys.argv[1:]] or [2, 6]
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float16)
N = 2
for (dd, TT, Tk) in ((40, 4096, 4096), (40, 4096, 77), (40, 6144, 6144), (40, 9216, 9216), (80, 1024, 1024)):
    CC = 8 * dd
    q_, k_ = rn(N * TT, CC), rn(N * Tk, CC)
    ld = (Tk + 7) // 8 * 8
    vt_ = rn(N, CC, ld)
    row = []
    for v in variants:
        o_ = torch.empty(N * TT, CC, dtype=torch.float16, device=DEV)
        pl = _lib.OpList()
        pl.append(*ops.flash_attn(q_, k_, vt_, o_, B=N, H=8, d=dd, Tq=TT, Tk=Tk, ldq=CC, ldk=CC, ldvt=ld, ldo=CC, sq=TT * CC, sk=Tk * CC,
                                  svt=CC * ld, so=TT * CC, variant=v))
        for _ in range(3):
            pl.run()
        torch.cuda.synchronize()
        ms = pl.time_ms(reps=20)
        fl = 4.0 * N * 8 * TT * Tk * dd
        row.append(f"v{v}: {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF")
    print(f"d{dd} Tq{TT} Tk{Tk}:  " + "   ".join(row))
