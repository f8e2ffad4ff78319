"""Cold-weight timing of the LayerNorm + GEGLU launches under explicit schedules (NW, NT, NL, S):
    python tools/wsgemm_geglu_time.py            (M, C) = (2048, 640), (512, 1280), (128, 1280)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops as L                                        # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).half()
CASES = ((2048, 640, [(5, 1, 2, 1), (10, 1, 2, 1), (8, 1, 2, 1), (4, 1, 2, 1)]), (512, 1280, [(5, 1, 2, 1), (10, 1, 2, 1), (8, 1, 2, 1), (4, 1, 2, 1)]),
         (128, 1280, [(2, 1, 2, 1), (5, 1, 2, 1), (10, 1, 2, 1)]), (3072, 640, [(5, 1, 2, 1), (10, 1, 2, 1)]), (4608, 640, [(5, 1, 2, 1), (10, 1, 2, 1)]))
if os.environ.get("SPLITK"):      # K slices beside the ten-wave form at the few-token levels
    CASES = ((512, 1280, [(5, 1, 2, 1), (10, 1, 2, 2), (5, 1, 2, 2), (10, 1, 2, 4), (8, 1, 2, 2)]), (128, 1280, [(2, 1, 2, 1), (10, 1, 2, 4), (10, 1, 2, 8), (5, 1, 2, 4), (4, 1, 2, 2)]),
             (768, 1280, [(8, 1, 2, 1), (10, 1, 2, 1), (10, 1, 2, 2)]), (1024, 1280, [(10, 1, 2, 1), (10, 1, 2, 2), (5, 1, 2, 1)]))
for (M, C, scheds) in CASES:
    ncopy = max(2, int(300e6 // (8 * C * C * 2)) + 1)          # rotating weight copies: cold like in the frame
    w, b = rnd(8 * C, C, scale=C ** -0.5).to(DEV), rnd(8 * C, scale=0.1).float().to(DEV)
    gm, bt = (1 + 0.2 * rnd(C).float()).half().to(DEV), (0.2 * rnd(C).float()).half().to(DEV)
    wp, bp, cs = L.pack_wsgemm(w, b, gm, bt, geglu=True)
    wps = [wp.clone() for _ in range(ncopy)]
    x = rnd(M, C).to(DEV)
    out = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    row = []
    for sched in scheds:
        pl = _lib.OpList()
        kw = {}
        if sched[3] > 1:
            nws, ncnt = L.wsgemm_sizes(M, 8 * C, sched[0], sched[1], sched[3])
            kw = dict(ws=torch.zeros(nws, dtype=torch.float32, device=DEV), cnt=torch.zeros(ncnt + 4, dtype=torch.int32, device=DEV))
        for wc in wps:
            pl.append(*L.wsgemm(x, wc, out, M=M, Nout=8 * C, C1=C, ldx1=C, ldo=4 * C, bias=bp, colsum=cs, epi=1, pro=1, sched=sched + (M <= 128,), **kw))
        pl.run(); torch.cuda.synchronize()
        us = min(pl.time_ms(3) for _ in range(3)) * 1e3 / ncopy
        row.append(f"{sched}: {us:6.1f} us")
    print(f"M{M} C{C}:  " + "   ".join(row))
