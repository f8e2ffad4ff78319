"""HIP-event timing of the level-0 block-tail chain (no checks: usable with the analysis builds of tools/variant_libs.sh):
    L2D_LIB=live2diff_amd/ablate/libl2d_rowchain_RC_X4.so python tools/rowchain_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops as L                                        # noqa: E402

DEV, C = "cuda", 320
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).half()
d = dict(wo=rnd(C, C, scale=C ** -0.5), bo=rnd(C, scale=0.1).float(), gm=(1 + 0.2 * rnd(C).float()).half(), bt=(0.2 * rnd(C).float()).half(),
         w1=rnd(8 * C, C, scale=C ** -0.5), b1=rnd(8 * C, scale=0.1).float(), w2=rnd(C, 4 * C, scale=(4 * C) ** -0.5), b2=rnd(C, scale=0.1).float(),
         wp=rnd(C, C, scale=C ** -0.5), bp=rnd(C, scale=0.1).float())
d = {k: v.to(DEV) for k, v in d.items()}
pk = dict(zip(("w_out", "b_out"), L.pack_rowgemm(d["wo"], d["bo"])))
pk.update(zip(("w_ff1", "b_ff1"), L.pack_rowgemm(d["w1"], d["b1"], d["gm"], d["bt"], geglu=True)))
pk.update(zip(("w_ff2", "b_ff2"), L.pack_rowgemm(d["w2"], d["b2"])))
pk.update(zip(("w_po", "b_po"), L.pack_rowgemm(d["wp"], d["bp"])))
row = []
NCOLD = 120                                       # rotating copies of the 2.9 MB of packed weights: 350 MB, beyond the 256 MB Infinity Cache
cold = [{k: (v.clone() if k.startswith("w_") else v) for k, v in pk.items()} for _ in range(NCOLD)]
for M in (8192, 6144, 12288):
    a, r1, r2 = (rnd(M, C).to(DEV) for _ in range(3))
    out = torch.zeros(M, C, dtype=torch.float16, device=DEV)
    op, keep = L.rowchain(a, r1, r2, out, M=M, C=C, eps=1e-5, **pk)
    pl = _lib.OpList(); pl.append(op, *keep)
    pl.run(); torch.cuda.synchronize()
    warm = 1e3 * min(pl.time_ms(20) for _ in range(3))
    plc = _lib.OpList()
    for c in cold:
        plc.append(*L.rowchain(a, r1, r2, out, M=M, C=C, eps=1e-5, **c))
    plc.run(); torch.cuda.synchronize()
    row.append(f"M{M}: {warm:6.1f} us warm {1e3 * min(plc.time_ms(2) for _ in range(3)) / NCOLD:6.1f} us cold weights")
print(os.environ.get("L2D_LIB", "product"), "  ".join(row))
