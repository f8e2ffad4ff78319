"""Where does a cconv block spend its life?  Needs the analysis build of the library (in-kernel s_memtime stamps):
    make -C live2diff_amd/csrc PROBES=1 LIB=../libl2d_hip_probes.so
    L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/cconv_stamps.py
Per (shape, schedule): median shader cycles (since the block's first stamp) of compute wave 0's phases and loader 0's, with cold weights
(rotating copies), plus the launch's in-chain duration."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops as L                                        # noqa: E402

DEV = "cuda"
CASES = [  # name, B, H, W, Cin, Cout, ups, sched
    ("L1 640->640", 2, 32, 32, 640, 640, 0, (2, 2, 2, 3)),
    ("L1 640->640 S1", 2, 32, 32, 640, 640, 0, (2, 2, 2, 1)),
    ("L1 1920->640", 2, 32, 32, 1920, 640, 0, (2, 2, 2, 3)),
    ("L2 1280->1280", 2, 16, 16, 1280, 1280, 0, (2, 2, 2, 6)),
    ("L2 1280->1280 KG4", 2, 16, 16, 1280, 1280, 0, (1, 4, 4, 3)),
    ("L1 640->640 KG4 S1", 2, 32, 32, 640, 640, 0, (1, 4, 4, 1)),
    ("L0 320->320", 2, 64, 64, 320, 320, 0, (1, 4, 4, 1)),
    ("UP2", 2, 32, 32, 1280, 1280, 1, (2, 2, 2, 1)),
]
NAMES = {0: "entry", 1: "ring requested", 2: "chunk 0 landed", 3: "loop done", 4: "K groups met", 5: "slab out + arrival", 6: "slabs summed",
         7: "tile staged", 8: "rows stored", 10: "L entry", 11: "L chunk0 issued", 12: "L chunk0 landed", 13: "L done"}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(torch.float16)


def main():
    _lib.lib.l2d_cconv_set_probe.argtypes = [ctypes.c_void_p]
    cnt = torch.zeros(1 << 14, dtype=torch.int32, device=DEV)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, B, H, W, Cin, Cout, ups, sched in CASES:
        if only and only not in name:
            continue
        CG, KG, NLD, S = sched
        M = B * H * W
        x = rnd(B, H >> ups, W >> ups, Cin, seed=1)
        w = rnd(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
        wp = L.pack_cconv(w, KG)
        R = max(2, -(-300 * (1 << 20) // (wp.numel() * 2)))
        wps = [wp] + [wp.clone() for _ in range(R - 1)]
        bias = rnd(Cout, seed=3).float()
        res = rnd(M, Cout, seed=5)
        out = torch.empty(M, Cout, dtype=torch.float16, device=DEV)
        n_ws, n_cnt = L.cconv_sizes(B, H, W, Cout, CG, S)
        ws = torch.empty(max(1, n_ws), dtype=torch.float32, device=DEV) if S > 1 else None
        nblk = (n_cnt // 2) * S
        probe = torch.zeros(nblk * 32, dtype=torch.int64, device=DEV)
        pl = _lib.OpList()
        for k in range(R):
            op, keep = L.cconv(x, wps[k], out, B=B, H=H, W=W, C1=Cin, ldx1=Cin, Nout=Cout, ldo=Cout, KG=KG, ups=ups, bias=bias, res=res, ldr=Cout,
                               sched=sched, ws=ws, cnt=(cnt if S > 1 else None))
            pl.append(op, *keep)
        _lib.lib.l2d_cconv_set_probe(None)
        pl.run(); torch.cuda.synchronize()
        us = sorted(pl.time_each_us(reps=3)[1:])
        # one stamped launch behind the chain (cold weights of copy 0)
        _lib.lib.l2d_cconv_set_probe(ctypes.c_void_p(probe.data_ptr()))
        pl2 = _lib.OpList()
        for k in (R - 1, 0):
            op, keep = L.cconv(x, wps[k], out, B=B, H=H, W=W, C1=Cin, ldx1=Cin, Nout=Cout, ldo=Cout, KG=KG, ups=ups, bias=bias, res=res, ldr=Cout,
                               sched=sched, ws=ws, cnt=(cnt if S > 1 else None))
            pl2.append(op, *keep)
        probe.zero_()
        pl2.run(); torch.cuda.synchronize()
        _lib.lib.l2d_cconv_set_probe(None)
        p = probe.view(nblk, 32).cpu()
        t0 = p[:, 0:1]
        rel = (p - t0).double()
        rel[p == 0] = float("nan")
        print(f"\n== {name} {sched}: {us[len(us) // 2]:.1f} us/launch in chain; {nblk} blocks; block entries spread over {int(p[:, 0].max() - p[:, 0].min())} cycles; "
              f"launch span {int(p[:, 8][p[:, 8] > 0].max() - p[:, 0].min())} cycles")
        def med(i):
            v = rel[:, i]; v = v[~torch.isnan(v)]
            return float("nan") if v.numel() == 0 else float(v.median())
        print("   compute: " + "  ".join(f"{NAMES[i]} {med(i):.0f}" for i in (1, 2, 3, 4, 5, 6, 7, 8)))
        print("   chunk done: " + " ".join(f"{med(16 + c):.0f}" for c in range(8)))
        print("   loader:  " + "  ".join(f"{NAMES[i]} {med(i):.0f}" for i in (10, 11, 12, 13)) + "   chunk c+1 landed: " + " ".join(f"{med(24 + c):.0f}" for c in range(6)) + "   chunk c+2 issued (c=1..3): " + " ".join(f"{med(13 + c):.0f}" for c in (1, 2, 3)))
        last = p[:, 6] > 0
        if S > 1 and last.any():
            r2 = rel[last]
            print(f"   last arrivers ({int(last.sum())}): arrival {float(r2[:, 5].median()):.0f}  summed {float(r2[:, 6].median()):.0f}  staged {float(r2[:, 7].median()):.0f}  stored {float(r2[:, 8].median()):.0f}")


if __name__ == "__main__":
    main()
