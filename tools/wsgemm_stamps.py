"""Where does a wsgemm block spend its life?  Needs the analysis build of the library (in-kernel s_memtime stamps):
    make -C live2diff_amd/csrc PROBES=1 LIB=../libl2d_hip_probes.so
    L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/wsgemm_stamps.py
Per (shape, schedule): median shader cycles of a block's phases -- loader wave 0: descriptors, first requests, then per stage
(landed, barrier passed); consumer wave 0: ring request, first barrier, per stage; loop end, arrival, epilogue -- with COLD
weights (a different weight copy per launch, > 256 MB in rotation), and the launch's HIP-event duration."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops as L                                        # noqa: E402

DEV = "cuda"
CASES = [  # (name, kind, M, K (or C), N, pro, epi, sched)
    ("GEGLU M128", "lin", 128, 1280, 10240, 1, 1, (4, 1, 2, 1, True)),
    ("GEGLU M128 S2", "lin", 128, 1280, 10240, 1, 1, (4, 1, 2, 2, True)),
    ("GEGLU M512", "lin", 512, 1280, 10240, 1, 1, (4, 1, 2, 1, False)),
    ("GEGLU M512 w5", "lin", 512, 1280, 10240, 1, 1, (5, 1, 2, 1, False)),
    ("FF2 M512", "lin", 512, 5120, 1280, 0, 0, (2, 1, 2, 3, False)),
    ("conv M128", "conv", 128, 1280, 1280, 0, 0, (2, 1, 2, 12, True)),
    ("conv M128 BN128", "conv", 128, 1280, 1280, 0, 0, (4, 1, 2, 12, True)),
    ("conv M512", "conv", 512, 1280, 1280, 0, 0, (4, 1, 2, 6, False)),
    ("linear M512", "lin", 512, 1280, 1280, 0, 0, (2, 1, 2, 1, False)),
    ("linear M128", "lin", 128, 1280, 1280, 0, 0, (1, 1, 2, 4, True)),
]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16)


def main():
    _lib.lib.l2d_wsgemm_set_probe.argtypes = [ctypes.c_void_p]
    cnt = torch.zeros(1 << 14, dtype=torch.int32, device=DEV)
    for name, kind, M, K, N, pro, epi, sched in CASES:
        NW, NT, NL, S, ntw = sched
        taps = 9 if kind == "conv" else 1
        Ktot = taps * K
        x = rnd(M, K, seed=1).to(DEV)
        No = N // 2 if epi else N
        out = torch.zeros(M, No, dtype=torch.float16, device=DEV)
        b = rnd(N, seed=4).float().to(DEV)
        if kind == "conv":
            w = rnd(N, K, 3, 3, seed=3, scale=Ktot ** -0.5).to(DEV)
            wp, bp, cs = L.pack_wsgemm_conv3x3(w), b, None
        else:
            w = rnd(N, K, seed=3, scale=K ** -0.5).to(DEV)
            gm = torch.ones(K, dtype=torch.float16, device=DEV) if pro else None
            wp, bp, cs = L.pack_wsgemm(w, b, gm, (torch.zeros(K, dtype=torch.float16, device=DEV) if pro else None), geglu=bool(epi))
        R = max(2, -(-300 * (1 << 20) // (wp.numel() * 2)))
        wps = [wp] + [wp.clone() for _ in range(R - 1)]
        kw = {}
        if S > 1:
            n_ws, n_cnt = L.wsgemm_sizes(M, N, NW, NT, S)
            kw = dict(ws=torch.empty(n_ws, dtype=torch.float32, device=DEV), cnt=cnt)
        H = int(round((M // 2) ** 0.5))
        pl = _lib.OpList()
        for r in range(R):
            op, keep = L.wsgemm(x, wps[r], out, M=M, Nout=N, C1=K, ldx1=K, ldo=No, bias=bp, colsum=cs, taps=taps, B=2, H=H, W=H, epi=epi, pro=pro,
                                T=M // 2, sched=sched, **kw)
            pl.append(op, *keep)
        nblk = -(-M // 128) * (N // 32 // (NW * NT)) * S
        probe = torch.zeros(nblk * 64, dtype=torch.int64, device=DEV)
        pl.run()
        torch.cuda.synchronize()
        us = pl.time_ms(3) * 1000.0 / R
        _lib.lib.l2d_wsgemm_set_probe(ctypes.c_void_p(probe.data_ptr()))
        pl.run()                                     # the LAST launch's stamps survive (weights cold: R copies in rotation)
        torch.cuda.synchronize()
        _lib.lib.l2d_wsgemm_set_probe(None)
        p = probe.view(nblk, 64).cpu().double()
        p[p == 0] = float("nan")
        t0 = torch.minimum(p[:, 0], p[:, 32]).unsqueeze(1)          # per block: first stamp of either wave
        d = p - t0
        n = (Ktot // 64) // S
        med = lambda col: float(d[:, col].nanmedian())
        first = float(torch.nan_to_num(t0, nan=float("inf")).min())
        span = float(torch.nan_to_num(p[:, [28, 60, 62, 63]], nan=0.0).max() - first)
        entry_spread = float(torch.nan_to_num(t0, nan=0.0).max() - first)
        print(f"\n== {name} {sched}: {us:.1f} us/launch cold; {nblk} blocks, {n} stages per block; launch span {span:.0f} cycles "
              f"(block entries spread over {entry_spread:.0f}); cycles since the block's entry, medians over blocks:")
        print(f"   loader:   entry {med(0):7.0f}  descriptors {med(1):7.0f}  first requests out {med(2):7.0f}  done {med(28):7.0f}")
        print("   loader stage landed / barrier passed: " + "  ".join(f"{med(3 + 2 * s):.0f}/{med(4 + 2 * s):.0f}" for s in range(min(n, 12))))
        print(f"   consumer: entry {med(32):7.0f}  ring requested {med(33):7.0f}  first barrier {med(34):7.0f}")
        print("   consumer stage done: " + "  ".join(f"{med(35 + 2 * s):.0f}" for s in range(min(n, 12))))
        last = d[:, 59] > 0                              # blocks that ran the epilogue (all of them without split-K)
        lmed = lambda col: float(d[last][:, col].nanmedian()) if last.any() else float("nan")
        print(f"   loop done {med(60):7.0f}  all waves {med(61):7.0f}  arrival known {med(62):7.0f}   | reducing blocks: arrival {lmed(62):7.0f}  "
              f"epilogue start {lmed(59):7.0f}  tile staged {lmed(57):7.0f}  all staged {lmed(58):7.0f}  rows stored {lmed(63):7.0f}")


if __name__ == "__main__":
    main()
