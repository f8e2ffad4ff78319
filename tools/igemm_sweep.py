#!/usr/bin/env python
"""On-GPU sweep of the igemm pipeline variants / tiles / split-K factors over the UNet's representative shapes
(cfg-2).  Every configuration is also checked against the default configuration's output.

  python tools/igemm_sweep.py > gpurun_out/igemm_sweep.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from live2diff_amd import _lib, ops  # noqa: E402

DEV = "cuda"
VARIANTS = {0: "BK32x4", 1: "BK64x3", 2: "BK64x4", 3: "BK32x6", 4: "BK32x3", 5: "BK64x2"}

# (name, taps, B, H, W, Cin, Cout, epi)   linear: H=W=1, B=M
SHAPES = [
    ("lin  M8192 N320  K320", 1, 8192, 1, 1, 320, 320, 0),
    ("lin  M8192 N320  K1280", 1, 8192, 1, 1, 1280, 320, 0),
    ("lin  M8192 N960  K320", 1, 8192, 1, 1, 320, 960, 0),
    ("gegl M8192 N2560 K320", 1, 8192, 1, 1, 320, 2560, 1),
    ("lin  M2048 N640  K640", 1, 2048, 1, 1, 640, 640, 0),
    ("lin  M2048 N640  K2560", 1, 2048, 1, 1, 2560, 640, 0),
    ("gegl M2048 N5120 K640", 1, 2048, 1, 1, 640, 5120, 1),
    ("lin  M512  N1280 K1280", 1, 512, 1, 1, 1280, 1280, 0),
    ("lin  M512  N1280 K5120", 1, 512, 1, 1, 5120, 1280, 0),
    ("gegl M512  N10240 K1280", 1, 512, 1, 1, 1280, 10240, 1),
    ("lin  M128  N1280 K1280", 1, 128, 1, 1, 1280, 1280, 0),
    ("conv 64x64 320->320", 9, 2, 64, 64, 320, 320, 0),
    ("conv 64x64 640->320", 9, 2, 64, 64, 640, 320, 0),
    ("conv 32x32 640->640", 9, 2, 32, 32, 640, 640, 0),
    ("conv 16x16 1280->1280", 9, 2, 16, 16, 1280, 1280, 0),
    ("conv 16x16 2560->1280", 9, 2, 16, 16, 2560, 1280, 0),
    ("conv 8x8 1280->1280", 9, 2, 8, 8, 1280, 1280, 0),
    ("conv 8x8 2560->1280", 9, 2, 8, 8, 2560, 1280, 0),
]


def main():
    print("device:", _lib.device_name())
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, taps, B, H, W, cin, cout, epi in SHAPES:
        M = B * H * W
        x = torch.randn(M, cin, generator=g, device=DEV, dtype=torch.float16)
        if taps == 9:
            wp = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g, device=DEV) * (9 * cin) ** -0.5)
            cinp = wp.shape[1] // 9
        else:
            wp = ops.pack_linear(torch.randn(cout, cin, generator=g, device=DEV) * cin ** -0.5)
            cinp = wp.shape[1]
        bias = torch.randn(cout, generator=g, device=DEV)
        nout_cols = cout // 2 if epi == 1 else cout
        flops = 2.0 * M * cout * taps * cin
        base = None
        rows = []
        auto_tile, auto_s, auto_v = ops.igemm_schedule(M, cout, taps * cinp, 1, epi)
        cands = set()
        for tile in (1, 2):
            for s in {1, auto_s, 2, 4, 8}:
                if epi == 1 and s != 1:
                    continue
                if s > (taps * cinp) // 64 // 8:
                    continue
                cands.add((tile, s))
        cands.add((auto_tile, auto_s))
        for variant in sorted(VARIANTS):
            for tile, s in sorted(cands):
                if variant != 0 and (tile, s) not in ((auto_tile, auto_s), (1, 1), (2, 1), (1, 2)):
                    continue
                out = torch.zeros(M, nout_cols, dtype=torch.float16, device=DEV)
                ws = torch.empty(s * M * cout, dtype=torch.float32, device=DEV) if s > 1 else None
                kw = dict(M=M, Nout=cout, C1=cin, ldx1=cin, CinP=cinp, ldo=nout_cols, bias=bias, epi=epi, splitk=s, tile=tile, ws=ws)
                if taps == 9:
                    kw.update(taps=9, B=B, Hin=H, Win=W, Hout=H, Wout=W)
                op, keep = ops.igemm(x, wp, out, **kw)
                op.i[23] = variant
                pl = _lib.OpList()
                pl.append(op, *keep)
                try:
                    pl.run()
                    torch.cuda.synchronize()
                    pl.time_ms(3)
                    us = 1e3 * min(pl.time_ms(20) for _ in range(3))
                except Exception as e:  # noqa: BLE001
                    rows.append((variant, tile, s, float("nan"), str(e)[:60]))
                    continue
                if base is None:
                    base = out.float().clone()
                    err = 0.0
                else:
                    err = ((out.float() - base).norm() / base.norm()).item()
                rows.append((variant, tile, s, us, f"{flops / us / 1e6:7.1f} TF  relerr {err:.1e}"))
        best = min(r[3] for r in rows if r[3] == r[3])
        print(f"\n== {name}   (schedule: tile {auto_tile} splitK {auto_s})")
        for variant, tile, s, us, info in rows:
            mark = " <== best" if us == best else ""
            sched = " [sched]" if (tile, s, variant) == (auto_tile, auto_s, auto_v) else ""
            print(f"   {VARIANTS[variant]:7s} tile{'128' if tile == 1 else ' 64'} S{s:<2d} {us:8.1f} us  {info}{sched}{mark}")


if __name__ == "__main__":
    main()
