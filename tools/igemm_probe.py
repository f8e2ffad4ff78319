#!/usr/bin/env python
"""Isolated timing of a few igemm shapes over every (tile, pipeline variant): what the tile kernel itself can reach
on the short-K shapes that dominate the frame (warm L2; HIP events over 20 back-to-back launches)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from live2diff_amd import _lib, ops  # noqa: E402

DEV = "cuda"
SHAPES = [  # M, N, K, epi
    (8192, 2560, 320, 1), (2048, 5120, 640, 1), (512, 10240, 1280, 1), (8192, 960, 320, 0), (8192, 320, 320, 0),
    (2048, 640, 640, 0), (512, 1280, 1280, 0), (8192, 320, 1280, 0),
]


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float16)
    for M, N, K, epi in SHAPES:
        x = rn(M, K)
        if epi == 1:
            w, b = ops.pack_geglu(rn(N, K) * K ** -0.5, torch.zeros(N, device=DEV))
            out = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
            ldo = N // 2
        else:
            w, b = ops.pack_linear(rn(N, K) * K ** -0.5), torch.zeros(N, device=DEV)
            out = torch.empty(M, N, dtype=torch.float16, device=DEV)
            ldo = N
        res = []
        for tile in (1, 2):
            for v in range(10):
                if (v in (6, 7) and K % 128) or (tile == 1 and v in (7, 8, 9)):
                    continue
                op, keep = ops.igemm(x, w, out, M=M, Nout=N, C1=K, ldx1=K, CinP=w.shape[1], ldo=ldo, bias=b, epi=epi, tile=tile, variant=v)
                pl = _lib.OpList()
                pl.append(op, *keep)
                pl.time_ms(3)
                us = 1e3 * pl.time_ms(20)
                res.append((us, tile, v))
        res.sort()
        fl = 2.0 * M * N * K
        print(f"M{M} N{N} K{K} e{epi}: " + "  ".join(f"t{t}v{v} {us:.1f}us({fl / us / 1e6:.0f}TF)" for us, t, v in res[:6]), flush=True)


if __name__ == "__main__":
    main()
