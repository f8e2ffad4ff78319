"""Per-level timing of the KV-cache attention variants with COLD caches (enough rotating copies to exceed the 256 MB Infinity
Cache): which kernel should the few-token levels use?   python tools/tattn_levels.py [variants ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from live2diff_amd import _lib, ops  # noqa: E402

dev = "cuda"
variants = [int(v) for v in sys.argv[1:]] or [13, 1, 2]
N, L = 2, 16
g = torch.Generator(device=dev).manual_seed(0)
pe_idx = torch.arange(L, device=dev).repeat(N, 1).contiguous()
upd = torch.full((N,), L - 1, dtype=torch.int64, device=dev)
bias = torch.zeros(N, L, dtype=torch.float16, device=dev)
for (C, T) in ((320, 4096), (640, 1024), (1280, 256), (1280, 64)):
    per = N * 2 * T * L * C * 2
    ncopy = max(2, int(600e6 // per) + 1)
    caches = [torch.randn(N, 2, T, L, C, device=dev, generator=g, dtype=torch.float16) for _ in range(ncopy)]
    qkv = torch.randn(N * T, 3 * C, device=dev, generator=g, dtype=torch.float16)
    pe = [torch.randn(L, C, device=dev, generator=g, dtype=torch.float16) for _ in range(3)]
    out = torch.empty(N * T, C, device=dev, dtype=torch.float16)
    row = []
    for v in variants:
        pl = _lib.OpList()
        for c in caches:
            pl.append(*ops.tattn_stream(qkv, c, pe[0], pe[1], pe[2], pe_idx, upd, bias, out, N=N, T=T, C=C, L=L, H=8, variant=v))
        pl.run()
        torch.cuda.synchronize()
        ms = min(pl.time_ms(3) for _ in range(3))
        us = ms * 1e3 / ncopy
        row.append(f"v{v}: {us:6.1f} us {per / us / 1e6:5.2f} TB/s")
    print(f"C{C} T{T} ({per / 1e6:.1f} MB, {ncopy} cold copies):  " + "   ".join(row))
