"""Stage phases of the KV-cache ring kernel (analysis build, make PROBES=1):
    L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/tattn_probe.py
Per wave, for one K stage and one V stage of the second pixel group: stage top -> own DMA share landed -> barrier passed ->
refill issued -> arithmetic issued."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops                                             # noqa: E402

DEV = "cuda"
_lib.lib.l2d_tattn_set_probe.argtypes = [ctypes.c_void_p]
g = torch.Generator(device=DEV).manual_seed(0)
for (C, T) in ((320, 4096), (640, 1024)):
    N, L, H = 2, 16, 8
    qkv = torch.randn(N * T, 3 * C, generator=g, device=DEV).half()
    cache = torch.randn(N, 2, T, L, C, generator=g, device=DEV).half()
    pe = [torch.randn(32, C, generator=g, device=DEV).half() for _ in range(3)]
    pe_idx = torch.arange(L, device=DEV).repeat(N, 1)
    upd = torch.tensor([9, 12], device=DEV)
    bias = torch.zeros(N, L, device=DEV).half()
    out = torch.empty(N * T, C, device=DEV).half()
    pl = _lib.OpList()
    pl.append(*ops.tattn_stream(qkv, cache, pe[0], pe[1], pe[2], pe_idx, upd, bias, out, N=N, T=T, C=C, L=L, H=H))
    for _ in range(3):
        pl.run()
    torch.cuda.synchronize()
    us = pl.time_ms(reps=20) * 1e3
    probe = torch.zeros(4096 * 16 * 2 * 8, dtype=torch.int64, device=DEV)
    _lib.lib.l2d_tattn_set_probe(ctypes.c_void_p(probe.data_ptr()))
    pl.run()
    torch.cuda.synchronize()
    _lib.lib.l2d_tattn_set_probe(None)
    p = probe.view(-1, 16, 2, 8).cpu()
    gb = N * 2 * T * L * C * 2 / (us * 1e-6) / 1e9
    print(f"C{C} T{T}: {us:.1f} us = {gb:.0f} GB/s of cache")
    loader_wave = 5 if os.environ.get("L2D_TATTN_RING", "") != "4" else None     # the loader-wave kernel: wave 5 only issues DMAs
    for who, sel in (("consumer waves", [w for w in range(5)]), ("loader wave", [5] if loader_wave else [])):
        if not sel:
            continue
        for kind, name in ((0, "K stage"), (1, "V stage")):
            q = p[:, sel, kind, :5].reshape(-1, 5)
            q = q[(q > 0).all(1)].double()
            if not q.shape[0]:
                continue
            d = q[:, 1:] - q[:, :-1]
            med, p90 = d.median(0).values, d.quantile(0.9, dim=0)
            names = ["wait own LDS reads", "barrier", "-", "arithmetic"] if (loader_wave and who[0] == "c") else \
                    ["wait DMA landed", "barrier", "issue refill", "-"] if who[0] == "l" else ["wait DMA", "barrier", "issue refill", "arithmetic"]
            print(f"   {who}, {name}: {q.shape[0]} waves; " + "  ".join(f"{n} {int(m)} (p90 {int(h)})" for n, m, h in zip(names, med, p90))
                  + f"  | stage total {int((q[:, 4] - q[:, 0]).median())}")
