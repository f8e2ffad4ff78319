"""Is the KV-cache kernel's HBM efficiency an access-ORDER effect?  Analysis build only (make PROBES=1):
    L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/kv_pattern_probe.py
Streams a 168 MB K slab + 168 MB V slab (one top-level cache row pair at cfg-2) HBM -> LDS with tattn_ring's ring discipline
and no arithmetic, in three stage orders (misc.hip kv_pattern_kernel)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib                                                  # noqa: E402

lib = _lib.lib
lib.l2d_kv_pattern_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
blocks, gpb = 256, 8                   # 256 blocks x 8 groups x 80 KB = 168 MB per slab (x2 slabs), like N=2 rows of level 0
slab = blocks * gpb * 81920
buf = torch.randn((2 * slab) // 4 + 1024, device="cuda")          # > Infinity Cache; several such buffers so launches do not re-hit it
bufs = [torch.randn_like(buf) for _ in range(3)] + [buf]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
names = {0: "4 rows x 8 pixels per stage (2.5 KB pieces at 10 KB stride: the kernel today)", 1: "16 rows x 2 pixels per stage (20 KB contiguous)",
         2: "same bytes, front to back"}
for pattern in (0, 1, 2, 0, 1):
    best = 0.0
    for b in bufs:
        gb = ctypes.c_float(0)
        _lib.check(lib.l2d_kv_pattern_bench(b.data_ptr(), sink.data_ptr(), slab, gpb, pattern, blocks, 3, ctypes.c_void_p(_lib.current_stream_ptr()),
                                            ctypes.byref(gb)), "kv_pattern_bench")
        best = max(best, gb.value)
    print(f"pattern {pattern}: {best:7.1f} GB/s   {names[pattern]}")
