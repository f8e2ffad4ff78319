#!/usr/bin/env python
"""Measured HBM rates on the box: copy (read+write) and pure read at several in-flight depths / occupancies."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from live2diff_amd import _lib  # noqa: E402

lib = _lib.lib
lib.l2d_read_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
a = torch.randn(512 * 1024 * 1024 // 4 * 3, device="cuda")      # 1.5 GB > Infinity Cache
b = torch.empty_like(a)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
g = ctypes.c_float(0)
s = ctypes.c_void_p(_lib.current_stream_ptr())
_lib.check(lib.l2d_copy_bench(a.data_ptr(), b.data_ptr(), a.numel() * 4, 5, s, ctypes.byref(g)), "copy")
print(f"copy (read+write) {g.value:8.1f} GB/s")
for bpc in (2, 4, 8):
    for u in (1, 2, 4, 8, 16):
        _lib.check(lib.l2d_read_bench(a.data_ptr(), sink.data_ptr(), a.numel() * 4, u, bpc, 5, s, ctypes.byref(g)), "read")
        print(f"read  blocks/CU={bpc} loads-in-flight/thread={u:2d}  {g.value:8.1f} GB/s")
