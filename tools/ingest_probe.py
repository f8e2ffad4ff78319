"""What can one CU ingest?  (analysis build: make -C live2diff_amd/csrc PROBES=1 LIB=../libl2d_hip_probes.so;
L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/ingest_probe.py)
GB/s per CU and chip-wide for streaming 1 KB wave-loads from HBM (private cold regions) and from L2 (one shared 2 MB region),
into VGPRs and by LDS-DMA, against the number of active CUs (grid), waves per CU and loads in flight per wave."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from live2diff_amd import _lib  # noqa: E402

lib = _lib.lib
lib.l2d_ingest_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
buf = torch.randn(1 << 29, device="cuda")            # 2 GB
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
g = ctypes.c_float(0)
s = ctypes.c_void_p(_lib.current_stream_ptr())
names = {0: "HBM -> VGPR", 1: "L2  -> VGPR", 2: "HBM -> LDS (DMA)", 3: "L2  -> LDS (DMA)"}
for mode in (0, 1, 2, 3):
    print(f"\n== {names[mode]}: GB/s per CU (chip TB/s)")
    for grid in (16, 64, 256, 512):
        for waves in (1, 4, 8):
            row = []
            for infl in (2, 4, 8, 16, 32):
                if mode >= 2 and waves * infl > 64:
                    row.append("      -      ")
                    continue
                per_wave = infl * 1024 * 64                      # bytes per wave over the run (64 batches)
                region = per_wave if mode in (0, 2) else (2 << 20)
                if mode in (0, 2) and region * grid * waves > buf.numel() * 4:
                    row.append("      -      ")
                    continue
                _lib.check(lib.l2d_ingest_bench(buf.data_ptr(), sink.data_ptr(), region, mode, grid, waves, infl, 64, s, ctypes.byref(g)), "ingest")
                cus = min(grid, 256)
                row.append(f"{g.value / cus:6.1f} ({g.value / 1000:4.2f})")
            print(f"   grid {grid:3d} waves {waves}: in flight/wave 2,4,8,16,32 KB: " + "  ".join(row))
