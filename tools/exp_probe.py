"""Cycles per wave64 v_exp_f32 on gfx950 (analysis build: make -C live2diff_amd/csrc PROBES=1 LIB=../libl2d_hip_probes.so;
L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/exp_probe.py).  Settles the VALU floor of the flash-attention softmax:
the d = 40 kernel spends 32 exponentials and 28 MFMAs (16x16x32) per wave and 64-key tile."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from live2diff_amd import _lib  # noqa: E402

lib = _lib.lib
lib.l2d_exp_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
s = ctypes.c_void_p(_lib.current_stream_ptr())
ITERS = 2000
names = {0: "16 v_exp_f32", 1: "16 v_add_f32", 2: "16 v_exp_f32 + 14 MFMA 16x16x32", 3: "14 MFMA 16x16x32", 4: "16 v_add_f32 + 14 MFMA 16x16x32"}
res = {}
for waves in (4, 8):
    for mode in (1, 0, 3, 4, 2):
        for _ in range(2):
            out.zero_()
            _lib.check(lib.l2d_exp_probe(out.data_ptr(), mode, waves, ITERS, s), "exp_probe")
            torch.cuda.synchronize()
        c = out[: 256 * waves].double().median().item() / ITERS
        res[(waves, mode)] = c
        print(f"{waves // 4} wave(s) per SIMD  {names[mode]:34s} {c:8.1f} cycles per iteration")
for waves in (4, 8):
    w = waves // 4
    e, a, m, am, em = res[(waves, 0)], res[(waves, 1)], res[(waves, 3)], res[(waves, 4)], res[(waves, 2)]
    print(f"\n{w} wave(s) per SIMD: v_exp_f32 {e / 16:.2f} cycles per wave-instruction alone (v_add_f32 {a / 16:.2f}: ratio {e / a:.2f}); "
          f"beside the MFMAs: +{(em - m) / 16:.2f} cycles per exp over the bare MFMA stream ({m:.0f} cycles), v_add +{(am - m) / 16:.2f}")
    if w == 2:
        print(f"   per SIMD (two waves share it): {e / 16 / 2:.2f} cycles per exp alone, {em / 2:.0f} cycles per (16 exp + 14 MFMA) pair-iteration per wave")
tile = 2 * res[(8, 2)]          # one 64-key tile of the d = 40 kernel = 32 exp + 28 MFMA per wave = two probe iterations
print(f"\nflash d = 40, T = 4096, B*H = 16: 64 key tiles x 64 query blocks of 128 rows / 256 CUs x 4 SIMDs with 2 waves each:")
print(f"   exp + MFMA issue alone: {tile:.0f} cycles per tile and wave pair-slot -> {64 * tile / 2.4e3 / 1.0:.1f} us per 64 tiles at 2.4 GHz")
