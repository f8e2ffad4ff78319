"""Reproducer attempt for the round-4 packed-fp32 finding, analysis build only:
    make -C live2diff_amd/csrc PROBES=1 LIB=../libl2d_hip_probes.so
    L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/pk_repro.py
Runs the LayerNorm-fold arithmetic of wsgemm's epilogue in isolation (packed vs element-wise, compared in the kernel) idle and beside
the depth detector on a second stream, and prints how many packed results differed."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from live2diff_amd import _lib  # noqa: E402
from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict  # noqa: E402

DEV = "cuda"
lib = _lib.lib
lib.l2d_pk_repro.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
g = torch.Generator().manual_seed(3)
par = torch.randn(512, generator=g).to(DEV)
x = torch.randn(128, 1280, generator=g)
mean, var = x.mean(1), x.var(1, unbiased=False)
rstd = (var + 1e-5).rsqrt()
stat = torch.stack([rstd, -mean * rstd], 1).contiguous().to(DEV)
nbad = torch.zeros(4, dtype=torch.int32, device=DEV)
first = torch.zeros(32, dtype=torch.int32, device=DEV)
det = HipMidas(random_midas_state_dict(), device=DEV)
img = torch.rand(1, 3, 384, 384, device=DEV).half()
side = torch.cuda.Stream()
BLOCKS, ITERS, LAUNCHES = int(os.environ.get("BLOCKS", "320")), int(os.environ.get("ITERS", "200")), int(os.environ.get("LAUNCHES", "150"))
per_launch = BLOCKS * 4 * 32 * ITERS
MODES = {0: "packed ops right behind the LDS wait", 1: "s_nop 7 between the wait and the packed ops", 2: "plain VALU copy in between"}
for mode, busy in ((0, 0), (0, 1), (1, 1), (2, 1), (0, 1)):
    nbad.zero_(); first.zero_()
    tot_l = 0
    for rep in range(LAUNCHES):
        if busy and rep % 2 == 0:
            with torch.cuda.stream(side):
                det(img)
        _lib.check(lib.l2d_pk_repro(par.data_ptr(), stat.data_ptr(), nbad.data_ptr(), first.data_ptr(), BLOCKS | (mode << 16), ITERS,
                                    ctypes.c_void_p(_lib.current_stream_ptr())), "pk_repro")
        tot_l += 1
    torch.cuda.synchronize()
    nb, nw = int(nbad[0]), int(nbad[1])
    print(f"[{MODES[mode]}] {'beside the depth detector' if busy else 'idle GPU':26s}: {nb} differing packed results in {tot_l} launches "
          f"({tot_l * per_launch / 1e9:.2f} G packed multiply-fma-add sequences), {nw} waves affected", flush=True)
    for k in range(min(nw, 3)):
        w, blk = int(first[2 * k]) & 0xffffffff, int(first[2 * k + 1])
        print(f"     first in block {blk}: lane {w >> 16} mt {(w >> 12) & 15} g4 {(w >> 8) & 15} element {(w >> 4) & 15} wave {w & 15}")
