"""ctypes binding of libl2d_hip.so (C ABI: include/l2d.h).

There is deliberately no fallback: if the shared library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C live2diff_amd/csrc`) importing this
module raises, and every op raises `L2DError` when the library reports a failure.
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede loading libl2d_hip.so: the extension has to bind to the HIP runtime torch
#                      already loaded (one runtime per process: shared streams and device pointers)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("L2D_LIB") or os.path.join(_HERE, "libl2d_hip.so")     # L2D_LIB: analysis builds (make PROBES=1), tools only

# op kinds (include/l2d.h)
OP_IGEMM, OP_GN_STATS, OP_GN_APPLY, OP_LAYERNORM, OP_FLASH_ATTN = 1, 2, 3, 4, 5
OP_TATTN_STREAM, OP_TATTN_WARMUP, OP_SKINNY_LINEAR, OP_TIMESTEP_EMBED = 6, 7, 8, 9
OP_NCHW_TO_NHWC, OP_NHWC_TO_NCHW, OP_LCM_STEP, OP_COPY = 10, 11, 12, 13
OP_RING_UPDATE, OP_STREAM_SHIFT, OP_RANDN = 14, 15, 16
OP_RESIZE_BILINEAR, OP_MINMAX, OP_DEPTH_NORM_RESIZE = 17, 18, 19
OP_STEM7X7, OP_RESAMPLE_NHWC, OP_EW, OP_ROWGEMM, OP_PCONV, OP_WSGEMM, OP_ROWCHAIN, OP_CCONV = 20, 21, 22, 23, 24, 25, 26, 27
ABI_VERSION = 6


class L2DError(RuntimeError):
    pass


class L2dOp(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("tag", ctypes.c_int32),
        ("p", ctypes.c_void_p * 16),
        ("i", ctypes.c_int32 * 32),
        ("l", ctypes.c_int64 * 4),
        ("f", ctypes.c_float * 4),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise L2DError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c \"import __graft_entry__ "
            "as g; g.build()\"` (hipcc --offload-arch=gfx950). There is no CPU/eager fallback by design.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.l2d_abi_version.restype = ctypes.c_int
    lib.l2d_last_error.restype = ctypes.c_char_p
    lib.l2d_device_check.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.l2d_set_dry_run.argtypes = [ctypes.c_int]
    lib.l2d_run_ops.argtypes = [ctypes.POINTER(L2dOp), ctypes.c_int, ctypes.c_void_p]
    lib.l2d_graph_create.argtypes = [ctypes.POINTER(L2dOp), ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.l2d_graph_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.l2d_graph_destroy.argtypes = [ctypes.c_void_p]
    lib.l2d_time_ops.argtypes = [ctypes.POINTER(L2dOp), ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_float)]
    lib.l2d_time_each.argtypes = [ctypes.POINTER(L2dOp), ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_float)]
    lib.l2d_copy_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.POINTER(ctypes.c_float)]
    lib.l2d_read_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    if lib.l2d_abi_version() != ABI_VERSION:
        raise L2DError(f"libl2d_hip.so ABI {lib.l2d_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise L2DError(f"{what}: rc={rc}: {lib.l2d_last_error().decode(errors='replace')}")


def device_name() -> str:
    buf = ctypes.create_string_buffer(96)
    check(lib.l2d_device_check(buf, 96), "l2d_device_check")
    return buf.value.decode()


_DRY_RUN = False


def set_dry_run(on: bool):
    """Validate-only mode (CPU test-suite): the C library checks every op's arguments and launches nothing; plans are
    built over CPU tensors and no HIP stream is ever touched."""
    global _DRY_RUN
    from . import ops
    check(lib.l2d_set_dry_run(1 if on else 0), "l2d_set_dry_run")
    _DRY_RUN = ops.DRY_RUN = bool(on)


def current_stream_ptr() -> int:
    """HIP stream handle of torch's current stream (the backend launches on the caller's stream, like the
    reference's PyTorch path; reference engine.py uses its own polygraphy stream + global syncs)."""
    if _DRY_RUN:
        return 0
    import torch

    return int(torch.cuda.current_stream().cuda_stream)


class OpList:
    """A contiguous array of l2d_op records (a *plan*)."""

    def __init__(self, ops=None):
        self._ops = list(ops or [])
        self._arr = None
        self._keep = []      # keep referenced tensors alive

    def append(self, op, *tensors):
        op.tag = len(self._ops)
        self._ops.append(op)
        self._keep.extend(t for t in tensors if t is not None)
        self._arr = None
        return op

    def extend(self, other: "OpList"):
        for op in other._ops:
            self.append(op)
        self._keep.extend(other._keep)

    def __len__(self):
        return len(self._ops)

    def __getitem__(self, i):
        return self._ops[i]

    def array(self):
        if self._arr is None:
            arr = (L2dOp * len(self._ops))()
            for j, op in enumerate(self._ops):
                ctypes.memmove(ctypes.byref(arr[j]), ctypes.byref(op), ctypes.sizeof(L2dOp))
            self._arr = arr
        return self._arr

    def run(self, stream=None):
        s = current_stream_ptr() if stream is None else stream
        check(lib.l2d_run_ops(self.array(), len(self._ops), ctypes.c_void_p(s)), "l2d_run_ops")

    def time_ms(self, reps=10, stream=None) -> float:
        s = current_stream_ptr() if stream is None else stream
        ms = ctypes.c_float(0)
        check(lib.l2d_time_ops(self.array(), len(self._ops), ctypes.c_void_p(s), reps, ctypes.byref(ms)), "l2d_time_ops")
        return float(ms.value)


    def time_each_us(self, reps=5, stream=None):
        """mean in-sequence duration of every op (microseconds, incl. the gap to its successor): l2d_time_each"""
        s = current_stream_ptr() if stream is None else stream
        out = (ctypes.c_float * len(self._ops))()
        check(lib.l2d_time_each(self.array(), len(self._ops), ctypes.c_void_p(s), reps, out), "l2d_time_each")
        return list(out)


class Graph:
    """hipGraph capture of an OpList (launch-overhead removal for the ~650-kernel frame)."""

    def __init__(self, ops: OpList, stream=None):
        s = current_stream_ptr() if stream is None else stream
        self._h = ctypes.c_void_p()
        self._ops = ops
        check(lib.l2d_graph_create(ops.array(), len(ops), ctypes.c_void_p(s), ctypes.byref(self._h)), "l2d_graph_create")

    def launch(self, stream=None):
        s = current_stream_ptr() if stream is None else stream
        check(lib.l2d_graph_launch(self._h, ctypes.c_void_p(s)), "l2d_graph_launch")

    def __del__(self):
        try:
            if self._h:
                lib.l2d_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass
