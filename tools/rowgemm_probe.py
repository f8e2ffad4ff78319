"""Where does a row-GEMM block spend its life?  Needs the analysis build of the library (in-kernel s_memtime stamps):
    make -C live2diff_amd/csrc clean && make -C live2diff_amd/csrc PROBES=1 LIB=../libl2d_hip_probes.so && make -C live2diff_amd/csrc clean && make -C live2diff_amd/csrc
    L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/rowgemm_probe.py
Per shape: median / p90 shader cycles between the stamps of a block (entry, weight ring requested, GroupNorm tables, activation
tile normalised + barrier, k loop, epilogue staging + barrier, row stores), and the launch's HIP-event duration."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops                                             # noqa: E402

DEV = "cuda"
SHAPES = [  # (name, M, K, N, pro, epi, res, sched)
    ("CxC L0 +res", 8192, 320, 320, 0, 0, True, None),
    ("CxC L0 +res (5,1,2)", 8192, 320, 320, 0, 0, True, (5, 1, 2)),
    ("CxC L0 +res (5,2,1)", 8192, 320, 320, 0, 0, True, (5, 2, 1)),
    ("LN+CxC L0", 8192, 320, 320, 1, 0, False, None),
    ("LN+qkv L0", 8192, 320, 960, 1, 0, False, None),
    ("LN+GEGLU L0", 8192, 320, 2560, 1, 1, False, None),
    ("LN+GEGLU L0 (4,4,1)", 8192, 320, 2560, 1, 1, False, (4, 4, 1)),
    ("FF2 L0 +res", 8192, 1280, 320, 0, 0, True, None),
    ("CxC L1 +res", 2048, 640, 640, 0, 0, True, None),
    ("LN+qkv L1", 2048, 640, 1920, 1, 0, False, None),
    ("CxC L2 +res", 512, 1280, 1280, 0, 0, True, None),
    ("CxC L2 +res (4,1,1)", 512, 1280, 1280, 0, 0, True, (4, 1, 1)),
    ("CxC L3 +res", 128, 1280, 1280, 0, 0, True, None),
]
NAMES = ["w_ring_issue", "gn_tables", "x_tile+norm+barrier", "k_loop", "stage+barrier", "row_stores"]


def main():
    _lib.lib.l2d_rowgemm_set_probe.argtypes = [ctypes.c_void_p]
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, M, K, N, pro, epi, use_res, sched in SHAPES:
        x = torch.randn(M, K, device=DEV, generator=g).half()
        w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).half()
        b = torch.randn(N, device=DEV, generator=g)
        gm = torch.ones(K, device=DEV).half() if pro else None
        wp, bp = ops.pack_rowgemm(w, b, gm, (torch.zeros(K, device=DEV).half() if pro else None), geglu=(epi == 1))
        No = N // 2 if epi == 1 else N
        out = torch.zeros(M, No, device=DEV, dtype=torch.float16)
        res = torch.randn(M, No, device=DEV, generator=g).half() if use_res else None
        op = ops.rowgemm(x, wp, out, M=M, K=K, Nout=N, ldx=K, ldo=No, bias=bp, res=res, ldr=(No if use_res else 0), epi=epi, pro=pro,
                         T=M // 2, sched=sched)
        nw, nt, mt = op[0].i[12], op[0].i[13], op[0].i[14]
        nblk = ((M + 32 * mt - 1) // (32 * mt)) * (N // 32 // (nw * nt))
        probe = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
        pl = _lib.OpList()
        pl.append(*op)
        for _ in range(3):
            pl.run()
        torch.cuda.synchronize()
        ms = pl.time_ms(reps=20)
        # cold: something else streams through the caches between two launches
        trash = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=DEV)
        trash.zero_()
        _lib.lib.l2d_rowgemm_set_probe(ctypes.c_void_p(probe.data_ptr()))
        pl.run()
        torch.cuda.synchronize()
        _lib.lib.l2d_rowgemm_set_probe(None)
        p = probe.view(nblk, 8)[:, :7].cpu()
        p = p[p[:, 0] > 0]
        for c in range(1, 7):
            p[:, c] = torch.where(p[:, c] > 0, p[:, c], p[:, c - 1])
        d = (p[:, 1:] - p[:, :-1]).double()
        life = (p[:, 6] - p[:, 0]).double()
        med = d.median(0).values.tolist()
        p90 = d.quantile(0.9, 0).tolist()
        print(f"{name:24s} geom ({nw},{nt},{mt}) blocks {nblk:5d}  warm {1e3 * ms:7.2f} us/launch   block life median {life.median().item():8.0f} p90 {life.quantile(0.9).item():8.0f} cycles")
        print("    " + "  ".join(f"{n_} {m_:.0f}/{q_:.0f}" for n_, m_, q_ in zip(NAMES, med, p90)))


if __name__ == "__main__":
    main()
