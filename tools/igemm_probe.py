"""Where does an igemm block spend its life?  Needs the analysis build of the library (in-kernel s_memtime stamps):
    make -C live2diff_amd/csrc clean && make -C live2diff_amd/csrc PROBES=1 LIB=../libl2d_hip_probes.so && make -C live2diff_amd/csrc clean && make -C live2diff_amd/csrc
    L2D_LIB=live2diff_amd/libl2d_hip_probes.so python tools/igemm_probe.py
Per shape: median cycles between the 8 stamps of a block (entry, first weight DMA issued, descriptors + prologue issued, first
stage landed, K loop done, tile transposed to LDS, row stores issued, done), the spread of block entry times and the kernel's
HIP-event duration.  Stamps are the shader clock (s_memtime)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops                                             # noqa: E402

DEV = "cuda"
SHAPES = [  # (name, taps, M, N, K(Cin), tile, S, variant, epi, res)
    ("CxC level0", 1, 8192, 320, 320, 2, 1, 1, 0, True),
    ("CxC level0 v8 (1 block/CU)", 1, 8192, 320, 320, 2, 1, 8, 0, True),
    ("CxC level0 256 rows only", 1, 4096, 320, 320, 2, 1, 1, 0, True),
    ("CxC level1", 1, 2048, 640, 640, 2, 1, 1, 0, True),
    ("CxC level2", 1, 512, 1280, 1280, 2, 1, 7, 0, True),
    ("CxC level3", 1, 128, 1280, 1280, 2, 3, 1, 0, True),
    ("tiny conv", 9, 8192, 16, 64, 2, 1, 9, 2, False),
    ("GEGLU l0", 1, 8192, 2560, 320, 1, 1, 4, 1, False),
    ("GEGLU l1", 1, 2048, 5120, 640, 1, 1, 10, 1, False),
    ("conv l0", 9, 8192, 320, 320, 2, 1, 1, 0, True),
    ("ff2 l0", 1, 8192, 320, 1280, 2, 1, 5, 0, True),
]


def main():
    _lib.lib.l2d_igemm_set_probe.argtypes = [ctypes.c_void_p]
    g = torch.Generator(device=DEV).manual_seed(0)
    res_all = {}
    for name, taps, M, N, K, tile, S, variant, epi, use_res in SHAPES:
        cinp = ops.round_up(K, 64)
        x = torch.randn(M, K, device=DEV, generator=g).half()
        w = (torch.randn(N, taps * cinp, device=DEV, generator=g) * (taps * K) ** -0.5).half()
        b = torch.randn(N, device=DEV, generator=g)
        No = N // 2 if epi == 1 else N
        ldo = max(4, No)
        out = torch.zeros(M, ldo, device=DEV, dtype=torch.float16)
        res = torch.randn(M, No, device=DEV, generator=g).half() if use_res else None
        kw = dict(M=M, Nout=N, C1=K, ldx1=K, CinP=cinp, ldo=ldo, bias=b, epi=epi, res=res, ldr=(No if use_res else 0), taps=taps, tile=tile,
                  variant=variant, splitk=S)
        if taps == 9:
            kw.update(B=2, Hin=64, Win=64, Hout=64, Wout=64)
        if S > 1:
            n_ws, n_cnt = ops.splitk_sizes(M, N, S, 1, tile)
            kw.update(ws=torch.zeros(n_ws, device=DEV), cnt=torch.zeros(n_cnt, dtype=torch.int32, device=DEV))
        op = ops.igemm(x, w, out, **kw)
        t = 128 if tile == 1 else 64
        nblk = ((M + t - 1) // t) * ((N + t - 1) // t) * S
        probe = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
        pl = _lib.OpList()
        pl.append(*op)
        for _ in range(3):
            pl.run()
        torch.cuda.synchronize()
        ms = pl.time_ms(reps=20)
        _lib.lib.l2d_igemm_set_probe(ctypes.c_void_p(probe.data_ptr()))
        pl.run()
        torch.cuda.synchronize()
        _lib.lib.l2d_igemm_set_probe(None)
        p = probe.view(nblk, 8).cpu()
        for c in range(1, 8):                                   # a stamp that was not reached (e.g. "first stage landed" when the
            p[:, c] = torch.where(p[:, c] > 0, p[:, c], p[:, c - 1])   # whole K range fits the prologue) counts as zero time
        p = p[p[:, 0] > 0]
        if p.shape[0] == 0:
            print(name, "no stamps recorded")
            continue
        names = ["w_issue", "desc+prologue", "first_stage_wait", "k_loop", "to_lds", "stores", "gn+end"]
        # the counter is per XCD (not synchronised across XCDs): order blocks by entry time within their own clock domain
        # (domains show up as clusters > 1e9 apart) and report early / late blocks of the largest cluster separately
        order = p[:, 0].argsort()
        p = p[order]
        gaps = (p[1:, 0] - p[:-1, 0]) > 10_000_000
        cl = torch.cat([torch.zeros(1, dtype=torch.long), gaps.long().cumsum(0)])
        big = cl.bincount().argmax()
        q = p[cl == big]
        d = (q[:, 1:] - q[:, :-1]).float()
        k = max(1, q.shape[0] // 3)
        entry = (q[:, 0] - q[0, 0]).float()
        row = dict(us=round(ms * 1e3, 2), blocks=int(p.shape[0]), clusters=int(cl.max()) + 1, in_cluster=int(q.shape[0]),
                   phases_med=dict(zip(names, [int(v) for v in d.median(0).values])),
                   phases_first_third=dict(zip(names, [int(v) for v in d[:k].median(0).values])),
                   phases_last_third=dict(zip(names, [int(v) for v in d[-k:].median(0).values])),
                   entry_first_third=int(entry[:k].median()), entry_last_third=int(entry[-k:].median()),
                   life_med=int((q[:, 7] - q[:, 0]).float().median()), span=int(q[:, 7].max() - q[0, 0]))
        res_all[name] = row
        print(name, json.dumps(row))
    return res_all


if __name__ == "__main__":
    main()
