"""Time cconv.hip schedules against the kernel the plan uses today for the same 3x3 conv, per shape, in ONE process.

Every candidate runs as a chain of NCOPY launches with rotating weight copies (the rotation exceeds the 256 MB Infinity Cache, so
the weights arrive from HBM as they do in the frame), timed per launch with l2d_time_each (an event in front of every launch).
Baselines: wsgemm (levels 1 / 2 resnet convs), pconv (level 0), igemm (up-samplers), each with the schedule the plan picks.

    python tools/cconv_time.py [--shapes cfg2] [--reps 5] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib, ops  # noqa: E402

DEV = "cuda"

# (name, B, H, W (output), Cin, Cout, ups, baseline)
CFG2 = [
    ("L0 320->320", 2, 64, 64, 320, 320, 0, "pconv"),
    ("L0 640->320", 2, 64, 64, 640, 320, 0, "pconv"),
    ("L0 960->320", 2, 64, 64, 960, 320, 0, "pconv"),
    ("L1 640->640", 2, 32, 32, 640, 640, 0, "wsgemm"),
    ("L1 960->640", 2, 32, 32, 960, 640, 0, "wsgemm"),
    ("L1 1280->640", 2, 32, 32, 1280, 640, 0, "wsgemm"),
    ("L1 1920->640", 2, 32, 32, 1920, 640, 0, "wsgemm"),
    ("L1 320->640", 2, 32, 32, 320, 640, 0, "wsgemm"),
    ("L2 1280->1280", 2, 16, 16, 1280, 1280, 0, "wsgemm"),
    ("L2 1920->1280", 2, 16, 16, 1920, 1280, 0, "wsgemm"),
    ("L2 2560->1280", 2, 16, 16, 2560, 1280, 0, "wsgemm"),
    ("L2 640->1280", 2, 16, 16, 640, 1280, 0, "wsgemm"),
    ("UP1 640->640 @64", 2, 64, 64, 640, 640, 1, "igemm"),
    ("UP2 1280->1280 @32", 2, 32, 32, 1280, 1280, 1, "igemm"),
]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(torch.float16)


def time_chain(make_op, ncopy, reps):
    pl = _lib.OpList()
    for k in range(ncopy):
        op, keep = make_op(k)
        pl.append(op, *keep)
    pl.run()
    torch.cuda.synchronize()
    us = pl.time_each_us(reps=reps)
    us = sorted(us[1:])            # (the first launch has no predecessor)
    return us[len(us) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None)
    ap.add_argument("--scheds", default=None, help="semicolon list of CG,KG,NLD,S to try instead of the built-in sweep")
    a = ap.parse_args()
    print("device:", _lib.device_name())
    results = []
    cnt = torch.zeros(1 << 16, dtype=torch.int32, device=DEV)
    for name, B, H, W, Cin, Cout, ups, base in CFG2:
        if a.only and a.only not in name:
            continue
        Hs, Ws = H >> ups, W >> ups
        M = B * H * W
        gflop = 2.0 * M * Cout * 9 * Cin / 1e9
        wbytes = Cout * 9 * Cin * 2
        ncopy = max(6, min(24, int(400e6 // wbytes) + 1))
        x = rnd(B, Hs, Ws, Cin, seed=1)
        ws_ = [rnd(Cout, Cin, 3, 3, seed=10 + k, scale=(9 * Cin) ** -0.5) for k in range(2)]     # two distinct weight tensors, packed copies rotate
        bias = rnd(Cout, seed=3).float()
        res = rnd(M, Cout, seed=5)
        out = torch.empty(M, Cout, dtype=torch.float16, device=DEV)
        row = {"shape": name, "gflop": gflop, "cands": {}}
        # ---- baseline
        if base == "wsgemm":
            wpk = [ops.pack_wsgemm_conv3x3(ws_[k % 2]).clone() for k in range(ncopy)]
            sched = ops.wsgemm_schedule(M, 9 * Cin, Cout, 0, 0, 0, 9)
            n_ws, n_cnt = ops.wsgemm_sizes(M, Cout, sched[0], sched[1], sched[3])
            wsb = torch.empty(max(n_ws, 1), dtype=torch.float32, device=DEV)
            mk = lambda k: ops.wsgemm(x, wpk[k], out, M=M, Nout=Cout, C1=Cin, ldx1=Cin, ldo=Cout, bias=bias, res=res, ldr=Cout, taps=9, B=B,
                                      H=H, W=W, T=H * W, sched=sched, ws=wsb, cnt=cnt)
            t = time_chain(mk, ncopy, a.reps)
            row["base"] = {"kernel": "wsgemm", "sched": list(sched), "us": t}
            del wpk
        elif base == "pconv":
            wpk = [ops.pack_conv3x3(ws_[k % 2]).clone() for k in range(ncopy)]
            mk = lambda k: ops.pconv(x, wpk[k], out, B=B, H=H, W=W, C1=Cin, ldx1=Cin, CinP=Cin, Nout=Cout, ldo=Cout, patch=(8, 16), bias=bias,
                                     res=res, ldr=Cout)
            t = time_chain(mk, ncopy, a.reps)
            row["base"] = {"kernel": "pconv", "us": t}
            del wpk
        else:
            wpk = [ops.pack_conv3x3(ws_[k % 2]).clone() for k in range(ncopy)]
            tile, S, variant = ops.igemm_schedule(M, Cout, 9 * Cin, 1, 0, 9)
            n_ws, n_cnt = ops.splitk_sizes(M, Cout, S, 1, tile)
            wsb = torch.empty(max(n_ws, 1), dtype=torch.float32, device=DEV)
            kw = dict(cnt=cnt) if ops.splitk_fused(S) else {}
            mk = lambda k: ops.igemm(x, wpk[k], out, M=M, Nout=Cout, C1=Cin, ldx1=Cin, CinP=Cin, ldo=Cout, bias=bias, taps=9, B=B, Hin=Hs, Win=Ws,
                                     Hout=H, Wout=W, stride=1, ups=ups, splitk=S, tile=tile, variant=variant, ws=wsb, order=1, **kw)
            t = time_chain(mk, ncopy, a.reps)
            row["base"] = {"kernel": "igemm", "sched": [tile, S, variant], "us": t}
            del wpk
        print(f"{name:22s} {gflop:6.1f} GF  base {row['base']['kernel']:6s} {row['base'].get('sched', '')}: {row['base']['us']:7.1f} us "
              f"{gflop / row['base']['us'] * 1e3:6.0f} TF/s", flush=True)
        # ---- cconv candidates
        nch = Cin // 64
        npat = B * (H // 8) * (W // 16)
        if a.scheds:
            cands = [tuple(int(v) for v in s.split(",")) for s in a.scheds.split(";")]
        else:
            cands = []
            for cg, kg in ((2, 2), (1, 4)):
                if Cout % (64 * cg):
                    continue
                tiles = npat * Cout // (64 * cg)
                for S in (1, 2, 3, 4, 5, 6, 8):
                    if S > nch or (S > 1 and tiles * S > (330 if kg != 2 or cg != 1 else 520)) or (S == 1 and tiles < 100 and nch > 4):
                        continue
                    for nld in ((2, 4) if (cg, kg) != (1, 2) else (1, 2)):
                        cands.append((cg, kg, nld, S))
        packed = {}
        for sched in cands:
            cg, kg, nld, S = sched
            if Cout % (64 * cg) or S > nch:
                continue
            if kg not in packed:
                packed[kg] = [ops.pack_cconv(ws_[k % 2], kg).clone() for k in range(ncopy)]
            wpk = packed[kg]
            n_ws, n_cnt = ops.cconv_sizes(B, H, W, Cout, cg, S)
            wsb = torch.empty(max(n_ws, 1), dtype=torch.float32, device=DEV) if S > 1 else None
            mk = lambda k: ops.cconv(x, wpk[k], out, B=B, H=H, W=W, C1=Cin, ldx1=Cin, Nout=Cout, ldo=Cout, KG=kg, ups=ups, bias=bias,
                                     res=(res if base != "igemm" else None), ldr=Cout, sched=sched, ws=wsb, cnt=(cnt if S > 1 else None))
            t = time_chain(mk, ncopy, a.reps)
            row["cands"][",".join(map(str, sched))] = t
            print(f"      cconv {sched}: {t:7.1f} us {gflop / t * 1e3:6.0f} TF/s   ({npat * Cout // (64 * cg) * S} blocks)", flush=True)
        if row["cands"]:
            best = min(row["cands"], key=row["cands"].get)
            row["best"] = best
            print(f"   -> best {best}: {row['cands'][best]:.1f} us vs {row['base']['us']:.1f} us ({row['base']['us'] / row['cands'][best]:.2f}x)", flush=True)
        results.append(row)
        del packed
        torch.cuda.empty_cache()
    if a.json:
        with open(a.json, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
