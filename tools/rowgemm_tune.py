"""Pick the (NW, NT, MT) block geometry of every row-GEMM shape of a configuration from IN-FRAME timings, in one process.

    python tools/rowgemm_tune.py [--height 512 --width 512 --denoise-steps 2 --window 16] [--out live2diff_amd/rowgemm_tuned.json]

For each candidate geometry the whole stream plan is replayed with that geometry forced on every row-GEMM launch it fits
(`l2d_time_each`: an event in front of every launch, so each launch is timed with its real neighbours and cold weights), the
per-shape winner is kept when it beats the default schedule by >= 3 %, and the table is merged into the JSON that
ops.rowgemm_schedule reads.  Prints the per-kind in-frame breakdown of the default plan first."""
import argparse
import collections
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--denoise-steps", type=int, default=2)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "live2diff_amd", "rowgemm_tuned.json"))
    ap.add_argument("--report", default="")
    args = ap.parse_args()
    from live2diff_amd import _lib, ops
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import device_random_state_dict
    import bench
    dev = torch.device("cuda", 0)
    cfg = sd15_config(window_size=args.window, sink_size=(4 if args.window == 12 else 8))
    N, h, w = args.denoise_steps, args.height // 8, args.width // 8
    unet = HipStreamingUNet(device_random_state_dict(cfg, dev), cfg, h, w, N, device=dev)
    kv = unet.prepare_cache(N)
    for c in kv:
        c.normal_()
    st = unet._plan("stream", kv)
    st.cond_pl.run()
    st.pl.run()
    torch.cuda.synchronize()
    base_ops = [st.pl[j] for j in range(len(st.pl))]

    def timed(mod=None):
        pl = _lib.OpList()
        for op in base_ops:
            c = _lib.L2dOp()
            ctypes.memmove(ctypes.byref(c), ctypes.byref(op), ctypes.sizeof(_lib.L2dOp))
            if mod is not None and c.kind == _lib.OP_ROWGEMM:
                mod(c)
            pl.append(c)
        pl.time_each_us(1)
        return pl, pl.time_each_us(args.reps)

    pl0, t0 = timed()
    kinds = collections.defaultdict(lambda: [0, 0.0])
    for op, t in zip(pl0._ops, t0):
        k = bench.KIND_NAMES.get(op.kind, str(op.kind))
        kinds[k][0] += 1
        kinds[k][1] += t
    lines = [f"in-frame breakdown, default schedule ({len(pl0)} launches, {sum(t0) / 1e3:.3f} ms incl. event overhead):"]
    for k, (n, t) in sorted(kinds.items(), key=lambda kv_: -kv_[1][1]):
        lines.append(f"  {k:24s} {n:4d} launches {t / 1e3:8.3f} ms  avg {t / n:7.2f} us")

    def shape_key(op):
        return f"{op.i[0]},{op.i[1]},{op.i[2]},{op.i[15] * 32},{op.i[6]}"

    def fits(op, g):
        nw, nt, mt = g
        tiles, ntr, T = op.i[2] // 32, op.i[15], op.i[9]
        if tiles % (nw * nt) or ntr % (nw * nt) or nw > (5 if nt >= 3 else 8) or not ops._rowgemm_mt_ok(mt, nt, op.i[1], T) or \
                (mt >= 2 and op.i[0] < 1024 * mt):
            return False
        return ops._rowgemm_lds(op.i[1], nw, nt, mt, op.i[6], op.i[7], ntr * 32, bool(op.p[9])) <= 163840

    per_shape = collections.defaultdict(dict)          # key -> {geometry: us}
    for op, t in zip(pl0._ops, t0):
        if op.kind == _lib.OP_ROWGEMM:
            d = per_shape[shape_key(op)].setdefault(("default", op.i[12], op.i[13], op.i[14]), [])
            d.append(t)
    cands = [(nw, nt, mt) for mt in (1, 2, 4) for nt in (1, 2, 3, 4) for nw in (1, 2, 3, 4, 5, 6, 8) if not (mt >= 2 and nt > 2) and nw <= (5 if nt >= 3 else 8)]
    for g in cands:
        def mod(c, g=g):
            if fits(c, g):
                c.i[12], c.i[13], c.i[14] = g
        pl, t = timed(mod)
        for op, tt in zip(pl._ops, t):
            if op.kind == _lib.OP_ROWGEMM and (op.i[12], op.i[13], op.i[14]) == g:
                per_shape[shape_key(op)].setdefault(g, []).append(tt)
    table = {}
    lines.append("row-GEMM shapes (M,K,Nout,ntr,epi): default geometry / us  ->  best geometry / us  [launches]")
    tot_def = tot_best = 0.0
    for key, res in sorted(per_shape.items()):
        dk = next(k for k in res if k[0] == "default")
        n = len(res[dk])
        d_us = sum(res[dk]) / n
        best_g, best_us = dk[1:], d_us
        for g, ts in res.items():
            if g[0] != "default" and sum(ts) / len(ts) < best_us:
                best_g, best_us = g, sum(ts) / len(ts)
        keep = best_us < 0.97 * d_us
        tot_def += d_us * n
        tot_best += (best_us if keep else d_us) * n
        if keep:
            table[key] = list(best_g)
        lines.append(f"  {key:28s} {dk[1:]} {d_us:7.2f} -> {tuple(best_g)} {best_us:7.2f} {'*' if keep else ' '} [{n}]")
    lines.append(f"row GEMMs per frame: default {tot_def / 1e3:.3f} ms -> picked {tot_best / 1e3:.3f} ms")
    text = "\n".join(lines)
    print(text)
    if args.report:
        with open(args.report, "w") as f:
            f.write(text + "\n")
    old = {}
    if os.path.exists(args.out):
        with open(args.out) as f:
            old = json.load(f).get("shapes", {})
    old.update(table)
    with open(args.out, "w") as f:
        json.dump({"comment": "row-GEMM (NW, NT, MT) per shape 'M,K,Nout,ntr,epi', picked in-frame on MI355X by tools/rowgemm_tune.py",
                   "shapes": old}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
