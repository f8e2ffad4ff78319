"""Pick the (NW, NT, NL, S) schedule of every wsgemm shape of a configuration from IN-FRAME timings, in one process.

    python tools/wsgemm_tune.py [--height 512 --width 512 --denoise-steps 2 --window 16] [--out live2diff_amd/wsgemm_tuned.json]

For each candidate schedule the stream plan is rebuilt with that schedule forced on every wsgemm launch it fits
(L2D_WSGEMM_FORCE; the split-K workspaces depend on it) and replayed with `l2d_time_each` (an event in front of every launch: each
launch is timed with its real neighbours and cold weights); the per-shape winner is kept when it beats the default schedule by
>= 3 %, and the table is merged into the JSON that ops.wsgemm_schedule reads."""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("L2D_WSGEMM", "1")
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--denoise-steps", type=int, default=2)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "live2diff_amd", "wsgemm_tuned.json"))
    ap.add_argument("--report", default="")
    ap.add_argument("--max-m", type=int, default=0, help="offer the weight-streaming packing to levels of up to this many stream tokens "
                                                         "(L2D_WSGEMM_MAX_M; default: the product's)")
    args = ap.parse_args()
    if args.max_m:
        os.environ["L2D_WSGEMM_MAX_M"] = str(args.max_m)
    os.environ["L2D_WSGEMM_LARGE_ALL"] = "1"          # offer the packing to every shape of the levels it may serve; the lists below decide
    from live2diff_amd import _lib, ops
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import device_random_state_dict
    dev = torch.device("cuda", 0)
    cfg = sd15_config(window_size=args.window, sink_size=(4 if args.window == 12 else 8))
    N, h, w = args.denoise_steps, args.height // 8, args.width // 8
    unet = HipStreamingUNet(device_random_state_dict(cfg, dev), cfg, h, w, N, device=dev)
    kv = unet.prepare_cache(N)
    for c in kv:
        c.normal_()

    def key_of(op):
        i = op.i
        return f"{i[0]},{i[13]},{i[0] * (i[1] + i[2])},{i[14]},{i[21] * 32},{i[19]},{i[20]}"

    def measure(force):
        if force is None:
            os.environ.pop("L2D_WSGEMM_FORCE", None)
        else:
            os.environ["L2D_WSGEMM_FORCE"] = ",".join(str(v) for v in force)
        unet._plans.clear()
        unet._graph.clear()
        st = unet._plan("stream", kv)
        st.cond_pl.run()
        st.pl.run()
        torch.cuda.synchronize()
        st.pl.time_each_us(1)
        us = st.pl.time_each_us(args.reps)
        per = collections.defaultdict(list)
        for j in range(len(st.pl)):
            op = st.pl[j]
            if op.kind == _lib.OP_WSGEMM:
                per[(key_of(op), (op.i[9], op.i[10], op.i[11], max(1, op.i[12])))].append(us[j])
        return {k: sum(v) / len(v) for k, v in per.items()}, {k: len(v) for k, v in per.items()}, sum(us)

    ops._WS_TUNED.clear()                      # measure against the cost model, not against an older table
    ops._WS_SKIP.clear()
    ops._WS_LARGE.clear()
    # ---- the round-3 kernels on the same layers (a second instance packed without wsgemm): in-frame time per shape key
    os.environ["L2D_WSGEMM"] = "0"
    unet0 = HipStreamingUNet(device_random_state_dict(cfg, dev), cfg, h, w, N, device=dev)
    os.environ["L2D_WSGEMM"] = "1"
    st0 = unet0._plan("stream", kv)
    st0.cond_pl.run(); st0.pl.run()
    torch.cuda.synchronize()
    st0.pl.time_each_us(1)
    us0 = st0.pl.time_each_us(args.reps)
    old = collections.defaultdict(list)
    for j in range(len(st0.pl)):
        op = st0.pl[j]
        i = op.i
        if op.kind == _lib.OP_IGEMM and max(1, i[20]) == 1 and i[11] == 1 and i[12] == 0:
            old[ops.wsgemm_key(i[0], i[13], i[0] * (i[1] + i[2]), i[14], 0, 1 if i[19] == 1 else 0, 0)].append(us0[j])
        elif op.kind == _lib.OP_ROWGEMM and i[7] != 2:
            old[ops.wsgemm_key(1, i[0], i[1], i[2], i[15] * 32, i[6], i[7])].append(us0[j])
        elif op.kind == _lib.OP_PCONV and i[19] == 0:          # the patch-resident 3x3 conv (levels 0 / 1 at the SD resolutions)
            old[ops.wsgemm_key(9, i[6] * i[7] * i[8], 9 * (i[1] + i[2]), i[14], 0, 0, 0)].append(us0[j])
    old = {k: sum(v) / len(v) for k, v in old.items()}
    del unet0, st0
    base, counts, frame0 = measure(None)
    best = {k[0]: (t, k[1]) for k, t in base.items()}
    default = dict(best)
    lines = [f"default schedules: frame (sum of in-frame launch times) {frame0 / 1e3:.3f} ms"]
    for nw, nt in ((1, 1), (2, 1), (4, 1), (5, 1), (8, 1), (2, 2), (4, 2)):
        for S in (1, 2, 3, 4, 6, 8, 12):
            for nl in (2,):                  # (one loader wave never won a shape in rounds of tuning; the kernel keeps the form)
                res, _c, fr = measure((nw, nt, nl, S))
                for (key, sched), t in res.items():
                    if sched != (nw, nt, nl, S):
                        continue                 # (the forced schedule did not fit this shape: it ran with its default)
                    if t < best[key][0]:
                        best[key] = (t, sched)
    table, skip, large = {}, [], []
    n_per_key = collections.Counter()
    for (key, _s), c in counts.items():
        n_per_key[key] += c
    gain = 0.0
    for key, (t, sched) in sorted(best.items()):
        t0, s0 = default[key]
        keep = t < 0.97 * t0
        t_old = old.get(key)
        worse = t_old is not None and t_old < 0.97 * t
        lines.append(f"{key:40s} x{n_per_key[key]:3d}  default {s0} {t0:6.1f} us   best {sched} {t:6.1f} us {'*' if keep else ' '}"
                     f"   round-3 kernel {t_old if t_old is not None else float('nan'):6.1f} us {'-> skip' if worse else ''}")
        big = int(key.split(",")[1]) > ops.WS_SMALL_M
        if big:
            # more tokens than the few-token levels: opt-in.  The weight-streaming form must be >= 3 % faster than the round-3 kernel
            if t_old is not None and t < 0.97 * t_old:
                large.append(key)
                table[key] = list(sched)
                lines[-1] += " -> large"
            continue
        if worse:
            skip.append(key)
        if keep:
            table[key] = list(sched)
            gain += (t0 - t) * n_per_key[key]
    lines.append(f"expected gain {gain / 1e3:.3f} ms per frame over the default schedules")
    os.environ.pop("L2D_WSGEMM_FORCE", None)
    prev, prev_skip, prev_large = {}, [], []
    if os.path.exists(args.out):
        d_ = json.load(open(args.out))
        prev, prev_skip, prev_large = d_.get("shapes", {}), d_.get("skip", []), d_.get("large", [])
    measured = set(best)
    prev = {k: v for k, v in prev.items() if k not in measured}       # (a shape measured now loses its older pick)
    prev.update(table)
    prev_skip = sorted((set(prev_skip) - measured) | set(skip))
    prev_large = sorted((set(prev_large) - measured) | set(large))
    with open(args.out, "w") as f:
        json.dump({"note": "in-frame picks of tools/wsgemm_tune.py: key = taps,M,Ktot,Nout,ntr,epi,pro -> [NW, NT, NL, S]; skip = shapes "
                           "where the round-3 kernel (igemm / rowgemm) measured >= 3 % faster in the frame: the packer keeps its form"
                           "; large = shapes with more than 1280 tokens where wsgemm measured >= 3 % faster than the round-3 kernel (opt-in)",
                   "shapes": prev, "skip": prev_skip, "large": prev_large}, f, indent=1, sort_keys=True)
    print("\n".join(lines))
    if args.report:
        open(args.report, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
