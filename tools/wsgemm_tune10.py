"""Round 6: offer the ten-consumer-wave schedule (10, 1, 2, 1) of wsgemm to every shape of a configuration it divides, IN THE FRAME:
one pass with the product's plan (whatever kernel each layer has today), one with the schedule forced and the weight-streaming
packing offered to every level (L2D_WSGEMM_LARGE_ALL); a shape takes the new schedule where it is >= 3 % faster than what the layer
runs today.  Merges into wsgemm_tuned.json (shapes; `large` for > 1280 tokens; removed from `skip`).

    python tools/wsgemm_tune10.py [--height 512 --width 512 --denoise-steps 2 --window 16] [--out live2diff_amd/wsgemm_tuned.json]"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

CAND = (10, 1, 2, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--denoise-steps", type=int, default=2)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "live2diff_amd", "wsgemm_tuned.json"))
    ap.add_argument("--report", default="")
    args = ap.parse_args()
    from live2diff_amd import _lib, ops
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import device_random_state_dict
    dev = torch.device("cuda", 0)
    cfg = sd15_config(window_size=args.window, sink_size=(4 if args.window == 12 else 8))
    N, h, w = args.denoise_steps, args.height // 8, args.width // 8

    def keyed_times(forced):
        """per-launch in-frame times of one plan, keyed like the wsgemm table, for every kernel a wsgemm-able layer may run on"""
        if forced:
            os.environ["L2D_WSGEMM_FORCE"] = ",".join(str(v) for v in CAND)
            os.environ["L2D_WSGEMM_LARGE_ALL"] = "1"
        else:
            os.environ.pop("L2D_WSGEMM_FORCE", None)
            os.environ.pop("L2D_WSGEMM_LARGE_ALL", None)
        unet = HipStreamingUNet(device_random_state_dict(cfg, dev), cfg, h, w, N, device=dev)
        kv = unet.prepare_cache(N)
        for c in kv:
            c.normal_()
        st = unet._plan("stream", kv)
        st.cond_pl.run(); st.pl.run()
        torch.cuda.synchronize()
        st.pl.time_each_us(1)
        us = st.pl.time_each_us(args.reps)
        per = collections.defaultdict(list)
        for j in range(len(st.pl)):
            op = st.pl[j]
            i = op.i
            if op.kind == _lib.OP_WSGEMM:
                key = f"{i[0]},{i[13]},{i[0] * (i[1] + i[2])},{i[14]},{i[21] * 32},{i[19]},{i[20]}"
                per[(key, "wsgemm", (i[9], i[10], i[11], max(1, i[12])))].append(us[j])
            elif op.kind == _lib.OP_IGEMM and max(1, i[20]) == 1 and i[11] == 1 and i[12] == 0:
                per[(ops.wsgemm_key(i[0], i[13], i[0] * (i[1] + i[2]), i[14], 0, 1 if i[19] == 1 else 0, 0), "igemm", None)].append(us[j])
            elif op.kind == _lib.OP_ROWGEMM and i[7] != 2:
                per[(ops.wsgemm_key(1, i[0], i[1], i[2], i[15] * 32, i[6], i[7]), "rowgemm", None)].append(us[j])
        del unet, st
        return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}, sum(us)

    base, frame0 = keyed_times(False)
    forc, frame1 = keyed_times(True)
    today = {}
    for (key, kern, sched), (t, n) in base.items():
        if key not in today or n > today[key][2]:
            today[key] = (t, f"{kern}{'' if sched is None else sched}", n)
    lines = [f"{args.height}x{args.width} N{N} L{args.window}: frame (sum of in-frame launch times) {frame0 / 1e3:.3f} ms today, {frame1 / 1e3:.3f} ms with {CAND} forced everywhere"]
    d = json.load(open(args.out))
    shapes, skip, large = d["shapes"], set(d.get("skip", [])), set(d.get("large", []))
    gain = 0.0
    for (key, kern, sched), (t, n) in sorted(forc.items()):
        if kern != "wsgemm" or sched != CAND or key not in today:
            continue
        t0, what, n0 = today[key]
        take = t < 0.97 * t0
        lines.append(f"{key:36s} x{n0:3d}  today {what:28s} {t0:6.1f} us   {CAND} {t:6.1f} us {'-> taken' if take else ''}")
        if take:
            shapes[key] = list(CAND)
            skip.discard(key)
            if int(key.split(",")[1]) > ops.WS_SMALL_M:
                large.add(key)
            gain += (t0 - t) * n0
    lines.append(f"expected gain {gain / 1e3:.3f} ms per frame")
    d["shapes"], d["skip"], d["large"] = shapes, sorted(skip), sorted(large)
    with open(args.out, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print("\n".join(lines))
    if args.report:
        open(args.report, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
