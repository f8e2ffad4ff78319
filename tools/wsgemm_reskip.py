"""Round 6 (third session): re-decide, IN THE FRAME, which wsgemm shapes of a configuration should go back to the round-3 kernels.

The `skip` / `large` lists of wsgemm_tuned.json were measured against igemm launches scheduled by the round-1 fallback rule wherever
igemm_tuned.json had no entry (every configuration but cfg-2).  ops._igemm_heuristic_r6 made those launches faster, so some verdicts
flip.  One pass with the product's plan, one with L2D_WSGEMM=0 (the same layers on igemm / rowgemm / pconv); a shape that runs on wsgemm
today and is >= 3 % slower than the round-3 kernel goes to `skip` (few tokens) or leaves `large` (more than 1280 tokens).  Schedules of
the shapes that stay are not touched.

    python tools/wsgemm_reskip.py [--height 512 --width 768 --denoise-steps 2 --window 24] [--out live2diff_amd/wsgemm_tuned.json]"""
import argparse
import collections
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
    ap.add_argument("--reps", type=int, default=7)
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

    def keyed_times(ws_on):
        os.environ["L2D_WSGEMM"] = "1" if ws_on else "0"
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
                per[(key, "wsgemm")].append(us[j])
            elif op.kind == _lib.OP_IGEMM and max(1, i[20]) == 1 and i[11] == 1 and i[12] == 0:
                per[(ops.wsgemm_key(i[0], i[13], i[0] * (i[1] + i[2]), i[14], 0, 1 if i[19] == 1 else 0, 0), "igemm")].append(us[j])
            elif op.kind == _lib.OP_ROWGEMM and i[7] != 2:
                per[(ops.wsgemm_key(1, i[0], i[1], i[2], i[15] * 32, i[6], i[7]), "rowgemm")].append(us[j])
            elif op.kind == _lib.OP_PCONV and i[19] == 0:
                per[(ops.wsgemm_key(9, i[6] * i[7] * i[8], 9 * (i[1] + i[2]), i[14], 0, 0, 0), "pconv")].append(us[j])
        del unet, st
        os.environ["L2D_WSGEMM"] = "1"
        return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}, sum(us)

    today, frame1 = keyed_times(True)
    old, frame0 = keyed_times(False)
    old_by_key = {}
    for (key, kern), (t, n) in old.items():
        if key not in old_by_key or n > old_by_key[key][2]:
            old_by_key[key] = (t, kern, n)
    d = json.load(open(args.out))
    shapes, skip, large = d["shapes"], set(d.get("skip", [])), set(d.get("large", []))
    lines = [f"{args.height}x{args.width} N{N} L{args.window}: frame (sum of in-frame launch times) {frame1 / 1e3:.3f} ms today, "
             f"{frame0 / 1e3:.3f} ms with L2D_WSGEMM=0"]
    gain = 0.0
    for (key, kern), (t, n) in sorted(today.items()):
        if kern != "wsgemm" or key not in old_by_key:
            continue
        t0, k0, n0 = old_by_key[key]
        back = t0 < 0.97 * t
        lines.append(f"{key:36s} x{n:3d}  wsgemm {t:6.1f} us   {k0:8s} {t0:6.1f} us {'-> back to ' + k0 if back else ''}")
        if back:
            gain += (t - t0) * n
            if int(key.split(",")[1]) > ops.WS_SMALL_M:
                large.discard(key)
                shapes.pop(key, None)
            else:
                skip.add(key)
    lines.append(f"expected gain {gain / 1e3:.3f} ms per frame")
    d["shapes"], d["skip"], d["large"] = shapes, sorted(skip), sorted(large)
    with open(args.out, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print("\n".join(lines))
    if args.report:
        open(args.report, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
