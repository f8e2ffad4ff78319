"""In-frame duration of every launch of a configuration's stream plan (l2d_time_each: one event in front of every launch, real
neighbours, cold weights), grouped by (kernel, dims):
    python tools/frame_each.py [--height 512 --width 512 --denoise-steps 2 --window 16] [--csv out.csv]
Prints per group: launches, total us per frame, mean us, TFLOP/s or GB/s of algorithmic work."""
import argparse
import collections
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
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--csv", default="")
    args = ap.parse_args()
    from live2diff_amd import _lib
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
    st.pl.time_each_us(2)
    us = st.pl.time_each_us(args.reps)
    rows = []
    for j in range(len(st.pl)):
        op = st.pl[j]
        fl, by = bench.op_work(op, _lib)
        rows.append((j, bench.KIND_NAMES.get(op.kind, str(op.kind)), bench.op_dims(op, _lib), fl, by, us[j]))
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("idx,kernel,dims,flops,bytes,us_in_frame\n")
            for r in rows:
                f.write(f"{r[0]},{r[1]},{r[2]},{r[3]:.0f},{r[4]:.0f},{r[5]:.2f}\n")
    grp = collections.OrderedDict()
    for r in rows:
        g = grp.setdefault((r[1], r[2]), [0, 0.0, 0.0, 0.0])
        g[0] += 1; g[1] += r[5]; g[2] += r[3]; g[3] += r[4]
    tot = sum(r[5] for r in rows)
    print(f"# {len(rows)} launches, {tot / 1e3:.3f} ms per frame (sum of in-frame launch times)")
    fam = collections.OrderedDict()
    for (k, d), g in grp.items():
        f = fam.setdefault(k, [0, 0.0, 0.0])
        f[0] += g[0]; f[1] += g[1]; f[2] += g[2]
    for k, f in sorted(fam.items(), key=lambda kv_: -kv_[1][1]):
        print(f"## {k}: {f[0]} launches, {f[1] / 1e3:.3f} ms" + (f", {f[2] / f[1] / 1e6:.0f} TFLOP/s" if f[2] > 1e9 else ""))
    for (k, d), g in sorted(grp.items(), key=lambda kv_: -kv_[1][1]):
        rate = f"{g[2] / g[1] / 1e6:7.0f} TF/s" if g[2] > 1e8 else f"{g[3] / g[1] / 1e3:7.0f} GB/s"
        print(f"{g[1]:8.1f} us  {g[0]:3d} x {g[1] / g[0]:7.2f}  {rate}  {k} {d}")


if __name__ == "__main__":
    main()
