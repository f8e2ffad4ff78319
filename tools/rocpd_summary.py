#!/usr/bin/env python
"""Summarise a rocprofv3 `--kernel-trace` run (ROCm 7.2 writes a rocpd SQLite database) into the per-kernel
statistics table that `rocprofv3 --stats` would print, plus a per-launch-shape table for the product kernels.

  python tools/rocpd_summary.py gpurun_out/prof_r2/r2_results.db profiles/r1_bench_cfg2 [frames]

writes <prefix>_kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs, MinNs, MaxNs, Percentage) and
<prefix>_shapes.csv (kernel, blocks, grid_y, grid_z, launches_per_frame, total_us_per_frame, avg_us).
"""
import collections
import sqlite3
import sys

PRODUCT = ("igemm", "rowgemm", "rowchain", "wsgemm", "pconv", "cconv", "gn_", "layernorm", "flash_attn", "flash_ring", "tattn", "skinny", "timestep", "nchw", "nhwc", "lcm_step", "copy_kernel")


def main():
    db, prefix = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(prefix + "_kernel_stats.csv", "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"\n')
        for r in rows:
            f.write(f'"{r[0]}",{r[1]},{int(r[2])},{r[3]:.1f},{int(r[4])},{int(r[5])},{100.0 * r[2] / tot:.2f}\n')
    disp = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, (end-start)/1e3 from kernels order by start").fetchall()
    prod = [r for r in disp if any(k in r[0] for k in PRODUCT)]
    per_frame = collections.Counter(r[0] for r in prod)
    if not frames:
        # the frame loop repeats the same launch sequence: infer the frame count from the rarest product kernel
        frames = max(1, min(v for k, v in per_frame.items() if "nhwc_to_nchw" in k) if any("nhwc_to_nchw" in k for k in per_frame) else 1)
    agg = collections.OrderedDict()
    for r in prod:
        key = (r[0].replace("void ", "").split("(")[0], r[1] // max(1, r[4]), r[2], r[3])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += r[5]
    with open(prefix + "_shapes.csv", "w") as f:
        f.write("kernel,blocks,grid_y,grid_z,launches_per_frame,total_us_per_frame,avg_us\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f'"{k[0]}",{k[1]},{k[2]},{k[3]},{a[0] / frames:.2f},{a[1] / frames:.1f},{a[1] / a[0]:.2f}\n')
    ptot = sum(a[1] for a in agg.values())
    print(f"{len(rows)} kernels, {len(disp)} dispatches, product kernels {ptot / frames / 1e3:.3f} ms/frame over {frames} frames")


if __name__ == "__main__":
    main()
