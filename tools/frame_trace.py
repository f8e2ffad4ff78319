#!/usr/bin/env python
"""In-frame duration of every launch of the streaming plan.

  python bench.py --no-cpu-baseline --breakdown 0 --dump-plan plan.csv      (under rocprofv3 --kernel-trace)
  python tools/frame_trace.py <results.db> plan.csv out.csv [skip_frames]

Joins the rocprofv3 kernel trace (rocpd SQLite, dispatches in start order) with the plan dump: the frame loop
replays the same launch sequence, so every run of dispatches whose kernel names match the plan's sequence is one
frame.  Writes, per plan launch, the mean in-frame duration over the matched frames (after `skip_frames` warm-up
frames), the achieved TFLOP/s and GB/s from the plan's algorithmic work, and the gap to the next dispatch.
"""
import csv
import sqlite3
import sys

FAMILIES = ("igemm_splitk_epilogue", "igemm_kernel", "rowgemm_kernel", "rowchain", "wsgemm_kernel", "pconv_kernel", "cconv_kernel", "gn_stats_kernel", "gn_apply_kernel", "layernorm_kernel", "flash_attn_kernel",
            "tattn_stream", "tattn_warmup_kernel", "skinny_linear_kernel", "timestep_embed_kernel", "nchw_to_nhwc_kernel",
            "nhwc_to_nchw_kernel", "lcm_step_kernel")


def family(name):
    if "flash_ring_kernel" in name:            # the LDS-DMA ring build of the same plan op (flash_attn_ring.hip)
        return "flash_attn_kernel"
    for f in FAMILIES:
        if f in name:
            return "tattn_stream_kernel" if f == "tattn_stream" else ("rowchain_kernel" if f == "rowchain" else f)   # (rowchain_tail / _head kernels: one plan op kind)
    return None


def main():
    db, plan_path, out_path = sys.argv[1:4]
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    plan = list(csv.DictReader(open(plan_path)))
    expected, owner = [], []
    for r in plan:
        nd = int(r["dispatches"])
        if nd >= 1:
            expected.append(r["kernel"])
            owner.append(int(r["idx"]))
        if nd == 2:
            expected.append("igemm_splitk_epilogue")
            owner.append(int(r["idx"]))
    c = sqlite3.connect(db)
    disp = [(family(n), s, e) for n, s, e in c.execute("select name, start, end from kernels order by start")]
    disp = [d for d in disp if d[0]]
    names = [d[0] for d in disp]
    n = len(expected)
    frames, p = [], 0
    while p + n <= len(names):
        if names[p] == expected[0] and names[p:p + n] == expected:
            frames.append(p)
            p += n
        else:
            p += 1
    frames = frames[skip:]
    if not frames:
        sys.exit(f"no frame matched the plan ({n} dispatches/frame, {len(names)} product dispatches in the trace)")
    dur = [0.0] * len(plan)
    gap = [0.0] * len(plan)
    span = 0.0
    for p in frames:
        for k in range(n):
            _, s, e = disp[p + k]
            dur[owner[k]] += (e - s) / 1e3
            if k + 1 < n:
                gap[owner[k]] += max(0, disp[p + k + 1][1] - e) / 1e3
        span += (disp[p + n - 1][2] - disp[p][1]) / 1e3
    nf = len(frames)
    tot = {}
    with open(out_path, "w") as f:
        f.write("idx,kernel,dims,us_in_frame,gap_us,tflops,gbps\n")
        for j, r in enumerate(plan):
            us = dur[j] / nf
            fl, by = float(r["flops"]), float(r["bytes"])
            f.write(f'{j},{r["kernel"]},{r["dims"]},{us:.2f},{gap[j] / nf:.2f},{fl / us / 1e6 if us else 0:.1f},{by / us / 1e3 if us else 0:.0f}\n')
            t = tot.setdefault(r["kernel"], [0, 0.0, 0.0])
            t[0] += 1
            t[1] += us
            t[2] += gap[j] / nf
    print(f"{nf} frames matched, {n} dispatches/frame, frame span {span / nf / 1e3:.3f} ms, "
          f"busy {sum(dur) / nf / 1e3:.3f} ms, gaps {sum(gap) / nf / 1e3:.3f} ms")
    for k, t in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:28s} launches {t[0]:4d}  {t[1] / 1e3:7.3f} ms  gaps {t[2] / 1e3:6.3f} ms")


if __name__ == "__main__":
    main()
