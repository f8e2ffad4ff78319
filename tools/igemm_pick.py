#!/usr/bin/env python
"""Pick the per-shape igemm schedule from in-frame traces.

  for cfg in "2,1,1" "2,2,1" "1,3,5" ...; do
      L2D_IGEMM_FORCE=$cfg rocprofv3 --kernel-trace ... python bench.py --dump-plan plan_$cfg.csv ...
      python tools/frame_trace.py <db> plan_$cfg.csv trace_$cfg.csv
  done
  python tools/igemm_pick.py live2diff_amd/igemm_tuned.json trace_*.csv

Every trace row of an igemm launch carries the schedule it actually ran with (S, t, v in `dims`); for every shape
key (taps, M, Nout, Kp, epi, batch) the config with the lowest mean IN-FRAME duration over that shape's launches
wins.  The first trace on the command line is the baseline (default schedule) and is used for the summary.
"""
import collections
import csv
import json
import re
import sys


def parse(dims):
    d = dict(re.findall(r"([A-Za-z]+)(\d+)", dims))
    key = f'{d["taps"]},{d["M"]},{d["N"]},{d["Kp"]},{d["e"]},{d["b"]}'
    return key, (int(d["t"]), int(d["S"]), int(d["v"]))


def main():
    out_path, traces = sys.argv[1], sys.argv[2:]
    per = collections.defaultdict(lambda: collections.defaultdict(list))   # key -> cfg -> [us of every launch]
    base = {}
    for ti, path in enumerate(traces):
        for r in csv.DictReader(open(path)):
            if r["kernel"] != "igemm_kernel":
                continue
            key, cfg = parse(r["dims"])
            per[key][cfg].append(float(r["us_in_frame"]))
            if ti == 0:
                base.setdefault(key, [cfg, 0.0, 0])
                base[key][1] += float(r["us_in_frame"])
                base[key][2] += 1
    shapes, rows = {}, []
    tot_base = tot_best = 0.0
    for key, cfgs in per.items():
        n = base[key][2] if key in base else 1
        mean = {c: sum(v) / len(v) for c, v in cfgs.items()}
        best = min(mean, key=mean.get)
        b_us = base[key][1] / n if key in base else mean[best]
        # keep the baseline's pick unless the winner is clearly better (>= 3 % and >= 0.2 us): traces are noisy
        if key in base and not (mean[best] < 0.97 * b_us and b_us - mean[best] >= 0.2):
            best = base[key][0]
        shapes[key] = list(best)
        tot_base += b_us * n
        tot_best += mean[best] * n
        rows.append((n * (b_us - mean[best]), key, n, base.get(key, [None])[0], b_us, best, mean[best]))
    with open(out_path, "w") as f:
        json.dump({"note": "per-shape (tile, splitk, variant) picked from in-frame rocprofv3 traces on MI355X by tools/igemm_pick.py; "
                           "key = taps,M,Nout,Kp,epi,batch", "shapes": dict(sorted(shapes.items()))}, f, indent=0)
    for gain, key, n, bc, b_us, best, us in sorted(rows, reverse=True)[:40]:
        print(f"{key:32s} n={n:3d} base {bc} {b_us:7.2f} us -> {best} {us:7.2f} us   (-{gain:6.1f} us/frame)")
    print(f"igemm in-frame: baseline {tot_base / 1e3:.3f} ms -> picked {tot_best / 1e3:.3f} ms over {len(shapes)} shapes")


if __name__ == "__main__":
    main()
