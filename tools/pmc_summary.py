#!/usr/bin/env python
"""Summarise rocprofv3 `--pmc` passes (rocpd SQLite databases) per kernel family.

  python tools/pmc_summary.py <out.txt> <db> [<db> ...] [--traffic profiles/traffic.json] [--mfma profiles/mfma.json]

Prints, per kernel name, the average of every collected counter per dispatch.  With --traffic, HBM bytes per
launch are derived from FETCH_SIZE / WRITE_SIZE (kB) exactly as MI355X_MICROARCH.md section HBM prescribes for
gfx950: FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x -> read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is used as reported (uncalibrated); the two counters come from separate passes.
With --mfma, matrix-pipe utilisation per kernel family and per kernel: SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs; = 32 x the number
of 32x32x16 MFMAs, 16 x the number of 16x16x32 ones) / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs) -- rocprofv3's own MfmaUtil expression
(reduce(SQ_VALU_MFMA_BUSY_CYCLES, sum) / (reduce(GRBM_GUI_ACTIVE, max) * SIMD_NUM)); the database holds GRBM_GUI_ACTIVE summed over
the 8 XCDs, so the per-XCD value is the sum / 8.  Separate passes, one counter each.
"""
import collections
import json
import sqlite3
import sys

PRODUCT = ("igemm", "rowgemm", "rowchain", "wsgemm", "pconv", "cconv", "gn_", "layernorm", "flash_attn", "flash_ring", "tattn", "skinny", "timestep", "nchw", "nhwc", "lcm_step")


def short(name):
    """kernel name with its template arguments kept (flash_ring_kernel<40, 2> and <80, 1> are different kernels)"""
    n = name.replace("void ", "").split("(")[0]
    return n


def family(name):
    if "rowchain_" in name:
        return "rowchain_kernel"
    for p in ("igemm_splitk_epilogue", "igemm_kernel", "rowgemm_kernel", "wsgemm_kernel", "pconv_kernel", "cconv_kernel", "gn_stats", "gn_apply", "layernorm", "flash_attn", "flash_ring", "tattn_stream", "tattn_warmup",
              "skinny_linear", "timestep_embed", "nchw_to_nhwc", "nhwc_to_nchw", "lcm_step"):
        if p in name:
            return p + ("_kernel" if not p.endswith("kernel") and p != "igemm_splitk_epilogue" else "")
    return None


def main():
    args = sys.argv[1:]
    traffic_path = mfma_path = None
    if "--mfma" in args:
        i = args.index("--mfma")
        mfma_path = args[i + 1]
        del args[i:i + 2]
    if "--traffic" in args:
        i = args.index("--traffic")
        traffic_path = args[i + 1]
        del args[i:i + 2]
    out_path, dbs = args[0], args[1:]
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    fam = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for db in dbs:
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, counter_name, value, dispatch_id, (end-start) from counters_collection").fetchall()
        seen = set()
        for k, cn, v, d, t in rows:
            ks = short(k)
            per_kernel[ks][cn].append(v)
            f = family(k)
            if f:
                fam[f][cn].append(v)
            if (db, d) not in seen:
                seen.add((db, d))
                dur[ks].append(t / 1e3)
    with open(out_path, "w") as f:
        for ks in sorted(per_kernel, key=lambda k: -sum(dur[k])):
            if not any(p in ks for p in PRODUCT):
                continue
            d = dur[ks]
            f.write(f"{ks}\n    dispatches(all passes)={len(d)} avg_us={sum(d) / len(d):.2f}\n")
            for cn, v in sorted(per_kernel[ks].items()):
                f.write(f"    {cn:34s} avg/dispatch = {sum(v) / len(v):.6g}\n")
    if traffic_path:
        tr = {}
        for k, ctr in fam.items():
            if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                rd = 2.0 * 1024.0 * sum(ctr["FETCH_SIZE"]) / len(ctr["FETCH_SIZE"])
                wr = 1024.0 * sum(ctr["WRITE_SIZE"]) / len(ctr["WRITE_SIZE"])
                tr[k] = {"hbm_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr),
                         "launches_sampled": len(ctr["FETCH_SIZE"]),
                         "note": "FETCH_SIZE*1024*2 (gfx950 16B/lane correction) + WRITE_SIZE*1024; separate --pmc passes"}
        with open(traffic_path, "w") as f:
            json.dump(tr, f, indent=1, sort_keys=True)
    if mfma_path:
        def util(ctr):
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in ctr or "GRBM_GUI_ACTIVE" not in ctr:
                return None
            busy = sum(ctr["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(ctr["SQ_VALU_MFMA_BUSY_CYCLES"])
            gui = sum(ctr["GRBM_GUI_ACTIVE"]) / len(ctr["GRBM_GUI_ACTIVE"]) / 8.0
            return {"mfma_busy_cycles_per_launch": round(busy), "gui_active_cycles_per_launch": round(gui), "mfma_util": round(busy / (gui * 1024.0), 4),
                    "launches_sampled": len(ctr["SQ_VALU_MFMA_BUSY_CYCLES"])}
        mj = {"families": {k: u for k, u in ((k, util(c)) for k, c in fam.items()) if u},
              "kernels": {k: u for k, u in ((k, util(c)) for k, c in per_kernel.items()) if u and u["mfma_busy_cycles_per_launch"] > 0},
              "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (sum over SIMDs) / (GRBM_GUI_ACTIVE per XCD * 1024 SIMDs): rocprofv3's MfmaUtil "
                      "expression; separate --pmc passes over tools/traffic_frame.py (in-frame launches, cold weights); kernels run ~1.1-1.4x "
                      "slower under counter collection"}
        with open(mfma_path, "w") as f:
            json.dump(mj, f, indent=1, sort_keys=True)
    print("wrote", out_path, traffic_path or "", mfma_path or "")


if __name__ == "__main__":
    main()
