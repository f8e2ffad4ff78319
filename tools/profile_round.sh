#!/bin/bash
# Round-end evidence run on the GPU box: bench line, kernel-trace stats, in-frame per-launch trace, HBM traffic counters.
#   tools/profile_round.sh <tag>     -> gpurun_out/<tag>/  (small CSV / JSON / txt only; the rocpd databases stay in /tmp)
# Copy what should be judged into profiles/ afterwards (profiles/README.md lists the files).
set -u
TAG=${1:-rX}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT" /tmp/prof
rm -rf /tmp/prof/*

# 1. the driver's default bench line (with the CPU baseline leg)
python bench.py > "$OUT/${TAG}_bench_cfg2.json" 2> "$OUT/bench_err.log"
cut -c1-400 "$OUT/${TAG}_bench_cfg2.json"

# 2. kernel trace of the same workload: per-kernel stats, per-shape table, in-frame duration of every launch
rocprofv3 --kernel-trace -d /tmp/prof/kt -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown 0 \
    --dump-plan "$OUT/plan.csv" > /dev/null 2> "$OUT/kt_err.log"
DB=$(find /tmp/prof/kt -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" "$OUT/${TAG}_bench_cfg2"
python tools/frame_trace.py "$DB" "$OUT/plan.csv" "$OUT/${TAG}_frame_trace.csv" 2 | tee "$OUT/${TAG}_frame_trace_summary.txt"

# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md), kernel-trace only, over the FRAME REPLAY target
# (tools/traffic_frame.py: the plan, three frames, nothing else -- round 4's passes over the whole bench.py died inside rocprofv3),
# each pass under its own timeout
for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && FRAMES=3 timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof/$c -o p -- python "$OLDPWD/tools/traffic_frame.py" \
        > "$OLDPWD/$OUT/pmc_${c}.log" 2>&1 ) || echo "PMC pass $c failed (see $OUT/pmc_${c}.log)"
done
python tools/pmc_summary.py "$OUT/${TAG}_pmc_frame.txt" $(find /tmp/prof/FETCH_SIZE /tmp/prof/WRITE_SIZE -name "*.db") --traffic "$OUT/traffic.json"
cat "$OUT/traffic.json" | head -40

# 4. matrix-pipe utilisation (north_star: "rocprof HBM GB/s and MFMA utilisation against peak"): SQ_VALU_MFMA_BUSY_CYCLES and
# GRBM_GUI_ACTIVE, one counter per pass, same frame replay target
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES; do
    ( cd /tmp && FRAMES=3 timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof/$c -o p -- python "$OLDPWD/tools/traffic_frame.py" \
        > "$OLDPWD/$OUT/pmc_${c}.log" 2>&1 ) || echo "PMC pass $c failed (see $OUT/pmc_${c}.log)"
done
python tools/pmc_summary.py "$OUT/${TAG}_pmc_mfma.txt" $(find /tmp/prof/SQ_VALU_MFMA_BUSY_CYCLES /tmp/prof/GRBM_GUI_ACTIVE /tmp/prof/SQ_WAVE_CYCLES /tmp/prof/SQ_BUSY_CU_CYCLES -name "*.db") --mfma "$OUT/mfma.json"
python -c "import json;d=json.load(open('$OUT/mfma.json'));print({k:v['mfma_util'] for k,v in d['families'].items()})"
