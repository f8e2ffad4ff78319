#!/bin/bash
# In-frame igemm schedule exploration on the GPU box (see tools/igemm_pick.py).
#   tools/tune_igemm.sh <outdir> [bench args]
# Every config is one traced bench run (~10 frames); only the small CSVs leave the box.
set -u
OUT=${1:-gpurun_out/tune}; shift || true
export TMPDIR=/tmp
mkdir -p "$OUT" /tmp/tune
EXTRA=("$@")
run() {  # name, env assignments
    local name=$1; shift
    rm -rf /tmp/tune/p; mkdir -p /tmp/tune/p
    env "$@" rocprofv3 --kernel-trace -d /tmp/tune/p -o t -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --breakdown 0 \
        --dump-plan "$OUT/plan_$name.csv" "${EXTRA[@]}" > "$OUT/bench_$name.json" 2> /tmp/tune/err_$name.log
    local db; db=$(find /tmp/tune/p -name "*.db" | head -1)
    python tools/frame_trace.py "$db" "$OUT/plan_$name.csv" "$OUT/trace_$name.csv" 2 | head -1 | sed "s/^/$name: /"
}
run base L2D_IGEMM_NO_TABLE=1
for cfg in 2,1,1 2,1,9 2,1,5 2,1,2 2,1,6 2,2,1 2,2,9 2,3,1 2,3,9 2,4,1 2,4,9 2,6,1 2,8,1 2,12,1 \
           1,1,5 1,1,1 1,1,4 1,1,0 1,2,5 1,2,1 1,3,5 1,3,1 1,4,5 1,4,1 1,6,5 1,6,1 1,8,5 1,8,1 1,12,5 1,12,1 1,16,5 1,24,5 1,32,5; do
    run "${cfg//,/_}" L2D_IGEMM_NO_TABLE=1 L2D_IGEMM_FORCE=$cfg
done
python tools/igemm_pick.py "$OUT/igemm_tuned.json" "$OUT"/trace_base.csv $(ls "$OUT"/trace_[0-9]*.csv) | tee "$OUT/pick.txt"
