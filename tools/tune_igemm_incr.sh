#!/bin/bash
# Incremental in-frame schedule exploration: baseline = the CURRENT table, plus the forced configurations given as
# arguments ("tile,S,variant" ...).  tools/igemm_pick.py keeps the table's pick unless a forced one is >= 3 % faster.
#   tools/tune_igemm_incr.sh <outdir> 2,1,8 2,1,3 ...
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT" /tmp/tune
run() {
    local name=$1; shift
    rm -rf /tmp/tune/p; mkdir -p /tmp/tune/p
    env "$@" rocprofv3 --kernel-trace -d /tmp/tune/p -o t -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --breakdown 0 \
        --dump-plan "$OUT/plan_$name.csv" > "$OUT/bench_$name.json" 2> /tmp/tune/err_$name.log
    local db; db=$(find /tmp/tune/p -name "*.db" | head -1)
    python tools/frame_trace.py "$db" "$OUT/plan_$name.csv" "$OUT/trace_$name.csv" 2 | head -1 | sed "s/^/$name: /"
}
run base L2D_DUMMY=1
for cfg in "$@"; do
    run "${cfg//,/_}" L2D_IGEMM_FORCE=$cfg
done
python tools/igemm_pick.py "$OUT/igemm_tuned.json" "$OUT"/trace_base.csv $(ls "$OUT"/trace_[0-9]*.csv) | tee "$OUT/pick.txt"
