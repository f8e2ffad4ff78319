export TMPDIR=/tmp; mkdir -p /tmp/prof gpurun_out/ft
for v in "$@"; do
rm -rf /tmp/prof/*; rocprofv3 --kernel-trace -d /tmp/prof -o ft -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --breakdown 0 --tattn-variant $v --dump-plan gpurun_out/ft/plan_$v.csv > gpurun_out/ft/bench_$v.json 2>/tmp/err.log
DB=$(find /tmp/prof -name "*.db" | head -1); python tools/frame_trace.py $DB gpurun_out/ft/plan_$v.csv gpurun_out/ft/trace_$v.csv 2 | head -4
grep tattn gpurun_out/ft/trace_$v.csv | awk -F, '{a[$3]+=$4; n[$3]++} END {for (k in a) printf "   v'$v' %s n=%d avg=%.1f us\n", k, n[k], a[k]/n[k]}' | sort
done
