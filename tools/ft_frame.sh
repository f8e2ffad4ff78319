export TMPDIR=/tmp; mkdir -p /tmp/prof gpurun_out/ft
rm -rf /tmp/prof/*; rocprofv3 --kernel-trace -d /tmp/prof -o ft -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --breakdown 0 --dump-plan gpurun_out/ft/plan_ln.csv > gpurun_out/ft/bench_ln.json 2>/tmp/err.log
DB=$(find /tmp/prof -name "*.db" | head -1); python tools/frame_trace.py $DB gpurun_out/ft/plan_ln.csv gpurun_out/ft/trace_ln.csv 2
