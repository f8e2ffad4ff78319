TAG=round3_final
mkdir -p gpurun_out/$TAG
(timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/$TAG/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/${TAG}_pytest_gpu.log); tail -14 gpurun_out/$TAG/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/${TAG}_smoke.log 2>&1; tail -3 gpurun_out/$TAG/${TAG}_smoke.log
timeout 900 bash tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1; tail -12 gpurun_out/$TAG/profile_round.log | cut -c1-300
export TMPDIR=/tmp
timeout 400 bash tools/pmc_ops.sh gpurun_out/$TAG/${TAG}_pmc_ops.txt > gpurun_out/$TAG/pmc_ops.log 2>&1; tail -2 gpurun_out/$TAG/pmc_ops.log
for cfgs in "256 256 1 12" "512 768 2 24" "512 512 4 16" "576 1024 2 40"; do set -- $cfgs
  timeout 500 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 5 --cpu-frames 0 > gpurun_out/$TAG/${TAG}_bench_$1x$2_n$3_L$4.json 2>> gpurun_out/$TAG/bench_other.err
  python -c "
import json
d=json.loads(open('gpurun_out/$TAG/${TAG}_bench_$1x$2_n$3_L$4.json').read().strip().splitlines()[-1]); print(d['config']['workload'][:60], d['value'], d['ms_per_step'], d['config']['kv_cache_GB_per_stream'], (d.get('whole_frame') or {}).get('frames_per_s'), (d.get('parity_vs_oracle_full_size') or {}).get('rel_l2'), (d.get('roofline_kv_cache_kernel') or {}).get('frac'))"
done
timeout 300 python tools/frame_each.py --csv gpurun_out/$TAG/${TAG}_frame_each_cfg2.csv > gpurun_out/$TAG/${TAG}_frame_each_cfg2.txt 2>> gpurun_out/$TAG/bench_other.err
timeout 200 python tools/midas_time.py 1 > gpurun_out/$TAG/${TAG}_midas_time.json 2>> gpurun_out/$TAG/bench_other.err; cat gpurun_out/$TAG/${TAG}_midas_time.json
