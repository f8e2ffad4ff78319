TAG=round3_end
mkdir -p gpurun_out/$TAG
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$TAG/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/${TAG}_pytest_gpu.log); tail -6 gpurun_out/$TAG/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/$TAG/${TAG}_smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/${TAG}_bench_cfg2.json 2> gpurun_out/$TAG/bench_err.log ) 2> gpurun_out/$TAG/bench_time.txt; tail -3 gpurun_out/$TAG/bench_time.txt
python -c "
import json
d=json.load(open('gpurun_out/$TAG/${TAG}_bench_cfg2.json')); w=d['whole_frame']; print(d['value'], d['ms_per_step'], w['frames_per_s'], w['pipelined'], d['roofline']['frac'], d['roofline_kv_cache_kernel']['frac'], d['streams_per_gpu']['S4'], d['parity_vs_oracle_full_size']['rel_l2'], d['cpu_baseline']['value'])"
