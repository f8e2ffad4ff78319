TAG=round3_last
mkdir -p gpurun_out/$TAG
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$TAG/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/${TAG}_pytest_gpu.log); tail -6 gpurun_out/$TAG/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/$TAG/${TAG}_smoke.log
( time timeout 900 python bench.py > gpurun_out/$TAG/${TAG}_bench_cfg2.json 2> gpurun_out/$TAG/bench_err.log ) 2> gpurun_out/$TAG/bench_time.txt; cat gpurun_out/$TAG/bench_time.txt | tail -4
python -c "
import json
d=json.load(open('gpurun_out/$TAG/${TAG}_bench_cfg2.json')); print(d['value'], d['ms_per_step'], d['whole_frame']['frames_per_s'], d['roofline'], d['roofline_kv_cache_kernel']['frac'], d['streams_per_gpu'], d['parity_vs_oracle_full_size']['rel_l2'])"
