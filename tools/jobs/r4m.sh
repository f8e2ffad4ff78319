# round 4: the whole GPU suite with wsgemm on by default + smoke + the default bench line
T=gpurun_out/r4m; mkdir -p $T
timeout 2400 python -m pytest tests -m gpu -q -x > $T/pytest_gpu.log 2>&1; tail -6 $T/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -3 $T/smoke.log
timeout 900 python bench.py > $T/bench_cfg2.json 2> $T/bench.err; python -c "
import json
d=json.loads(open('$T/bench_cfg2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d.get('roofline_small_m'), d.get('roofline_gemm_kernels'), d.get('roofline_kv_cache_kernel'), d.get('whole_frame'), d.get('cpu_baseline'), d.get('parity_vs_oracle_full_size'))"
