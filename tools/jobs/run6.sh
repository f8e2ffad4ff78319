mkdir -p gpurun_out/r2f
(timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "flash" > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log); tail -5 gpurun_out/r2f/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 40 --whole-frame 0 --per-op gpurun_out/r2f/per_op.csv > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2f/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['kernels']['flash_attn_kernel'])"
grep flash gpurun_out/r2f/per_op.csv | awk -F, '{k=$3; n[k]++; s[k]+=$4} END{for(k in n) printf "%s  n=%d avg_us=%.1f\n", k, n[k], s[k]/n[k]}'
