mkdir -p gpurun_out/r2e
(timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "flash" > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e/pytest.log); tail -5 gpurun_out/r2e/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 40 --whole-frame 0 --per-op gpurun_out/r2e/per_op.csv > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2e/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['kernels']['flash_attn_kernel'], d.get('hbm_copy_gbps_measured'), d.get('hbm_read_gbps_measured'))"
grep flash gpurun_out/r2e/per_op.csv | awk -F, '{k=$3; n[k]++; s[k]+=$4} END{for(k in n) printf "%s  n=%d avg_us=%.1f\n", k, n[k], s[k]/n[k]}'
export TMPDIR=/tmp
L2D_PROF_ONLY=flash timeout 300 bash tools/pmc_ops.sh gpurun_out/r2e/pmc_flash.txt > gpurun_out/r2e/pmc.log 2>&1; tail -2 gpurun_out/r2e/pmc.log
cat gpurun_out/r2e/pmc_flash.txt | head -60
