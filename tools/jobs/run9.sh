mkdir -p gpurun_out/r2i
(timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "flash or igemm or groupnorm" > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2i/pytest.log); tail -5 gpurun_out/r2i/pytest.log
for g in 2 3 4 5 2 4; do
  L2D_FLASH_GEO=$g timeout 300 python bench.py --no-cpu-baseline --steps 30 --whole-frame 0 --per-op gpurun_out/r2i/per_op_$g.csv > gpurun_out/r2i/bench_$g.json 2> gpurun_out/r2i/bench.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2i/bench_$g.json').read().strip().splitlines()[-1]); print('geo=$g', d['value'], d['kernels']['flash_attn_kernel']['ms_per_frame'])"
  grep flash gpurun_out/r2i/per_op_$g.csv | awk -F, '{k=$3; n[k]++; s[k]+=$4} END{for(k in n) printf "   %s  n=%d avg_us=%.1f\n", k, n[k], s[k]/n[k]}' | grep -E "Tq4096"
done
