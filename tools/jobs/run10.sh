mkdir -p gpurun_out/r2j
(timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -q -k "flash or igemm or groupnorm or tiny_unet_rollout or cfg2 or sd15_width_single" > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j/pytest.log); tail -8 gpurun_out/r2j/pytest.log
for cfgs in "2 1" "4 1" "2 0" "2 1"; do set -- $cfgs
  L2D_FLASH_GEO=$1 L2D_GN_FUSE=$2 timeout 300 python bench.py --no-cpu-baseline --steps 30 --whole-frame 0 --per-op gpurun_out/r2j/per_op_$1_$2.csv > gpurun_out/r2j/bench_$1_$2.json 2> gpurun_out/r2j/bench.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2j/bench_$1_$2.json').read().strip().splitlines()[-1]); k=d['kernels']; print('geo=$1 gnfuse=$2', d['value'], d['config']['plan_launches'], {n: k[n]['ms_per_frame'] for n in k if n in ('flash_attn_kernel','igemm_kernel','gn_apply_kernel','gn_stats_kernel')})"
  grep flash gpurun_out/r2j/per_op_$1_$2.csv | awk -F, '{k=$3; n[k]++; s[k]+=$4} END{for(k in n) printf "   %s  n=%d avg_us=%.1f\n", k, n[k], s[k]/n[k]}' | grep -E "Tk4096"
done
