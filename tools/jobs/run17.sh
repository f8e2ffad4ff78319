# fused split-K: row skip + S cap; the other BASELINE configs with the reduction fused / capped / separate
T=gpurun_out/r3g; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "splitk or groupnorm_statistics or schedule_matches" > $T/pytest_k.log 2>&1; tail -3 $T/pytest_k.log
for cfgs in "256 256 1 12" "512 768 2 24" "512 512 4 16" "576 1024 2 40" "512 512 2 16"; do set -- $cfgs
 for mode in sep cap16 all; do
  case $mode in sep) export L2D_IGEMM_SPLITK_FUSED=0; unset L2D_IGEMM_SPLITK_FUSED_MAX;; cap16) unset L2D_IGEMM_SPLITK_FUSED; unset L2D_IGEMM_SPLITK_FUSED_MAX;; all) unset L2D_IGEMM_SPLITK_FUSED; export L2D_IGEMM_SPLITK_FUSED_MAX=64;; esac
  timeout 300 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 20 --warmup 5 --no-cpu-baseline --breakdown 0 --whole-frame 0 > $T/bench_$1x$2_n$3_$mode.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_$1x$2_n$3_$mode.json').read().strip().splitlines()[-1]); print('$1x$2 n$3 L$4 $mode', d['value'], d['ms_per_step'], d['config']['plan_launches'])"
 done
done
