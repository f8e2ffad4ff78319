# round 4: in-frame wsgemm tuning (with the round-3 kernels as the per-shape baseline -> skip list) for the five BASELINE configs,
# then a same-box A/B of each
T=gpurun_out/r4w; mkdir -p $T
cfgs=("512 512 2 16" "512 768 2 24" "512 512 4 16" "576 1024 2 40" "256 256 1 12")
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  timeout 900 python tools/wsgemm_tune.py --height $1 --width $2 --denoise-steps $3 --window $4 --report $T/wsgemm_tune_$tag.txt --out live2diff_amd/wsgemm_tuned.json > $T/tune_$tag.log 2>&1; tail -3 $T/tune_$tag.log
done
cp live2diff_amd/wsgemm_tuned.json $T/wsgemm_tuned.json
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  for mode in 1 0; do
    L2D_WSGEMM=$mode timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 > $T/bench_${tag}_ws$mode.json 2>> $T/bench.err
    python -c "
import json
d=json.loads(open('$T/bench_${tag}_ws$mode.json').read().strip().splitlines()[-1]); print('$tag wsgemm=$mode', d['value'], d['ms_per_step'])"
  done
done
