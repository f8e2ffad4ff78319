# round 5: block chains at the other BASELINE configs (level 0: 384 / 512 / 576 blocks of 32 tokens = 1.5 / 2 / 2.25 rounds of one block per CU)
T=gpurun_out/r5o; mkdir -p $T
cfgs=("512 768 2 24" "512 512 4 16" "576 1024 2 40")
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  for mode in 1 0; do
    L2D_ROWCHAIN=$mode timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 > $T/bench_${tag}_chain$mode.json 2>> $T/bench.err
    python -c "
import json
d=json.loads(open('$T/bench_${tag}_chain$mode.json').read().strip().splitlines()[-1]); print('$tag chain=$mode', d['value'], d['ms_per_step'], d['config']['plan_launches'], d['kernels'].get('rowchain_kernel'))"
  done
done
