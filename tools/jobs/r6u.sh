# round 6, third session: igemm fallback heuristic refit to the tuned table (ops._igemm_heuristic_r6) against the round-1 rule, same box:
# the four BASELINE configs that have no igemm table entries, two shapes outside the list, and cfg-2 with the table switched off
T=gpurun_out/r6u; mkdir -p $T
run() { # tag h w n L extra-env...
  tag=$1; shift; h=$1; w=$2; n=$3; L=$4; shift 4
  for heur in 1 6; do
    env "$@" L2D_IGEMM_HEUR=$heur timeout 400 python bench.py --height $h --width $w --denoise-steps $n --window $L --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_${tag}_heur$heur.json
    python -c "
import json
d=json.load(open('$T/bench_${tag}_heur$heur.json')); print('$tag heur=$heur', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches', 'igemm', d['kernels'].get('igemm_kernel',{}).get('ms_per_frame'))"
  done
}
run cfg1 256 256 1 12 X=0
run cfg3 512 768 2 24 X=0
run cfg4 512 512 4 16 X=0
run cfg5 576 1024 2 40 X=0
run 384x384 384 384 2 16 X=0
run 640x640 640 640 2 16 X=0
run cfg2_notable 512 512 2 16 L2D_IGEMM_NO_TABLE=1
run cfg2_table 512 512 2 16 X=0
timeout 900 python -m pytest tests/test_gpu_unet.py -m gpu -q -p no:cacheprovider -k "full_size or other_baseline or sd15_width" > $T/pytest_unet.log 2>&1; tail -3 $T/pytest_unet.log
