# round 6, third session: one-launch GroupNorm (gn_self_kernel): unit tests, depth-detector tests, full-size parity of the six configurations,
# bit-identity properties, then same-box A/B (L2D_GN_SELF=0/1) on the configurations whose plans it changes
T=gpurun_out/r6aa; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_midas.py -m gpu -q -p no:cacheprovider -k "groupnorm or midas" > $T/pytest_gn.log 2>&1; tail -3 $T/pytest_gn.log
timeout 1200 python -m pytest tests/test_gpu_unet.py -m gpu -q -p no:cacheprovider -k "full_size or other_baseline or sd15_width or rollout" > $T/pytest_unet.log 2>&1; tail -3 $T/pytest_unet.log
run() { tag=$1; shift; h=$1; w=$2; n=$3; L=$4; shift 4
  env "$@" timeout 400 python bench.py --height $h --width $w --denoise-steps $n --window $L --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_$tag.json
  python -c "
import json
d=json.load(open('$T/bench_$tag.json')); k=d['kernels']; print('$tag', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches', {n[:-7]: v['ms_per_frame'] for n, v in k.items() if n.startswith('gn')})"
}
for s in "256 256 1 12" "576 1024 2 40" "384 384 2 16" "640 640 2 16" "320 320 2 16"; do set -- $s
  for rep in 1 2; do
    run ${1}x${2}_n${3}_self0_$rep $1 $2 $3 $4 L2D_GN_SELF=0
    run ${1}x${2}_n${3}_self1_$rep $1 $2 $3 $4 X=0
  done
done
run cfg2_whole 512 512 2 16 X=0
timeout 300 python tools/midas_time.py > $T/midas_time.json 2>> $T/bench.err; tail -c 600 $T/midas_time.json
