# round 6 (second session): ten consumer waves in wsgemm offered to every shape of the five BASELINE configs in the frame; then same-box A/B
T=gpurun_out/r6p; mkdir -p $T
cp live2diff_amd/wsgemm_tuned.json $T/wsgemm_tuned_before.json
cfgs=("512 512 2 16" "512 768 2 24" "512 512 4 16" "576 1024 2 40" "256 256 1 12")
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  timeout 400 python tools/wsgemm_tune10.py --height $1 --width $2 --denoise-steps $3 --window $4 --report $T/tune10_$tag.txt > $T/tune10_$tag.log 2>&1; tail -1 $T/tune10_$tag.log
done
cp live2diff_amd/wsgemm_tuned.json $T/wsgemm_tuned_after.json
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  for mode in new old new old; do
    if [ $mode = old ]; then E="L2D_WSGEMM_TABLE=$PWD/$T/wsgemm_tuned_before.json"; else E="A=1"; fi
    env $E timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_${tag}_$mode.json
    python -c "
import json
d=json.load(open('$T/bench_${tag}_$mode.json')); print('$tag $mode', d['value'], d['ms_per_step'])"
  done
done
