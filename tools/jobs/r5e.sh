# round 5: weight-streaming kernel offered to the levels with 1280 < M <= 4608 stream tokens (level 1 of the BASELINE configs), opt-in
# per measured shape (`large` list).  In-frame tuner for the five configs, then same-box A/B against the round-4 table / bound.
T=gpurun_out/r5e; mkdir -p $T
cfgs=("512 512 2 16" "512 768 2 24" "512 512 4 16" "576 1024 2 40" "256 256 1 12")
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  timeout 600 python tools/wsgemm_tune.py --height $1 --width $2 --denoise-steps $3 --window $4 --max-m 4608 --report $T/wsgemm_tune_$tag.txt --out live2diff_amd/wsgemm_tuned.json > $T/tune_$tag.log 2>&1; tail -1 $T/tune_$tag.log
done
cp live2diff_amd/wsgemm_tuned.json $T/wsgemm_tuned.json
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  for mode in new old; do
    if [ $mode = old ]; then E="L2D_WSGEMM_MAX_M=1280 L2D_WSGEMM_TABLE=$PWD/tools/jobs/data/wsgemm_tuned_round4.json"; else E="A=1"; fi
    env $E timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 > $T/bench_${tag}_$mode.json 2>> $T/bench.err
    python -c "
import json
d=json.loads(open('$T/bench_${tag}_$mode.json').read().strip().splitlines()[-1]); print('$tag $mode', d['value'], d['ms_per_step'])"
  done
done
