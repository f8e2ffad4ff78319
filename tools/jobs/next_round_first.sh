# First GPU call of the next round (prepared at the end of round 4, not run): the two-tiles-per-wave form of wsgemm is in the library
# and in the tuner's candidate list but in no tuning table.  1. re-tune the five BASELINE configurations in the frame (NT = 2
# candidates included) and A/B each against the round-3 kernel set; 2. bit-stability of the full frame under concurrent load with the
# new table; 3. the wsgemm GPU tests.  Copy gpurun_out/r5a/wsgemm_tuned.json over live2diff_amd/wsgemm_tuned.json if the A/B says so.
T=gpurun_out/r5a; mkdir -p $T
cfgs=("512 512 2 16" "512 768 2 24" "512 512 4 16" "576 1024 2 40" "256 256 1 12")
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  timeout 900 python tools/wsgemm_tune.py --height $1 --width $2 --denoise-steps $3 --window $4 --report $T/wsgemm_tune_$tag.txt --out live2diff_amd/wsgemm_tuned.json > $T/tune_$tag.log 2>&1; tail -3 $T/tune_$tag.log
done
cp live2diff_amd/wsgemm_tuned.json $T/wsgemm_tuned.json
grep -c '2, 2, \|, 2, 2' $T/wsgemm_tuned.json
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  for mode in 1 0; do
    L2D_WSGEMM=$mode timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 > $T/bench_${tag}_ws$mode.json 2>> $T/bench.err
    python -c "
import json
d=json.loads(open('$T/bench_${tag}_ws$mode.json').read().strip().splitlines()[-1]); print('$tag wsgemm=$mode', d['value'], d['ms_per_step'])"
  done
done
REPS=300 timeout 300 python tools/frame_stress.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $T/frame_stress.txt
timeout 200 python -m pytest tests/test_gpu_wsgemm.py -x -q 2>&1 | tail -2
