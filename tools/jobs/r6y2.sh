# round 6, third session: wsgemm shapes of cfg-1 / 3 / 4 / 5 re-decided against the round-3 kernels now that igemm's fallback rule is the refit one
# (tools/wsgemm_reskip.py), then same-box A/B of the four configs (and cfg-2 as the control: its plan must not change)
T=gpurun_out/r6y2; mkdir -p $T
cp live2diff_amd/wsgemm_tuned.json $T/wsgemm_tuned_before.json
cfgs=("512 768 2 24" "512 512 4 16" "576 1024 2 40" "256 256 1 12")
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  timeout 400 python tools/wsgemm_reskip.py --height $1 --width $2 --denoise-steps $3 --window $4 --report $T/reskip_$tag.txt > $T/reskip_$tag.log 2>&1; tail -1 $T/reskip_$tag.log
done
cp live2diff_amd/wsgemm_tuned.json $T/wsgemm_tuned_after.json
cfgs+=("512 512 2 16")
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  for mode in new old new old; do
    if [ $mode = old ]; then E="L2D_WSGEMM_TABLE=$PWD/$T/wsgemm_tuned_before.json"; else E="A=1"; fi
    env $E timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_${tag}_$mode.json
    python -c "
import json
d=json.load(open('$T/bench_${tag}_$mode.json')); print('$tag $mode', d['value'], d['ms_per_step'], d['config']['plan_launches'])"
  done
done
