# round 6, third session: hipGraph replay against direct launches on the launch-bound configurations (cfg-1, 320x320, 384x384) and cfg-2
T=gpurun_out/r6ac; mkdir -p $T
for c in "256 256 1 12" "320 320 2 16" "384 384 2 16" "512 512 2 16"; do set -- $c; tag=${1}x${2}_n${3}
  for g in 0 1 0 1; do
    timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --graph $g --steps 40 --warmup 10 --no-cpu-baseline --whole-frame 0 --multi-stream 0 --breakdown 0 2>> $T/bench.err | tail -1 > $T/bench_${tag}_graph$g.json
    python -c "
import json
d=json.load(open('$T/bench_${tag}_graph$g.json')); print('$tag graph=$g', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['latency_per_step'])"
  done
done
