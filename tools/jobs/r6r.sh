# round 6, second session: the other BASELINE configurations on the final build, one box
T=gpurun_out/r6r; mkdir -p $T
cfgs=("256 256 1 12" "512 512 2 16" "512 768 2 24" "512 512 4 16" "576 1024 2 40")
for c in "${cfgs[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_$tag.json
  python -c "
import json
d=json.load(open('$T/bench_$tag.json')); print('$tag', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches', 'flash', d['kernels'].get('flash_attn_kernel',{}).get('ms_per_frame'), 'kv', d['roofline_kv_cache_kernel']['frac'])"
done
