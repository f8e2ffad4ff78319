# round 6, third session: wsgemm_wanted rule for untuned token counts (L2D_WSGEMM_RULE=0: the old default) and the chain threshold, same box;
# then full-size parity of the new defaults at two shapes outside the BASELINE list
T=gpurun_out/r6w; mkdir -p $T
run() { tag=$1; shift; h=$1; w=$2; n=$3; L=$4; shift 4
  env "$@" timeout 400 python bench.py --height $h --width $w --denoise-steps $n --window $L --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_$tag.json
  python -c "
import json
d=json.load(open('$T/bench_$tag.json')); k=d['kernels']; print('$tag', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches', {n[:-7]: v['ms_per_frame'] for n, v in k.items() if v['ms_per_frame'] > 0.25})"
}
for s in "384 384 2 16" "640 640 2 16" "448 704 1 12" "512 896 3 16" "320 320 2 16"; do set -- $s
  run ${1}x${2}_n${3}_rule0 $1 $2 $3 $4 L2D_WSGEMM_RULE=0
  run ${1}x${2}_n${3}_rule1 $1 $2 $3 $4 X=0
done
run 320x320_n2_chain64 320 320 2 16 L2D_ROWCHAIN_MIN_BLOCKS=64
run 384x384_n2_chain192 384 384 2 16 L2D_ROWCHAIN_MIN_BLOCKS=192
for s in "384 384 2 16" "640 640 2 16"; do set -- $s
  timeout 600 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 20 --warmup 6 --cpu-frames 1 --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/parity_${1}x${2}.json
  python -c "
import json
d=json.load(open('$T/parity_${1}x${2}.json')); p=d['parity_vs_oracle_full_size']; print('parity ${1}x${2}', d['ms_per_step'], 'ms', p['rel_l2'], p['cosine'])"
done
