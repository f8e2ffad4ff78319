# round 6, third session: chain kernel offered to cfg-1 (32 blocks) and to 256x384 / 288x288 shapes (48-81 blocks): where does it stop paying?
T=gpurun_out/r6ab; mkdir -p $T
run() { tag=$1; shift; h=$1; w=$2; n=$3; L=$4; shift 4
  env "$@" timeout 400 python bench.py --height $h --width $w --denoise-steps $n --window $L --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_$tag.json
  python -c "
import json
d=json.load(open('$T/bench_$tag.json')); print('$tag', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches')"
}
for rep in 1 2; do
run cfg1_chain96_$rep 256 256 1 12 X=0
run cfg1_chain32_$rep 256 256 1 12 L2D_ROWCHAIN_MIN_BLOCKS=32
run 256x384_n2_chain96_$rep 256 384 2 16 X=0
run 256x384_n2_chain48_$rep 256 384 2 16 L2D_ROWCHAIN_MIN_BLOCKS=48
run 288x288_n2_chain96_$rep 288 288 2 16 X=0
run 288x288_n2_chain64_$rep 288 288 2 16 L2D_ROWCHAIN_MIN_BLOCKS=64
done
