# round 6, third session: how much do the other fallback rules lose?  cfg-2 with each tuned table switched off (the proxy for shapes that have
# no entries), the chain kernel offered at 144 blocks (384x384), per-launch times of the 384x384 frame
T=gpurun_out/r6v; mkdir -p $T
run() { tag=$1; shift; h=$1; w=$2; n=$3; L=$4; shift 4
  env "$@" timeout 400 python bench.py --height $h --width $w --denoise-steps $n --window $L --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_$tag.json
  python -c "
import json
d=json.load(open('$T/bench_$tag.json')); k=d['kernels']; print('$tag', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches', {n[:-7]: v['ms_per_frame'] for n, v in k.items() if v['ms_per_frame'] > 0.2})"
}
run cfg2_tables 512 512 2 16 X=0
run cfg2_no_ws_table 512 512 2 16 L2D_WSGEMM_NO_TABLE=1
run cfg2_no_rg_table 512 512 2 16 L2D_ROWGEMM_NO_TABLE=1
run cfg2_no_tables 512 512 2 16 L2D_WSGEMM_NO_TABLE=1 L2D_ROWGEMM_NO_TABLE=1 L2D_IGEMM_NO_TABLE=1
run cfg2_tables_again 512 512 2 16 X=0
run 384_chain192 384 384 2 16 X=0
run 384_chain128 384 384 2 16 L2D_ROWCHAIN_MIN_BLOCKS=128
run 448x704_chain192 448 704 1 12 X=0
run 448x704_chain128 448 704 1 12 L2D_ROWCHAIN_MIN_BLOCKS=128
timeout 300 python bench.py --height 384 --width 384 --denoise-steps 2 --window 16 --steps 10 --warmup 4 --no-cpu-baseline --whole-frame 0 --multi-stream 0 --per-op $T/per_op_384.csv > /dev/null 2>> $T/bench.err
