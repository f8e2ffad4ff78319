# round 6, third session: wsgemm table off (rule + cost-model schedules) against the table at cfg-2 / cfg-3 / cfg-5: what the fallback still loses
T=gpurun_out/r6x; mkdir -p $T
run() { tag=$1; shift; h=$1; w=$2; n=$3; L=$4; shift 4
  env "$@" timeout 400 python bench.py --height $h --width $w --denoise-steps $n --window $L --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_$tag.json
  python -c "
import json
d=json.load(open('$T/bench_$tag.json')); k=d['kernels']; print('$tag', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches', {n[:-7]: v['ms_per_frame'] for n, v in k.items() if v['ms_per_frame'] > 0.25})"
}
run cfg2_table 512 512 2 16 X=0
run cfg2_notable_rule 512 512 2 16 L2D_WSGEMM_NO_TABLE=1
run cfg2_notable_norule 512 512 2 16 L2D_WSGEMM_NO_TABLE=1 L2D_WSGEMM_RULE=0
run cfg3_table 512 768 2 24 X=0
run cfg3_notable_rule 512 768 2 24 L2D_WSGEMM_NO_TABLE=1
run cfg5_table 576 1024 2 40 X=0
run cfg5_notable_rule 576 1024 2 40 L2D_WSGEMM_NO_TABLE=1
L2D_WSGEMM_NO_TABLE=1 timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --whole-frame 0 --multi-stream 0 --per-op $T/per_op_cfg2_notable.csv > /dev/null 2>> $T/bench.err
timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --whole-frame 0 --multi-stream 0 --per-op $T/per_op_cfg2_table.csv > /dev/null 2>> $T/bench.err
