T=gpurun_out/r3y; mkdir -p $T
for cfgs in "384 640 2 16" "320 320 2 16" "256 448 1 12"; do set -- $cfgs
  timeout 500 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 20 --warmup 5 --cpu-frames 0 --whole-frame 0 --multi-stream 0 > $T/bench_$1x$2_n$3_L$4.json 2>> $T/err2.log
  python -c "
import json
d=json.loads(open('$T/bench_$1x$2_n$3_L$4.json').read().strip().splitlines()[-1]); print('$1x$2 N$3 L$4', d['value'], d['ms_per_step'], d['config']['plan_launches'], d['config']['output_finite'], d.get('parity_vs_oracle_full_size'), d.get('tattn_variants_ms_per_frame'))"
done
grep -v amdgpu $T/err2.log | tail -5
