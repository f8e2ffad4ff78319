# round 5: does the weight-streaming kernel (128-token tiles, 4 MFMAs per weight fragment, DMA'd activations) beat the round-3 row
# GEMM / igemm / pconv on the levels with MORE tokens (cfg-2 level 1: M = 2048, level 0: M = 8192)?  In-frame tuner with the
# packing offered to every level, per-shape skip list, then same-box A/B of the frame.
T=gpurun_out/r5d; mkdir -p $T
for mm in 2048 8192; do
  cp live2diff_amd/wsgemm_tuned.json $T/table_$mm.json
  timeout 600 python tools/wsgemm_tune.py --max-m $mm --out $T/table_$mm.json --report $T/tune_$mm.txt > $T/tune_$mm.log 2>&1; tail -2 $T/tune_$mm.log
done
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --whole-frame 0 --multi-stream 0 > $T/bench_$tag.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'], d['config']['plan_launches'], d.get('cpu_baseline', {}).get('parity'))"; }
run base A=1
run m2048 L2D_WSGEMM_MAX_M=2048 L2D_WSGEMM_TABLE=$PWD/$T/table_2048.json
run m8192 L2D_WSGEMM_MAX_M=8192 L2D_WSGEMM_TABLE=$PWD/$T/table_8192.json
run base2 A=1
