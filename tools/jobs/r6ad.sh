# round 6, third session, EXPERIMENT: prefetch blocks riding on the GroupNorm launches (verdict item 3's cheap variant): blocks beyond the
# GroupNorm's own touch the weights of the conv behind it.  Same box, interleaved; GroupNorm tests first.
T=gpurun_out/r6ad; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "groupnorm" > $T/pytest_gn.log 2>&1; tail -2 $T/pytest_gn.log
for rep in 1 2; do
for P in 0 64 224 480; do
  L2D_GN_PREFETCH=$P timeout 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_pf${P}_$rep.json
  python -c "
import json
d=json.load(open('$T/bench_pf${P}_$rep.json')); k=d['kernels']; print('prefetch blocks $P rep $rep', d['value'], 'frames/s', d['ms_per_step'], 'ms', {n[:-7]: v['ms_per_frame'] for n, v in k.items() if n[:2] in ('ig','cc','pc','gn','ws')})"
done; done
L2D_GN_PREFETCH=224 timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --whole-frame 0 --multi-stream 0 --per-op $T/per_op_pf224.csv > /dev/null 2>> $T/bench.err
L2D_GN_PREFETCH=0 timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --whole-frame 0 --multi-stream 0 --per-op $T/per_op_pf0.csv > /dev/null 2>> $T/bench.err
