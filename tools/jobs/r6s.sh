# round 6, third session: GroupNorm apply with one-trip blocks (norm.hip): unit tests, then same-box frame A/B against the 16 KB blocks
T=gpurun_out/r6s; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_midas.py tests/test_gpu_vae.py -m gpu -q -p no:cacheprovider -k "groupnorm or gn or midas or vae" > $T/pytest_gn.log 2>&1; tail -3 $T/pytest_gn.log
for rep in 1 2; do
for mode in 1 0; do
  L2D_GN_BLOCK16K=$mode timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_old${mode}_$rep.json
  python -c "
import json
d=json.load(open('$T/bench_old${mode}_$rep.json')); print('block16k=$mode rep $rep', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'gn', d['kernels'].get('gn_apply_kernel'))"
done; done
