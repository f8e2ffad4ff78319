# round 5: split-K slab sum of wsgemm with two slabs of loads in flight: same-box A/B of the frame (prev = library with the old loop)
T=gpurun_out/r5g; mkdir -p $T
for rep in 1 2; do for lib in new prev; do
  if [ $lib = prev ]; then export L2D_LIB=$PWD/live2diff_amd/libl2d_hip_prev.so; else unset L2D_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --whole-frame 0 --multi-stream 0 > $T/bench_${lib}_$rep.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_${lib}_$rep.json').read().strip().splitlines()[-1]); print('$lib $rep', d['value'], d['ms_per_step'], d['kernels'].get('wsgemm_kernel'))"
done; done
unset L2D_LIB
timeout 300 python -m pytest tests/test_gpu_wsgemm.py -q -p no:cacheprovider 2>&1 | tail -2
