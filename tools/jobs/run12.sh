mkdir -p gpurun_out/r2l
for g in 0 1; do
  (L2D_TATTN_RING=$g timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "tattn_stream" > gpurun_out/r2l/pytest_$g.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest_$g.log); tail -3 gpurun_out/r2l/pytest_$g.log
done
for g in 0 1 0 1; do
  L2D_TATTN_RING=$g timeout 300 python bench.py --no-cpu-baseline --steps 30 --whole-frame 0 > gpurun_out/r2l/bench_${g}_$RANDOM.json 2> gpurun_out/r2l/bench.err
done
for f in gpurun_out/r2l/bench_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['kernels']['tattn_stream_kernel'], d['kernels']['igemm_kernel']['ms_per_frame'])"; done
(timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "groupnorm_statistics" > gpurun_out/r2l/pytest_gn.log 2>&1; tail -3 gpurun_out/r2l/pytest_gn.log)
