T=gpurun_out/r3s; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "groupnorm or gn_" > $T/pytest_gn.log 2>&1; tail -3 $T/pytest_gn.log
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --whole-frame 0 > $T/bench.json 2>> $T/err.log
python -c "
import json
d=json.load(open('$T/bench.json')); print(d['value'], d['ms_per_step'], d['kernels']['gn_apply_kernel'], d['kernels_sum_ms'])"
tail -2 $T/err.log
