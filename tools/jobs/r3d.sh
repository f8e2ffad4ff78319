# round 3, call 4: patch-resident 3x3 conv (pconv) -- kernel tests, rollouts, same-box A/B of the frame
T=gpurun_out/r3d; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_pconv.py -q -x > $T/pytest_pconv.log 2>&1; tail -6 $T/pytest_pconv.log
timeout 600 python -m pytest tests/test_gpu_rowgemm.py -q -x > $T/pytest_rowgemm.log 2>&1; tail -3 $T/pytest_rowgemm.log
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "golden or sd15_width_single_step" > $T/pytest_unet.log 2>&1; tail -4 $T/pytest_unet.log
for mode in 1 0; do
  L2D_PCONV=$mode timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --per-op $T/per_op_pconv$mode.csv > $T/bench_pconv$mode.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_pconv$mode.json').read().strip().splitlines()[-1]); print('pconv=$mode', d['value'], d['ms_per_step'], d['config']['plan_launches']); print({k:(v['launches'],v['ms_per_frame'],v.get('tflops')) for k,v in d.get('kernels',{}).items()})"
done
grep -i pconv $T/per_op_pconv1.csv | head -30
tail -3 $T/bench.err
