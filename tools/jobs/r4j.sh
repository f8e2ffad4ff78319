# round 4: wsgemm in the plan -- UNet parity with it on, same-box A/B of the frame, in-frame schedule tuning, A/B again with the table
T=gpurun_out/r4j; mkdir -p $T
L2D_WSGEMM=1 timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "golden or sd15_width or tiny_unet_rollout or full_size_frame or repeat or graph" > $T/pytest_unet_ws1.log 2>&1; tail -5 $T/pytest_unet_ws1.log
for mode in 1 0; do
  L2D_WSGEMM=$mode timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 > $T/bench_ws$mode.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_ws$mode.json').read().strip().splitlines()[-1]); print('wsgemm=$mode', d['value'], d['ms_per_step'], d['config'].get('plan_launches')); print({k:(v['launches'],round(v['ms_per_frame'],3)) for k,v in d.get('kernels',{}).items()})"
done
timeout 600 python tools/wsgemm_tune.py --report $T/wsgemm_tune.txt --out $T/wsgemm_tuned.json > $T/tune.log 2>&1; tail -40 $T/tune.log
cp $T/wsgemm_tuned.json live2diff_amd/wsgemm_tuned.json
L2D_WSGEMM=1 timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 > $T/bench_ws1_tuned.json 2>> $T/bench.err
python -c "
import json
d=json.loads(open('$T/bench_ws1_tuned.json').read().strip().splitlines()[-1]); print('wsgemm=1 tuned', d['value'], d['ms_per_step']); print({k:(v['launches'],round(v['ms_per_frame'],3)) for k,v in d.get('kernels',{}).items()})"
tail -5 $T/bench.err
