T=gpurun_out/r3side; mkdir -p $T
L2D_SIDE_SHORTCUT=1 timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "cfg2 or golden or test_tiny_unet_rollout" > $T/pytest_side.log 2>&1; tail -3 $T/pytest_side.log
for r in 1 2; do for v in 0 1; do
L2D_SIDE_SHORTCUT=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --breakdown 0 --whole-frame 0 --multi-stream 0 > $T/bench_side$v.$r.json 2>> $T/err.log
L2D_SIDE_SHORTCUT=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --breakdown 0 --whole-frame 0 --multi-stream 0 --graph 1 > $T/bench_side${v}_graph.$r.json 2>> $T/err.log
python -c "
import json
a=json.load(open('$T/bench_side$v.$r.json')); b=json.load(open('$T/bench_side${v}_graph.$r.json')); print('L2D_SIDE_SHORTCUT=$v direct', a['ms_per_step'], 'graph', b['ms_per_step'], a['config']['output_finite'])"
done; done
grep -v amdgpu $T/err.log | tail -3
