# round 5: head segments of rowchain.hip: op-level parity + timing, frame A/B (heads on / off, chain off), parity of the frame
T=gpurun_out/r5n; mkdir -p $T
timeout 300 python -m pytest tests/test_gpu_rowchain.py -x -q -s -p no:cacheprovider > $T/pytest_rowchain.log 2>&1; grep -E "head segment|rowchain|passed|failed|Error|assert" $T/pytest_rowchain.log | head -20
for rep in 1 2; do for mode in "1 1" "1 0" "0 0"; do set -- $mode
  L2D_ROWCHAIN=$1 L2D_ROWCHAIN_HEADS=$2 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --whole-frame 0 --multi-stream 0 > $T/bench_c$1h$2_$rep.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_c$1h$2_$rep.json').read().strip().splitlines()[-1]); print('tail=$1 heads=$2 rep $rep', d['value'], d['ms_per_step'], d['config']['plan_launches'], d['kernels'].get('rowchain_kernel'))"
done; done
timeout 600 python -m pytest "tests/test_gpu_unet.py::test_full_size_frame_against_oracle" tests/test_gpu_z_properties.py -x -q -p no:cacheprovider 2>&1 | tail -3
