T=gpurun_out/r3sk; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_rowgemm.py -q -x -k "split_k" > $T/pytest_splitk.log 2>&1; tail -8 $T/pytest_splitk.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --breakdown 0 --whole-frame 0 --multi-stream 0 > $T/bench_$tag.json 2>> $T/err.log; python -c "
import json
d=json.load(open('$T/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['plan_launches'], d['config']['output_finite'])"; }
for r in 1 2; do
run off.$r L2D_ROWGEMM_SPLIT_SLICE=0
run s320.$r L2D_ROWGEMM_SPLIT_SLICE=320
run s320_512.$r L2D_ROWGEMM_SPLIT_SLICE=320 L2D_ROWGEMM_SPLIT_GEO=5,1,2
run s320_412.$r L2D_ROWGEMM_SPLIT_SLICE=320 L2D_ROWGEMM_SPLIT_GEO=4,1,2
run s320_521.$r L2D_ROWGEMM_SPLIT_SLICE=320 L2D_ROWGEMM_SPLIT_GEO=5,2,1
run s640.$r L2D_ROWGEMM_SPLIT_SLICE=640
run s320_m8192.$r L2D_ROWGEMM_SPLIT_SLICE=320 L2D_ROWGEMM_SPLIT_MAX_M=8192
done
grep -v amdgpu $T/err.log | tail -5
