# round 3, call 3: row GEMM after the request-order / LayerNorm-in-registers / residual-prefetch rework
T=gpurun_out/r3c; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_rowgemm.py -q -x > $T/pytest_rowgemm.log 2>&1; tail -4 $T/pytest_rowgemm.log
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/rowgemm_probe.py > $T/rowgemm_block_phases.txt 2>&1; cat $T/rowgemm_block_phases.txt
for k in 640 2048; do
  L2D_ROWGEMM_PLAIN_MAX_K=$k timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 > $T/bench_plain$k.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_plain$k.json').read().strip().splitlines()[-1]); print('plain_max_k=$k', d['value'], d['ms_per_step'], d['config']['plan_launches']); print({k:(v['launches'],v['ms_per_frame']) for k,v in d.get('kernels',{}).items()})"
done
timeout 400 python tools/rowgemm_tune.py --report $T/rowgemm_tune.txt --out $T/rowgemm_tuned.json > $T/tune.log 2>&1; tail -32 $T/tune.log
cp $T/rowgemm_tuned.json live2diff_amd/rowgemm_tuned.json
timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 > $T/bench_tuned.json 2>> $T/bench.err
python -c "
import json
d=json.loads(open('$T/bench_tuned.json').read().strip().splitlines()[-1]); print('tuned', d['value'], d['ms_per_step'], d['config']['plan_launches']); print({k:(v['launches'],v['ms_per_frame']) for k,v in d.get('kernels',{}).items()})"
