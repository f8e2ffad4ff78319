# round 3, first GPU call: the new token-row GEMM -- kernel tests, RCCL world-1 test, SD-width rollouts, same-box A/B of the frame
# (L2D_ROWGEMM=0: every linear layer on igemm + separate norm launches = the round-2 plan), in-frame geometry tuning
T=gpurun_out/r3a; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_rowgemm.py -q -x > $T/pytest_rowgemm.log 2>&1; tail -15 $T/pytest_rowgemm.log
timeout 300 python -m pytest tests/test_gpu_rccl.py -q > $T/pytest_rccl.log 2>&1; tail -5 $T/pytest_rccl.log
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "golden or sd15_width_single_step or tiny_unet_rollout" > $T/pytest_unet.log 2>&1; tail -8 $T/pytest_unet.log
for mode in 1 0; do
  L2D_ROWGEMM=$mode timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 > $T/bench_rg$mode.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_rg$mode.json').read().strip().splitlines()[-1]); print('rowgemm=$mode', d['value'], d['ms_per_step'], d['config']['plan_launches']); print({k:(v['launches'],v['ms_per_frame']) for k,v in d.get('kernels',{}).items()})"
done
timeout 400 python tools/rowgemm_tune.py --report $T/rowgemm_tune.txt --out $T/rowgemm_tuned.json > $T/tune.log 2>&1; tail -45 $T/tune.log
