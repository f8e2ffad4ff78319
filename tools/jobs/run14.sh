# fused split-K reduction (sc1 accesses, no fences): kernel tests, full-size parity + repeatability, same-box A/B
T=gpurun_out/r3b; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "splitk or groupnorm_statistics or schedule_matches" > $T/pytest_k.log 2>&1; tail -3 $T/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "test_tiny_unet_rollout_n3_graph or cfg2_full_size or cfg2_repeatable or sd15_width_single_step" > $T/pytest_u.log 2>&1; tail -3 $T/pytest_u.log
for rep in 1 2; do for f in 0 1; do
  L2D_IGEMM_SPLITK_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --whole-frame 0 > $T/bench_f${f}_$rep.json 2>> $T/bench.err
  python -c "
import json; d=json.loads(open('$T/bench_f${f}_$rep.json').read().strip().splitlines()[-1]); k=d['kernels']; print('fused=$f', d['value'], d['ms_per_step'], d['config']['plan_launches'], {n:round(v['ms_per_frame'],3) for n,v in k.items() if v['ms_per_frame']>0.3})"
done; done
