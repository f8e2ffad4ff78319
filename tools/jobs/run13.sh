# fused split-K reduction: kernel tests, rollout tests, same-box A/B, two-stream probe
T=gpurun_out/r3a; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "splitk or groupnorm_statistics or schedule_matches" > $T/pytest_k.log 2>&1; tail -5 $T/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "tiny_unet_rollout or sd15_width_single_step or bit" > $T/pytest_u.log 2>&1; tail -5 $T/pytest_u.log
for rep in 1 2; do for f in 0 1; do
  L2D_IGEMM_SPLITK_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --whole-frame 0 > $T/bench_f${f}_$rep.json 2>> $T/bench.err
  python -c "
import json; d=json.loads(open('$T/bench_f${f}_$rep.json').read().strip().splitlines()[-1]); k=d['kernels']; print('fused=$f', d['value'], d['ms_per_step'], d['config']['plan_launches'], {n:round(v['ms_per_frame'],3) for n,v in k.items() if v['ms_per_frame']>0.3})"
done; done
PROBE_GRAPH=1 timeout 300 python tools/two_stream_probe.py > $T/probe_g1.json 2> $T/probe.err; cat $T/probe_g1.json
PROBE_GRAPH=0 timeout 300 python tools/two_stream_probe.py > $T/probe_g0.json 2>> $T/probe.err; cat $T/probe_g0.json; tail -3 $T/probe.err
