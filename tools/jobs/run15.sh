# igemm prologue cleanup (host-computed tile counts / reciprocals, kernarg warm-up): tests, probe, same-box A/B vs previous lib
T=gpurun_out/r3e; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vae.py -q -x -k "igemm or vae or conv" > $T/pytest_k.log 2>&1; tail -3 $T/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "test_tiny_unet_rollout_n3_graph or cfg2_full_size or cfg2_repeatable" > $T/pytest_u.log 2>&1; tail -3 $T/pytest_u.log
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/igemm_probe.py > $T/probe.txt 2> $T/probe.err; python - <<'PY'
import json
for line in open('gpurun_out/r3e/probe.txt'):
    name, js = line.split(' {', 1); d = json.loads('{' + js)
    print(f"{name:12s} {d['us']:6.2f} us life {d['life_med']:6d}", d['phases_med'])
PY
for rep in 1 2; do for lib in prev cur; do
  if [ $lib = prev ]; then export L2D_LIB=live2diff_amd/libl2d_hip_prev.so; else unset L2D_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --whole-frame 0 > $T/bench_${lib}_$rep.json 2>> $T/bench.err
  python -c "
import json; d=json.loads(open('$T/bench_${lib}_$rep.json').read().strip().splitlines()[-1]); k=d['kernels']; print('$lib', d['value'], d['ms_per_step'], {n:round(v['ms_per_frame'],3) for n,v in k.items() if v['ms_per_frame']>0.3})"
done; done
