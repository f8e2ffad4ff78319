# rowbias kept out of the prologue wait + in-frame re-tune on the new prologue
T=gpurun_out/r3f; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vae.py tests/test_gpu_midas.py -q -x -k "igemm or vae or conv or midas" > $T/pytest_k.log 2>&1; tail -3 $T/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "test_tiny_unet_rollout_n3_graph or cfg2_full_size or cfg2_repeatable" > $T/pytest_u.log 2>&1; tail -3 $T/pytest_u.log
for rep in 1 2; do for lib in prev cur; do
  if [ $lib = prev ]; then export L2D_LIB=live2diff_amd/libl2d_hip_prev.so; else unset L2D_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --whole-frame 0 > $T/bench_${lib}_$rep.json 2>> $T/bench.err
  python -c "
import json; d=json.loads(open('$T/bench_${lib}_$rep.json').read().strip().splitlines()[-1]); k=d['kernels']; print('$lib', d['value'], d['ms_per_step'], {n:round(v['ms_per_frame'],3) for n,v in k.items() if v['ms_per_frame']>0.3})"
done; done
unset L2D_LIB
bash tools/tune_igemm_incr.sh $T/tune 2,1,1 2,1,9 2,1,7 2,1,8 2,2,1 2,3,1 2,4,1 2,6,1 1,1,5 1,1,4 1,1,10 1,2,5 1,3,5 > $T/tune.log 2>&1; tail -32 $T/tune.log
