# tattn ring: branch-free DMA source selection -- parity + same-box A/B against the previous library (+ the 16-pixel geometry again)
T=gpurun_out/r3s; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn" > $T/pytest_k.log 2>&1; tail -3 $T/pytest_k.log
timeout 600 python -m pytest tests/test_gpu_unet.py -q -x -k "cfg2_full_size or cfg2_cache_update or cfg2_repeatable or test_tiny_unet_rollout_n3_graph" > $T/pytest_u.log 2>&1; tail -3 $T/pytest_u.log
for rep in 1 2; do for m in prev cur cur16; do
  unset L2D_LIB L2D_TATTN_RING
  if [ $m = prev ]; then export L2D_LIB=live2diff_amd/libl2d_hip_prev.so; fi
  if [ $m = cur16 ]; then export L2D_TATTN_RING=2; fi
  timeout 300 python bench.py --no-cpu-baseline --whole-frame 0 > $T/bench_${m}_$rep.json 2>> $T/bench.err
  python -c "
import json; d=json.loads(open('$T/bench_${m}_$rep.json').read().strip().splitlines()[-1]); k=d['kernels']; print('$m', d['value'], d['ms_per_step'], round(k['tattn_stream_kernel']['ms_per_frame'],3), d['roofline_kv_cache_kernel']['achieved'])"
done; done
