# tattn ring with 16-pixel groups (10 waves per block): parity + same-box A/B
T=gpurun_out/r3q; mkdir -p $T
L2D_TATTN_RING=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn" > $T/pytest_k.log 2>&1; tail -3 $T/pytest_k.log
L2D_TATTN_RING=2 timeout 600 python -m pytest tests/test_gpu_unet.py -q -x -k "cfg2_full_size or cfg2_cache_update or test_sd15_width_single_step" > $T/pytest_u.log 2>&1; tail -3 $T/pytest_u.log
for rep in 1 2; do for g in 0 2; do
  L2D_TATTN_RING=$g timeout 300 python bench.py --no-cpu-baseline --whole-frame 0 > $T/bench_g${g}_$rep.json 2>> $T/bench.err
  python -c "
import json; d=json.loads(open('$T/bench_g${g}_$rep.json').read().strip().splitlines()[-1]); k=d['kernels']; print('ring=$g', d['value'], d['ms_per_step'], round(k['tattn_stream_kernel']['ms_per_frame'],3), d['roofline_kv_cache_kernel']['achieved'])"
done; done
