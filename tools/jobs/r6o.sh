# round 6 (second session): ten-wave block chains -- rowchain tests at the new default, same-box frame A/B against the five-wave build, UNet tests
T=gpurun_out/r6o; mkdir -p $T
timeout 300 python -m pytest tests/test_gpu_rowchain.py -q -s -p no:cacheprovider > $T/pytest_rowchain.log 2>&1; tail -2 $T/pytest_rowchain.log
OLD=live2diff_amd/ablate/libl2d_rowchain_RC_TAIL_NT2.so
for v in new old new old; do
  if [ $v = old ]; then export L2D_LIB=$OLD; else unset L2D_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --multi-stream 0 --whole-frame 0 2>/dev/null | tail -1 > $T/bench_$v.json
  python - <<PY
import json; d=json.load(open("$T/bench_$v.json")); print("$v", d["ms_per_step"], d["kernels"].get("rowchain_kernel"))
PY
done
unset L2D_LIB
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_z_properties.py -q -x -p no:cacheprovider > $T/pytest_unet.log 2>&1; tail -3 $T/pytest_unet.log
