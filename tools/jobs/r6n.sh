# round 6 (second session): flash pipelined loop as the default -- same-box A/B in the frame (L2D_FLASH_VARIANT 0 = auto = pipelined, 2 = plain ring)
T=gpurun_out/r6n; mkdir -p $T
for v in 0 2 0 2; do
  L2D_FLASH_VARIANT=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --multi-stream 0 --whole-frame 0 2>/dev/null | tail -1 > $T/bench_fv$v.json
  python - <<PY
import json; d=json.load(open("$T/bench_fv$v.json")); print("L2D_FLASH_VARIANT=$v", d["ms_per_step"], d["kernels"].get("flash_attn_kernel"))
PY
done
