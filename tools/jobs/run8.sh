mkdir -p gpurun_out/r2h
for x in 0 1 0 1; do
  L2D_FLASH_XCD=$x timeout 300 python bench.py --no-cpu-baseline --steps 40 --whole-frame 0 --per-op gpurun_out/r2h/per_op_$x.csv > gpurun_out/r2h/bench_$x.json 2> gpurun_out/r2h/bench.err
  python -c "
import json
d=json.loads(open('gpurun_out/r2h/bench_$x.json').read().strip().splitlines()[-1]); print('xcd=$x', d['value'], d['kernels']['flash_attn_kernel'])"
  grep flash gpurun_out/r2h/per_op_$x.csv | awk -F, '{k=$3; n[k]++; s[k]+=$4} END{for(k in n) printf "   %s  n=%d avg_us=%.1f\n", k, n[k], s[k]/n[k]}' | grep -E "Tk4096|Tk1024|Tk256"
done
