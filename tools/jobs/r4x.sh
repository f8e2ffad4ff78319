# round 4: cross-launch weight prefetch experiment (side-stream walker paced by the plan's timeline): same-box A/B at cfg-2
T=gpurun_out/r4x2; mkdir -p $T
for pf in 0 "300,128,0" "300,256,0" "600,256,0" "300,512,0" "100,256,0" 0; do
  L2D_PREFETCH=$pf timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --breakdown 0 --whole-frame 0 > $T/bench_pf_${pf//,/_}.json 2>> $T/err.log
  python -c "
import json
d=json.loads(open('$T/bench_pf_${pf//,/_}.json').read().strip().splitlines()[-1]); print('prefetch=$pf', d['value'], d['ms_per_step'])" | tee -a $T/summary.txt
done
tail -5 $T/err.log
