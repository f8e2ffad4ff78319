# round 4: what does building EVERY kernel without packed fp32 VALU ops cost?  same-box A/B of the bench line
T=gpurun_out/r4q2; mkdir -p $T
for v in default nopk default nopk; do
  lib=""; [ $v = nopk ] && lib=$PWD/build_variants/nopk.so
  L2D_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --breakdown 0 > $T/bench_$v.json 2>> $T/err.log
  python -c "
import json
d=json.loads(open('$T/bench_$v.json').read().strip().splitlines()[-1]); w=d.get('whole_frame') or {}; print('$v', d['value'], d['ms_per_step'], w.get('frames_per_s'), (w.get('pipelined') or {}).get('frames_per_s'))" | tee -a $T/summary.txt
done
