# round 6, third session: shapes OUTSIDE the BASELINE list on the final kernels (heuristic picks, no tuned-table entries): bench line + full-size
# parity against the fp32 oracle for each (weak-12 of the round-5 verdict: "any non-BASELINE resolution runs on heuristics nobody benchmarked")
T=gpurun_out/r6t; mkdir -p $T
shapes=("384 384 2 16" "640 640 2 16" "512 896 3 16" "768 768 2 24" "448 704 1 12")
for c in "${shapes[@]}"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  timeout 600 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --cpu-frames 1 --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_$tag.json
  python -c "
import json
d=json.load(open('$T/bench_$tag.json')); p=d.get('parity_vs_oracle_full_size',{})
print('$tag', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches', 'gemm frac', d['roofline_gemm_kernels']['frac'], 'kv', d['roofline_kv_cache_kernel']['frac'], 'parity rel_l2', p.get('rel_l2'), 'cos', p.get('cosine'))"
done
