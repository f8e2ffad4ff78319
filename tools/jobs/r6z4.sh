# round 6, third session, FINAL product-code commit: the whole -m gpu suite (no -x), smoke, evidence run (bench line, kernel trace, in-frame trace,
# HBM traffic counters, matrix-pipe counters), then the other BASELINE configurations and five shapes outside the list on the same box
TAG=round6_final4
T=gpurun_out/r6z4; mkdir -p $T
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider > $T/pytest_gpu.log 2>&1; tail -3 $T/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -2 $T/smoke.log
timeout 1500 bash tools/profile_round.sh $TAG > $T/profile_round.log 2>&1; tail -4 $T/profile_round.log
for c in "256 256 1 12" "512 512 2 16" "512 768 2 24" "512 512 4 16" "576 1024 2 40" "320 320 2 16" "384 384 2 16" "448 704 1 12" "640 640 2 16" "512 896 3 16" "768 768 2 24"; do set -- $c; tag=${1}x${2}_n${3}_L${4}
  timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 0 --multi-stream 0 2>> $T/bench.err | tail -1 > $T/bench_$tag.json
  python -c "
import json
d=json.load(open('$T/bench_$tag.json')); print('$tag', d['value'], 'frames/s', d['ms_per_step'], 'ms', d['config']['plan_launches'], 'launches', 'gemm frac', d['roofline_gemm_kernels']['frac'], 'kv', d['roofline_kv_cache_kernel']['frac'])"
done
