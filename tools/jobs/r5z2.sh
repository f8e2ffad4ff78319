# round 5: the whole -m gpu suite + smoke again at the last commit that touches product code (cleanup of ops.py / parallel.py after r5z)
T=gpurun_out/r5z2; mkdir -p $T
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $T/pytest_gpu.log 2>&1; tail -3 $T/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -2 $T/smoke.log
timeout 400 python bench.py --steps 50 --warmup 10 > $T/bench_default.json 2> $T/bench.err; cut -c1-300 $T/bench_default.json
