# round 6, third session: the whole -m gpu suite, smoke and the default bench line at the LAST commit (after the packed-cache layout key change)
T=gpurun_out/r6z5; mkdir -p $T
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider > $T/pytest_gpu.log 2>&1; tail -3 $T/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -2 $T/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $T/bench_driver_cmd.json 2> $T/bench.err; cut -c1-300 $T/bench_driver_cmd.json
