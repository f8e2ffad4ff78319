# round 3, call 8: the whole -m gpu suite on the current build + smoke()
T=gpurun_out/r3h; mkdir -p $T
timeout 1500 python -m pytest tests -m gpu -q > $T/pytest_gpu.log 2>&1; tail -15 $T/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -3 $T/smoke.log
