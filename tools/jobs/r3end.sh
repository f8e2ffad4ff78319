T=gpurun_out/r3end; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_rowgemm.py tests/test_gpu_pconv.py tests/test_gpu_unet.py -q -x -k "not sd15_width_other and not other_baseline" > $T/pytest_subset.log 2>&1; tail -3 $T/pytest_subset.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -1 $T/smoke.log
