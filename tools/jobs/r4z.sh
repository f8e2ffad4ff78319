# round 4, last check after the knob retirement: smoke + the kernel-level GPU tests (flash / igemm / norm / tattn live in test_gpu_kernels.py)
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pconv.py -x -q 2>&1 | tail -3
