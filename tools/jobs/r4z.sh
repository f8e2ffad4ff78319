# round 4, last checks after the knob retirement: smoke + kernel-level GPU tests, then the UNet tests that do not need the CPU oracle at full size
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pconv.py -x -q 2>&1 | tail -3
timeout 420 python -m pytest tests/test_gpu_unet.py tests/test_gpu_stream_step.py -x -q -k "not full_size and not sd15_width_other" --durations=6 2>&1 | tail -12
