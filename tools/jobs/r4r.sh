T=gpurun_out/r4r; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_wsgemm.py -q 2>&1 | tail -3
timeout 900 python tools/wsgemm_stress.py > $T/stress.log 2>&1; tail -22 $T/stress.log
timeout 600 python tools/race_hunt.py 2>&1 | tail -6
timeout 2400 python -m pytest tests -m gpu -q > $T/pytest_gpu.log 2>&1; tail -6 $T/pytest_gpu.log
