T=gpurun_out/r3last; mkdir -p $T
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -1 $T/smoke.log
timeout 600 python -m pytest tests/test_gpu_unet.py -q -x -k "cfg2_repeatable or golden" > $T/pytest.log 2>&1; tail -2 $T/pytest.log
