T=gpurun_out/r3k; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn" > $T/pytest_tattn.log 2>&1; tail -2 $T/pytest_tattn.log
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/tattn_probe.py > $T/tattn_lw_stage_phases.txt 2> $T/err.log; cat $T/tattn_lw_stage_phases.txt; tail -3 $T/err.log
