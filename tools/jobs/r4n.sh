T=gpurun_out/r4n; mkdir -p $T
timeout 600 python tools/wsgemm_stress.py > $T/stress.log 2>&1; tail -30 $T/stress.log
for m in 0 1; do L2D_WSGEMM=$m timeout 600 python -m pytest tests/test_gpu_midas.py -q -x -k push_pop 2>&1 | tail -3; done
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/exp_probe.py > $T/exp_probe.txt 2>&1; cat $T/exp_probe.txt
