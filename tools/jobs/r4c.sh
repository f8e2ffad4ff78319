# round 4, third GPU call: loader with precomputed row offsets / tap masks; stamps (fixed tool) + probe again
T=gpurun_out/r4c; mkdir -p $T
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/wsgemm_stamps.py > $T/wsgemm_stamps.txt 2>&1; cat $T/wsgemm_stamps.txt
timeout 600 python -m pytest tests/test_gpu_wsgemm.py -q 2>&1 | tail -3
timeout 900 python tools/wsgemm_probe.py --quick --out $T/wsgemm_probe.json > $T/probe.log 2>&1; grep -A5 "^==" $T/probe.log | head -120
