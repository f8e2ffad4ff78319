# round 4, second GPU call: why is wsgemm's stage ~2000 cycles?  in-kernel stamps + the per-CU ingest probe (analysis build)
T=gpurun_out/r4b; mkdir -p $T
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/ingest_probe.py > $T/ingest_probe.txt 2>&1; cat $T/ingest_probe.txt
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/wsgemm_stamps.py > $T/wsgemm_stamps.txt 2>&1; cat $T/wsgemm_stamps.txt
timeout 600 python -m pytest tests/test_gpu_wsgemm.py -q 2>&1 | tail -3
