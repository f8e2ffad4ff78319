# round 4: does the packed-fp32 anomaly reproduce outside wsgemm?  (analysis build)
L2D_LIB=$PWD/live2diff_amd/libl2d_hip_probes.so timeout 240 python tools/pk_repro.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pk_repro.txt
