# round 4: stamps after the loader rewrite; how much of the frame is cold-weight latency (upper bound for weight prefetch)
T=gpurun_out/r4e; mkdir -p $T
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/wsgemm_stamps.py > $T/wsgemm_stamps.txt 2>&1; cat $T/wsgemm_stamps.txt
timeout 600 python tools/warm_vs_cold.py --out $T/warm_vs_cold.json > $T/warm_vs_cold.txt 2>&1; cat $T/warm_vs_cold.txt
