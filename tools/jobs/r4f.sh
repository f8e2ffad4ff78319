T=gpurun_out/r4f; mkdir -p $T
timeout 600 python tools/warm_vs_cold.py --out $T/warm_vs_cold.json > $T/warm_vs_cold.txt 2>&1; cat $T/warm_vs_cold.txt
