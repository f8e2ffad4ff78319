# round 4: what would a perfect weight prefetcher buy?  the cold-weight probe with the weight copies sized to stay in the 256 MB
# Infinity Cache (but not in the 32 MB of L2): --cold-mb 100 vs 400
T=gpurun_out/r4l; mkdir -p $T
for mb in 400 100; do
  timeout 900 python tools/wsgemm_probe.py --quick --cold-mb $mb --out $T/probe_$mb.json > $T/probe_$mb.log 2>&1
  echo "=== cold-mb $mb"; grep -A2 "^==" $T/probe_$mb.log | grep -v "^--"
done
