# round 4: HBM traffic counters (the PMC passes of tools/profile_round.sh crashed inside rocprofv3 on the first try): retry, and with
# the round-3 kernel set to see whether the new code object is what rocprofv3 trips over
export TMPDIR=/tmp; OUT=gpurun_out/r4y; mkdir -p $OUT /tmp/prof; rm -rf /tmp/prof/*
for ws in 1 0; do
for c in FETCH_SIZE WRITE_SIZE; do
    L2D_WSGEMM=$ws timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof/${c}_$ws -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --breakdown 0 --whole-frame 0 \
        > /dev/null 2> "$OUT/pmc_${c}_ws${ws}_err.log"; echo "ws=$ws $c rc=$?"
done
python tools/pmc_summary.py "$OUT/pmc_bench_ws$ws.txt" $(find /tmp/prof/FETCH_SIZE_$ws /tmp/prof/WRITE_SIZE_$ws -name "*.db" 2>/dev/null) --traffic "$OUT/traffic_ws$ws.json" 2>&1 | tail -2
cat "$OUT/traffic_ws$ws.json" | head -30
done
