T=gpurun_out/r3i; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn" > $T/pytest_tattn.log 2>&1; tail -3 $T/pytest_tattn.log
for v in 0 2 0 2; do timeout 300 python tools/tattn_time.py --height 576 --width 1024 --window 40 --variant $v >> $T/tattn_l40_ab.txt 2>> $T/err.log; done
cat $T/tattn_l40_ab.txt; tail -3 $T/err.log
timeout 600 python bench.py --height 576 --width 1024 --window 40 --steps 30 --warmup 5 --cpu-frames 0 > $T/bench_cfg5.json 2> $T/bench_cfg5.err; cut -c1-600 $T/bench_cfg5.json
