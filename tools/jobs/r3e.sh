# round 3, call 5: temporal-attention ring kernel with the refill DMAs interleaved with the row arithmetic (A/B in separate processes)
T=gpurun_out/r3e; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn_stream" > $T/pytest_tattn.log 2>&1; tail -3 $T/pytest_tattn.log
for r in 1 2; do for g in 0 3; do L2D_TATTN_RING=$g timeout 200 python tools/tattn_time.py >> $T/tattn_ilv_ab.txt 2>> $T/err.log; done; done
L2D_TATTN_RING=0 timeout 200 python tools/tattn_time.py --height 512 --width 768 --window 24 >> $T/tattn_ilv_ab.txt 2>> $T/err.log
cat $T/tattn_ilv_ab.txt; tail -3 $T/err.log
