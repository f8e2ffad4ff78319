T=gpurun_out/r3g; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn_stream" > $T/pytest_tattn.log 2>&1; tail -3 $T/pytest_tattn.log
for r in 1 2; do for g in 0 5 6 4; do L2D_TATTN_RING=$g timeout 200 python tools/tattn_time.py >> $T/tattn_lw_ab.txt 2>> $T/err.log; done; done
for g in 0 5; do L2D_TATTN_RING=$g timeout 200 python tools/tattn_time.py --height 512 --width 768 --window 24 >> $T/tattn_lw_ab.txt 2>> $T/err.log; done
cat $T/tattn_lw_ab.txt; tail -3 $T/err.log
