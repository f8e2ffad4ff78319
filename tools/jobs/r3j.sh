T=gpurun_out/r3j; mkdir -p $T
for g in 7 8; do L2D_TATTN_RING=$g timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn_stream" > $T/pytest_tattn_$g.log 2>&1; tail -2 $T/pytest_tattn_$g.log; done
for r in 1 2; do for g in 0 7 8; do L2D_TATTN_RING=$g timeout 200 python tools/tattn_time.py >> $T/tattn_long_ab.txt 2>> $T/err.log; done; done
for g in 0 7 8; do L2D_TATTN_RING=$g timeout 200 python tools/tattn_time.py --height 512 --width 768 --window 24 >> $T/tattn_long_ab.txt 2>> $T/err.log; done
for g in 0 7 8; do L2D_TATTN_RING=$g timeout 200 python tools/tattn_time.py --denoise-steps 4 >> $T/tattn_long_ab.txt 2>> $T/err.log; done
cat $T/tattn_long_ab.txt; tail -3 $T/err.log
