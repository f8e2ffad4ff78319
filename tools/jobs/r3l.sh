T=gpurun_out/r3l; mkdir -p $T
timeout 400 python tools/frame_each.py --csv $T/frame_each_cfg2.csv > $T/frame_each_cfg2.txt 2> $T/err.log; head -70 $T/frame_each_cfg2.txt; tail -3 $T/err.log
