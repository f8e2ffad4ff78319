T=gpurun_out/r3t; mkdir -p $T
timeout 600 python tools/multi_stream_probe.py > $T/multi_stream_probe.json 2> $T/err.log; cat $T/multi_stream_probe.json; tail -3 $T/err.log
