T=gpurun_out/r3v; mkdir -p $T
PROBE_STREAMS=8 timeout 600 python tools/multi_stream_probe.py > $T/multi_stream_probe_s8.json 2> $T/err.log; cat $T/multi_stream_probe_s8.json; tail -3 $T/err.log
