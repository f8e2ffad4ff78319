T=gpurun_out/r3sb; mkdir -p $T
PROBE_GRAPH=1 timeout 300 python tools/two_stream_probe.py > $T/two_stream_probe_graph.json 2> $T/err.log; cat $T/two_stream_probe_graph.json
PROBE_GRAPH=0 timeout 300 python tools/two_stream_probe.py > $T/two_stream_probe_direct.json 2>> $T/err.log; cat $T/two_stream_probe_direct.json
