T=gpurun_out/r3chk; mkdir -p $T
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --whole-frame 0 --multi-stream 0 > $T/bench.json 2> $T/err.log; echo rc=$?
python -c "
import json
d=json.load(open('$T/bench.json')); print(d['value'], d['roofline'])"
