T=gpurun_out/r3u; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_unet.py -q -x -k "cfg2" > $T/pytest_cfg2.log 2>&1; tail -5 $T/pytest_cfg2.log
timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown 0 > $T/bench.json 2>> $T/err.log
python -c "
import json
d=json.load(open('$T/bench.json')); print(d['value'], d['ms_per_step'], d['whole_frame']['frames_per_s'], json.dumps(d['streams_per_gpu']))"
timeout 400 python bench.py --height 576 --width 1024 --window 40 --steps 20 --warmup 5 --no-cpu-baseline --breakdown 0 --whole-frame 0 > $T/bench_cfg5.json 2>> $T/err.log
python -c "
import json
d=json.load(open('$T/bench_cfg5.json')); print(d['value'], d['ms_per_step'], json.dumps(d['streams_per_gpu']))"
tail -3 $T/err.log
