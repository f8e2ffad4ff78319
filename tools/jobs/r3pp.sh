T=gpurun_out/r3pp; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_midas.py -q -x -k "pipelined or whole_pipeline" > $T/pytest_pipelined.log 2>&1; tail -6 $T/pytest_pipelined.log
timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown 0 --multi-stream 0 > $T/bench.json 2>> $T/err.log
python -c "
import json
d=json.load(open('$T/bench.json')); w=d['whole_frame']; print(d['value'], d['ms_per_step'], w['frames_per_s'], w['ms_per_frame'], w.get('pipelined'))"
grep -v amdgpu $T/err.log | tail -3
