T=gpurun_out/r3y; mkdir -p $T
L2D_TATTN_RING=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn_stream" > $T/pytest_tattn_ring4.log 2>&1; tail -2 $T/pytest_tattn_ring4.log
for cfgs in "384 640 2 16" "512 512 3 16" "320 320 2 16" "256 448 1 12"; do set -- $cfgs
  timeout 500 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 20 --warmup 5 --cpu-frames 0 --whole-frame 0 --multi-stream 0 > $T/bench_$1x$2_n$3_L$4.json 2>> $T/err.log
  python -c "
import json
d=json.loads(open('$T/bench_$1x$2_n$3_L$4.json').read().strip().splitlines()[-1]); print('$1x$2 N$3 L$4', d['value'], d['ms_per_step'], d['config']['plan_launches'], d['config']['output_finite'], d.get('parity_vs_oracle_full_size'))"
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown 0 --whole-frame 0 --multi-stream 0 --graph 1 > $T/bench_graph.json 2>> $T/err.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown 0 --whole-frame 0 --multi-stream 0 --device-step 1 > $T/bench_devstep.json 2>> $T/err.log
python -c "
import json
for n in ('graph','devstep'):
    d=json.load(open('$T/bench_%s.json' % n)); print(n, d['value'], d['ms_per_step'], d['config']['output_finite'])"
tail -3 $T/err.log
