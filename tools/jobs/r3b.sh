# round 3, call 2: row-GEMM block phases (probe build), the ring temporal-attention kernel at L = 24 / 40, frame with the tuned table
T=gpurun_out/r3b; mkdir -p $T
L2D_LIB=live2diff_amd/libl2d_hip_probes.so timeout 300 python tools/rowgemm_probe.py > $T/rowgemm_block_phases.txt 2>&1; cat $T/rowgemm_block_phases.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "tattn_stream" > $T/pytest_tattn.log 2>&1; tail -5 $T/pytest_tattn.log
timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --whole-frame 1 > $T/bench_cfg2.json 2>> $T/bench.err
python -c "
import json
d=json.loads(open('$T/bench_cfg2.json').read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'], d['config']['plan_launches']); print({k:(v['launches'],v['ms_per_frame']) for k,v in d.get('kernels',{}).items()}); print(d.get('whole_frame'))"
for cfgs in "512 768 2 24" "576 1024 2 40"; do set -- $cfgs
  timeout 400 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 20 --warmup 5 --no-cpu-baseline --whole-frame 0 > $T/bench_$1x$2_L$4.json 2>> $T/bench.err
  python -c "
import json
d=json.loads(open('$T/bench_$1x$2_L$4.json').read().strip().splitlines()[-1]); print('$1x$2 L$4', d['value'], d['ms_per_step'], d['config']['plan_launches'], d.get('roofline_kv_cache_kernel')); print({k:(v['launches'],v['ms_per_frame']) for k,v in d.get('kernels',{}).items()})"
done
tail -5 $T/bench.err
