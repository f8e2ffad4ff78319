# round 4: the whole library without packed fp32 ops: smoke, kernel-level GPU tests, frame stress, bench line
T=gpurun_out/r4z2; mkdir -p $T
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pconv.py tests/test_gpu_wsgemm.py tests/test_gpu_rowgemm.py tests/test_gpu_vae.py -x -q 2>&1 | tail -2
REPS=300 timeout 300 python tools/frame_stress.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $T/frame_stress.txt
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $T/bench_cfg2.json 2>> $T/err.log
python -c "
import json
d=json.loads(open('$T/bench_cfg2.json').read().strip().splitlines()[-1]); w=d.get('whole_frame') or {}; print(d['value'], d['ms_per_step'], w.get('frames_per_s'), d['roofline_small_m']['frac'], d['roofline_flash']['frac'], d['roofline_kv_cache_kernel']['frac'])"
