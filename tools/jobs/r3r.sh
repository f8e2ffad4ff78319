T=gpurun_out/r3r; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_pconv.py tests/test_gpu_vae.py tests/test_gpu_midas.py -q -x > $T/pytest.log 2>&1; tail -5 $T/pytest.log
for p in 1 0; do
L2D_PCONV=$p timeout 200 python tools/midas_time.py 1 > $T/midas_time_pconv$p.json 2>> $T/err.log; cut -c1-400 $T/midas_time_pconv$p.json
L2D_PCONV=$p timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown 0 > $T/bench_pconv$p.json 2>> $T/err.log
python -c "
import json
d=json.load(open('$T/bench_pconv$p.json')); print('L2D_PCONV=$p', d['value'], d['ms_per_step'], {k:d['whole_frame'][k] for k in ('frames_per_s','ms_per_frame','depth_time_ema_ms')}, d['whole_frame'].get('vae_launches'))"
done
tail -3 $T/err.log
