T=gpurun_out/r3m; mkdir -p $T
for r in 1 2; do for k in 1280 640 320; do
L2D_ROWGEMM_FF1_MAX_K=$k timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --breakdown 0 --whole-frame 0 > $T/bench_ff1_$k.$r.json 2>> $T/err.log
python -c "import json,sys; d=json.load(open('$T/bench_ff1_$k.$r.json')); print('FF1_MAX_K=$k', d['value'], d['ms_per_step'], d['config']['plan_launches'])"
done; done
tail -3 $T/err.log
