T=gpurun_out/r3x; mkdir -p $T
cp live2diff_amd/igemm_tuned.json $T/igemm_tuned.json
timeout 900 python tools/igemm_tune_each.py --out $T/igemm_tuned.json --report $T/igemm_tune_each_cfg2.txt > /dev/null 2> $T/err.log; head -50 $T/igemm_tune_each_cfg2.txt; tail -2 $T/igemm_tune_each_cfg2.txt; tail -5 $T/err.log
cp live2diff_amd/igemm_tuned.json $T/igemm_tuned_before.json
for r in 1 2; do
cp $T/igemm_tuned_before.json live2diff_amd/igemm_tuned.json
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --breakdown 0 --whole-frame 0 --multi-stream 0 > $T/bench_before.$r.json 2>> $T/err.log
cp $T/igemm_tuned.json live2diff_amd/igemm_tuned.json
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --breakdown 0 --whole-frame 0 --multi-stream 0 > $T/bench_after.$r.json 2>> $T/err.log
python -c "
import json
a=json.load(open('$T/bench_before.$r.json')); b=json.load(open('$T/bench_after.$r.json')); print('before', a['ms_per_step'], 'after', b['ms_per_step'])"
done
