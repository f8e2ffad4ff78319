T=gpurun_out/r3o; mkdir -p $T
cp live2diff_amd/rowgemm_tuned.json $T/rowgemm_tuned.json
for cfgs in "256 256 1 12" "512 768 2 24" "512 512 4 16" "576 1024 2 40"; do set -- $cfgs
  timeout 500 python tools/rowgemm_tune.py --height $1 --width $2 --denoise-steps $3 --window $4 --out $T/rowgemm_tuned.json --report $T/rowgemm_tune_$1x$2_n$3_L$4.txt > /dev/null 2>> $T/err.log
  tail -1 $T/rowgemm_tune_$1x$2_n$3_L$4.txt
done
cp $T/rowgemm_tuned.json live2diff_amd/rowgemm_tuned.json
for cfgs in "256 256 1 12" "512 768 2 24" "512 512 4 16" "576 1024 2 40"; do set -- $cfgs
  timeout 300 python bench.py --height $1 --width $2 --denoise-steps $3 --window $4 --steps 30 --warmup 5 --no-cpu-baseline --breakdown 0 --whole-frame 0 > $T/bench_$1x$2_n$3_L$4.json 2>> $T/err.log
  python -c "
import json
d=json.loads(open('$T/bench_$1x$2_n$3_L$4.json').read().strip().splitlines()[-1]); print(d['config']['workload'][:40], d['value'], d['ms_per_step'])"
done
tail -3 $T/err.log
