T=gpurun_out/r3p; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_rowgemm.py -q -x > $T/pytest_rowgemm.log 2>&1; tail -5 $T/pytest_rowgemm.log
cp live2diff_amd/rowgemm_tuned.json $T/rowgemm_tuned.json
for cfgs in "512 512 2 16" "512 768 2 24" "512 512 4 16" "576 1024 2 40"; do set -- $cfgs
  timeout 500 python tools/rowgemm_tune.py --height $1 --width $2 --denoise-steps $3 --window $4 --out $T/rowgemm_tuned.json --report $T/rowgemm_tune_mt4_$1x$2_n$3_L$4.txt > /dev/null 2>> $T/err.log
  grep -E "\*|per frame" $T/rowgemm_tune_mt4_$1x$2_n$3_L$4.txt | tail -12
done
tail -3 $T/err.log
