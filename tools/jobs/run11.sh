mkdir -p gpurun_out/r2k
(timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "igemm or groupnorm" > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest.log); tail -4 gpurun_out/r2k/pytest.log
bash tools/tune_igemm_incr.sh gpurun_out/r2k/tune 1,1,4 1,1,10 1,1,0 1,2,10 1,2,4 1,3,10 1,4,10 1,6,10 1,12,10 2,1,1 2,1,5 2,1,9 2,2,1 2,3,1 2,4,9 1,1,5 1,3,5 2>&1 | tail -60
