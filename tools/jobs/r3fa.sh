T=gpurun_out/r3fa; mkdir -p $T
for lw in 1 2; do L2D_FLASH_LW=$lw timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "flash" > $T/pytest_flash_lw$lw.log 2>&1; tail -2 $T/pytest_flash_lw$lw.log; done
for r in 1 2; do for lw in 0 1 2; do echo "L2D_FLASH_LW=$lw" >> $T/flash_lw_ab.txt; L2D_FLASH_LW=$lw timeout 200 python tools/flash_time.py 2 3 >> $T/flash_lw_ab.txt 2>> $T/err.log; done; done
cat $T/flash_lw_ab.txt; grep -v amdgpu $T/err.log | tail -3
