for v in 8 12 8 12 16; do
  L2D_LIB=live2diff_amd/ablate/libl2d_rowchain_RC_RD$v.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --multi-stream 0 --whole-frame 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RC_RD=$v', d['ms_per_step'], d['kernels']['rowchain_kernel']['ms_per_frame'])"
done
