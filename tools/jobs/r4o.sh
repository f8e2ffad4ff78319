T=gpurun_out/r4o; mkdir -p $T
for f in "" "1,1,1,1" "1,1,2,1" "2,1,1,1" "1,1,1,2"; do
  echo "== L2D_WSGEMM_FORCE='$f'"
  L2D_WSGEMM_FORCE="$f" L2D_WSGEMM_NO_TABLE=1 timeout 600 python -m pytest tests/test_gpu_midas.py -q -x -k push_pop 2>&1 | grep -E "AssertionError:|passed|failed" | head -3
done
echo "== default again (determinism of the failure)"
timeout 600 python -m pytest tests/test_gpu_midas.py -q -x -k push_pop 2>&1 | grep -E "AssertionError:|passed|failed" | head -3
