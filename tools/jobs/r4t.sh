for v in v0 v1 v0 v1; do
echo "=== $v"
L2D_LIB=$PWD/build_variants/$v.so SIDE=1 REPS=2400 SCHEDS="4,1,2,1;5,1,2,1" timeout 600 python tools/wsgemm_diag.py 2>&1 | grep "differing runs"
done
