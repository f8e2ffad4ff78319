echo "== round-3 kernels only"; L2D_WSGEMM=0 L2D_LIB=$PWD/build_variants/v0.so REPS=800 timeout 600 python tools/frame_stress.py 2>&1 | grep -v amdgpu.ids | tail -8
echo "== wsgemm, packed fp32 epilogue (v0)"; L2D_LIB=$PWD/build_variants/v0.so REPS=400 timeout 600 python tools/frame_stress.py 2>&1 | grep -v amdgpu.ids | tail -8
echo "== wsgemm, scalar fp32 epilogue (v1)"; L2D_LIB=$PWD/build_variants/v1.so REPS=800 timeout 600 python tools/frame_stress.py 2>&1 | grep -v amdgpu.ids | tail -8
