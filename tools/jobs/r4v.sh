mkdir -p gpurun_out/r4v
SIDE=1 REPS=2400 SCHEDS="4,1,2,1;5,1,2,1" timeout 600 python tools/wsgemm_diag.py 2>&1 | grep "differing runs" | tee gpurun_out/r4v/diag.txt
REPS=600 timeout 600 python tools/frame_stress.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r4v/frame_stress.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r4v/pytest.txt
