SIDE=1 REPS=2400 SCHEDS="4,1,2,1;4,1,1,1" timeout 900 python tools/wsgemm_diag.py 2>&1 | grep -v amdgpu.ids | tail -40
