# round 4, first GPU call: the weight-streaming GEMM (wsgemm.hip) -- kernel parity tests through the C ABI, then the cold-weight
# probe of the frame's M <= 512 shapes over a grid of schedules against the kernels that serve them today
T=gpurun_out/r4a; mkdir -p $T
timeout 1200 python -m pytest tests/test_gpu_wsgemm.py -q > $T/pytest_wsgemm.log 2>&1; tail -40 $T/pytest_wsgemm.log
timeout 900 python tools/wsgemm_probe.py --out $T/wsgemm_probe.json > $T/probe.log 2>&1; grep -A9 "^==" $T/probe.log | head -220
