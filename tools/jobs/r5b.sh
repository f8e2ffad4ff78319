# round 5, second GPU call: the row-symmetric time-embedding GEMV -> flip symmetry restored?  then the whole GPU suite (no -x)
T=gpurun_out/r5b; mkdir -p $T
REPS=6 PHASES=1 timeout 300 python tools/flip_diag.py > $T/p1.txt 2>&1; tail -2 $T/p1.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $T/pytest_gpu.log 2>&1; tail -30 $T/pytest_gpu.log
