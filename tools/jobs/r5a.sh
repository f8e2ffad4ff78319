# round 5, first GPU call: does the flip-symmetry failure of the driver's round-4 run reproduce at HEAD, and where does it start?
T=gpurun_out/r5a; mkdir -p $T
export REPS=12
PHASES=1 timeout 300 python tools/flip_diag.py > $T/p1_default.txt 2>&1; tail -4 $T/p1_default.txt
PHASES=1 L2D_WSGEMM=0 timeout 300 python tools/flip_diag.py > $T/p1_ws0.txt 2>&1; tail -2 $T/p1_ws0.txt
PHASES=1 L2D_WSGEMM_NO_TABLE=1 timeout 300 python tools/flip_diag.py > $T/p1_notable.txt 2>&1; tail -2 $T/p1_notable.txt
PHASES=1 POISON=1 timeout 300 python tools/flip_diag.py > $T/p1_poison.txt 2>&1; tail -2 $T/p1_poison.txt
REPS=6 PHASES=23 timeout 600 python tools/flip_diag.py > $T/p23_default.txt 2>&1; tail -30 $T/p23_default.txt
REPS=6 PHASES=3 POISON=1 timeout 400 python tools/flip_diag.py > $T/p3_poison.txt 2>&1; tail -12 $T/p3_poison.txt
