# round 5, final evidence at the final product-code commit (block chains in): the whole -m gpu suite (no -x), smoke, evidence run
T=gpurun_out/r5z3; mkdir -p $T
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $T/pytest_gpu.log 2>&1; tail -3 $T/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -2 $T/smoke.log
timeout 1200 bash tools/profile_round.sh round5_final > $T/profile_round.log 2>&1; tail -3 $T/profile_round.log
