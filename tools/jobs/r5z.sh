# round 5, final evidence at the final commit: the whole -m gpu suite (no -x), smoke, then the round-end evidence run
T=gpurun_out/r5z; mkdir -p $T
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $T/pytest_gpu.log 2>&1; tail -3 $T/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -2 $T/smoke.log
timeout 1200 bash tools/profile_round.sh round5_final > $T/profile_round.log 2>&1; tail -5 $T/profile_round.log
