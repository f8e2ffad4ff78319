# round 6, third session: evidence at the product-code commit after the GroupNorm one-trip blocks and the fallback rules: the whole -m gpu suite
# (no -x), smoke, evidence run (bench line, kernel trace, in-frame trace, HBM traffic counters, matrix-pipe counters)
TAG=${1:-round6_final3}
T=gpurun_out/r6z3; mkdir -p $T
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider > $T/pytest_gpu.log 2>&1; tail -3 $T/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $T/smoke.log 2>&1; tail -2 $T/smoke.log
timeout 1500 bash tools/profile_round.sh $TAG > $T/profile_round.log 2>&1; tail -6 $T/profile_round.log
