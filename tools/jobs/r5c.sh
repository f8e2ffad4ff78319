# round 5: HBM-traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) -- staged, every pass under its own
# short timeout: round 4's passes over bench.py crashed / hung inside rocprofv3.  1. a trivial torch program (is --pmc alive on this
# box at all?)  2. the frame replay target.
T=gpurun_out/r5c; mkdir -p $T; export TMPDIR=/tmp; rm -rf /tmp/pmc5; mkdir -p /tmp/pmc5
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc5/triv -o p -- python -c "
import torch
x = torch.ones(1 << 24, device='cuda'); y = (x * 2).sum().item(); print('trivial ok', y)" > $GRAFT_REPO_ROOT/$T/triv.log 2>&1
rc=$?; echo "trivial pass rc=$rc"; tail -2 $GRAFT_REPO_ROOT/$T/triv.log
if [ $rc -ne 0 ]; then echo "PMC is not usable on this box"; exit 0; fi
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && FRAMES=3 timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc5/$c -o p -- python $GRAFT_REPO_ROOT/tools/traffic_frame.py > $GRAFT_REPO_ROOT/$T/pmc_$c.log 2>&1 ); echo "$c rc=$?"; tail -1 $T/pmc_$c.log
done
python tools/pmc_summary.py $T/round5_pmc_frame.txt $(find /tmp/pmc5/FETCH_SIZE /tmp/pmc5/WRITE_SIZE -name "*.db") --traffic $T/traffic.json
head -c 1500 $T/traffic.json
