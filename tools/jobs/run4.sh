mkdir -p gpurun_out/r2d
(timeout 500 python -m pytest tests/test_gpu_vae.py tests/test_gpu_stream_step.py -m gpu -q --durations=6 > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log); tail -22 gpurun_out/r2d/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 40 --per-op gpurun_out/r2d/per_op.csv > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r2d/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['whole_frame'])"
grep -E "flash_attn" gpurun_out/r2d/per_op.csv | sort -t, -k3,3 | awk -F, '{k=\$3; n[k]++; s[k]+=\$4} END{for(k in n) print k, n[k], s[k]/n[k]}'
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 400 bash tools/pmc_ops.sh gpurun_out/r2d/pmc_ops.txt > gpurun_out/r2d/pmc.log 2>&1; tail -3 gpurun_out/r2d/pmc.log
grep -A14 "flash_ring" gpurun_out/r2d/pmc_ops.txt | head -80
