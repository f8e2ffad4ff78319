mkdir -p gpurun_out/r6k
run() { # name H W N L
  for v in 1 0 1 0; do
    L2D_CCONV=$v timeout 400 python bench.py --steps 20 --warmup 5 --height $2 --width $3 --denoise-steps $4 --window $5 --no-cpu-baseline --breakdown 0 --multi-stream 0 --whole-frame 0 2>/dev/null | tail -1 > /tmp/b.json
    python -c "import json;d=json.load(open('/tmp/b.json'));print('$1', $v, d['ms_per_step'])"
  done
}
run cfg3 512 768 2 24
run cfg4 512 512 4 16
run cfg5 576 1024 2 40
run cfg1 256 256 1 12
