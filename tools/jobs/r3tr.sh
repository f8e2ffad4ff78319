T=gpurun_out/r3tr; mkdir -p $T
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --whole-frame 0 --multi-stream 0 > $T/bench_torchrun_n1.json 2> $T/err.log; echo rc=$?
tail -c 600 $T/bench_torchrun_n1.json; grep -v amdgpu $T/err.log | tail -3
timeout 120 python bench.py --gpus 2 --steps 5 --warmup 2 > $T/bench_gpus2.out 2>&1; echo "gpus2 rc=$?"; tail -3 $T/bench_gpus2.out
