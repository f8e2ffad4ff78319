# round 6, third session: bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU, RCCL), with the one GPU
# of this box: rendezvous, packed-weight replication path, barrier + max-over-ranks timing, one JSON line from rank 0
T=gpurun_out/r6ae; mkdir -p $T
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --whole-frame 0 --multi-stream 0 > $T/bench_torchrun_n1.json 2> $T/torchrun.err; echo rc=$?
tail -1 $T/bench_torchrun_n1.json | cut -c1-400; tail -3 $T/torchrun.err | cut -c1-300
