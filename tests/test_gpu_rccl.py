"""-m gpu: the multi-GPU path's collectives on the REAL `nccl` (= RCCL) backend, as a one-rank process group.

The GPU boxes of this build have one GPU, so the N > 1 path is covered by the world_size-2 `gloo` test on CPU
(tests/test_parallel_gloo.py).  What that cannot show is that the calls `parallel.py` makes exist and run on RCCL with
device tensors: `dist.scatter`, `all_gather_into_tensor`, the fp64 MIN / MAX `all_reduce` of the checksum, `all_gather` of
per-rank results, `barrier`.  This test forces them through a world-size-1 group (`force_collectives=True` drops the
`world == 1` early return) in a subprocess, so the process group cannot leak into other tests."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, os.environ["L2D_ROOT"])
    import torch.distributed as dist
    from live2diff_amd import parallel
    from live2diff_amd.config import tiny_config
    from live2diff_amd.weights import random_state_dict, unet_param_spec
    rank, world, local = parallel.init_distributed("nccl", force=True)
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    cfg = tiny_config(channels=(32, 64, 64, 64), cross_attention_dim=64)
    spec = unet_param_spec(cfg)
    sd = random_state_dict(cfg, dtype=torch.float16)
    for algo in ("scatter_allgather", "broadcast"):
        out = parallel.broadcast_state_dict(spec, sd, "cuda", bucket_elems=200_001, algo=algo, force_collectives=True)
        assert set(out) == set(sd)
        for k in sd:
            assert out[k].is_cuda and torch.equal(out[k].cpu(), sd[k]), (algo, k)
    assert parallel.gather_floats(3.5, device="cuda") == [3.5]
    parallel.barrier()
    assert parallel.max_over_ranks(2.0, device="cuda") == 2.0 and parallel.sum_over_ranks(2.0, device="cuda") == 2.0
    torch.cuda.synchronize()
    print("rccl world-1 ok", flush=True)
    dist.destroy_process_group()
""")


def test_rccl_world1_collectives(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               L2D_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0, p.stdout.decode()
    assert "rccl world-1 ok" in p.stdout.decode()
