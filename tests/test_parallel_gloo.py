"""world_size-2 `gloo` test of the multi-GPU path (CPU): weight broadcast + checksum, barrier, the max / sum
reductions around the timed region, and that two ranks driving independent streams keep independent state."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, os.environ["L2D_ROOT"])
    from live2diff_amd import parallel
    from live2diff_amd.config import tiny_config
    from live2diff_amd.weights import random_state_dict, unet_param_spec
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init, ring_buffer_update
    rank, world, local = parallel.init_distributed("gloo")
    assert world == 2
    cfg = tiny_config(channels=(32, 64, 64, 64), cross_attention_dim=64)
    spec = unet_param_spec(cfg)
    sd = random_state_dict(cfg, dtype=torch.float16) if rank == 0 else None
    ref = random_state_dict(cfg, dtype=torch.float16)
    for algo in ("scatter_allgather", "broadcast"):      # SURVEY 8e: scatter 1/G shards + all-gather (default); plain bcast
        out = parallel.broadcast_state_dict(spec, sd, "cpu", bucket_elems=200_001, algo=algo)   # several ragged buckets
        assert set(out) == set(ref)
        for k in ref:
            assert out[k].shape == ref[k].shape and torch.equal(out[k], ref[k]), (algo, k)
    assert parallel.gather_floats(10.0 + rank) == [10.0, 11.0]
    # independent streams: rank r advances its ring buffer r+3 frames; states differ, nothing is shared
    rb = ring_buffer_init(2)
    for _ in range(rank + 3):
        ring_buffer_update(*rb)
    parallel.barrier()
    assert parallel.max_over_ranks(float(rank + 1)) == 2.0
    assert parallel.sum_over_ranks(float(rb[2][0])) == float((8 + 3) + (8 + 4))
    # the checksum is position sensitive: two equal-sized shards in the wrong order must change it
    t = torch.arange(64, dtype=torch.float16)
    assert not torch.equal(parallel._checksum(t, 64), parallel._checksum(torch.cat([t[32:], t[:32]]), 64))
    parallel.barrier()
    print("rank", rank, "ok", flush=True)
    torch.distributed.destroy_process_group()
""")


def test_two_rank_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), L2D_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()
        assert f"rank {r} ok" in out.decode()
