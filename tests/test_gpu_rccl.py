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
    # the PACKED-weight replication (what bench.py --gpus N uses): object broadcast of the metadata, uint8 scatter + all-gather of
    # mixed fp16 / fp32 tensors, byte checksum -- through RCCL with device tensors; the replica builds a working instance
    from live2diff_amd.unet_hip import HipStreamingUNet
    cfg2 = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    sd2 = {k: v.cuda() for k, v in random_state_dict(cfg2, dtype=torch.float16).items()}
    u0 = HipStreamingUNet(sd2, cfg2, 16, 16, 2)
    W = parallel.replicate_tensors(u0.W, "cuda", bucket_bytes=3_000_001, force_collectives=True)
    assert set(W) == set(u0.W) and all(W[k].is_cuda and W[k].dtype == u0.W[k].dtype and torch.equal(W[k], u0.W[k]) for k in W)
    assert all(W[k].data_ptr() % 16 == 0 for k in W)
    from live2diff_amd.unet_hip import PackedWeights
    u1 = HipStreamingUNet(PackedWeights(W, u0._packed_meta()), cfg2, 16, 16, 2)
    kv0, kv1 = u0.prepare_cache(2), u1.prepare_cache(2)
    g = torch.Generator(device="cuda").manual_seed(5)
    rn = lambda *s_: torch.randn(*s_, generator=g, device="cuda", dtype=torch.float16)
    for c0, c1 in zip(kv0, kv1):
        c0.normal_(generator=g); c1.copy_(c0)
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init
    rb = ring_buffer_init(2, cfg2.window_size, cfg2.sink_size)
    args = dict(encoder_hidden_states=rn(2, 77, 64), temporal_attention_mask=rb[0].half().cuda(), depth_sample=rn(2, 4, 1, 16, 16),
                pe_idx=rb[1].cuda(), update_idx=rb[2].cuda())
    x, ts = rn(2, 4, 1, 16, 16), torch.tensor([399, 199], device="cuda")
    a = u0(x, ts, kv_cache=kv0, **args)["sample"].clone()
    b = u1(x, ts, kv_cache=kv1, **args)["sample"].clone()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.equal(a, b), "an instance built from replicated packed weights must give the same frame"
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
