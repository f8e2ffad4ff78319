"""world_size-2 `gloo` test of the multi-GPU path (CPU): weight broadcast + checksum (raw state dict and PACKED weights: rank 0
packs once, rank 1 builds its plan from the replicated packed tensors), barrier, the max / sum reductions around the timed
region, and that two ranks driving independent streams keep independent state."""
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
    # SURVEY 8e as written: rank 0 packs ONCE, the PACKED weights (fp16 matrices + fp32 biases / column sums) are replicated, rank 1
    # builds its instance from what it received -- no packing pass there -- and both ranks hold identical packed tensors and plans
    from live2diff_amd import _lib
    from live2diff_amd.unet_hip import HipStreamingUNet, PackedWeights
    _lib.set_dry_run(True)
    cfg = tiny_config()                                                    # (the widths the dry-run plan tests use)
    ref = random_state_dict(cfg, dtype=torch.float16)
    u0 = HipStreamingUNet(ref, cfg, 16, 16, 2, device="cpu") if rank == 0 else None
    packed = parallel.replicate_packed_weights(u0, "cpu")
    assert isinstance(packed, PackedWeights)
    calls = []
    orig = HipStreamingUNet._pack_weights
    HipStreamingUNet._pack_weights = lambda self, sd_: calls.append(1) or orig(self, sd_)
    u = u0 if rank == 0 else HipStreamingUNet(packed, cfg, 16, 16, 2, device="cpu")
    assert not calls, "a receiving rank must not run the packing pass"
    mine = HipStreamingUNet(ref, cfg, 16, 16, 2, device="cpu")          # (what this rank WOULD have packed itself)
    assert set(u.W) == set(mine.W)
    for k in mine.W:
        assert u.W[k].dtype == mine.W[k].dtype and torch.equal(u.W[k], mine.W[k]), k
    assert (u.temb_offsets, u.text_offsets, u.n_map_blocks, u.temb_total, u.text_total) == (mine.temb_offsets, mine.text_offsets, mine.n_map_blocks, mine.temb_total, mine.text_total)
    st = u._plan("stream", u.prepare_cache(2))
    st.pl.run(stream=0)                                                    # every op validates against the received tensors
    assert len(st.pl) == len(mine._plan("stream", mine.prepare_cache(2)).pl)
    try:
        HipStreamingUNet(packed, cfg, 8, 8, 2, device="cpu")              # packed for another latent size: refused, not mis-used
        raise SystemExit("a packed layout for another latent size was accepted")
    except ValueError:
        pass
    _lib.set_dry_run(False)
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
