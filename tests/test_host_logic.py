"""CPU tests (-m "not gpu"): host-side logic against the reference goldens, and the C-ABI library surface."""
import ctypes
import sys
import os
import re

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 3, 4])
def test_ring_buffer_trace_matches_reference(golden, n):
    """40-frame trace of (attn_bias, pe_idx, update_idx) captured from the reference's
    initialize_attn_bias_pe_and_update_idx / update_attn_bias (pipeline :403-438)."""
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init, ring_buffer_update
    g = golden("state_machine")
    b, p, u = ring_buffer_init(n)
    for f in range(41):
        assert torch.equal(b, torch.from_numpy(g[f"bias_n{n}"][f])), f
        assert torch.equal(p, torch.from_numpy(g[f"pe_idx_n{n}"][f])), f
        assert torch.equal(u, torch.from_numpy(g[f"update_idx_n{n}"][f])), f
        ring_buffer_update(b, p, u)


def test_ring_buffer_general_windows():
    """N=1 and other sink/window sizes (undefined in the reference): invariants of the steady state."""
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init, ring_buffer_update
    for n, L, S in [(1, 12, 4), (2, 24, 8), (2, 40, 8), (4, 16, 8)]:
        b, p, u = ring_buffer_init(n, L, S)
        for f in range(3 * L):
            ring_buffer_update(b, p, u, L, S)
        assert (b == 0).all()
        for i in range(n):
            assert sorted(p[i].tolist()) == list(range(L))          # a permutation
            assert p[i, :S].tolist() == list(range(S))              # sink slots keep PE 0..S-1
            assert p[i, u[i]] == L - 1 and S <= u[i] < L            # newest frame always gets the largest PE


def test_lcm_step_and_add_noise(golden):
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth as S
    g = golden("state_machine")
    fake = S.__new__(S)
    fake.alpha_prod_t_sqrt, fake.beta_prod_t_sqrt = torch.from_numpy(g["lcm_alpha"]), torch.from_numpy(g["lcm_beta"])
    fake.c_skip, fake.c_out = torch.from_numpy(g["lcm_c_skip"]), torch.from_numpy(g["lcm_c_out"])
    x, eps = torch.from_numpy(g["lcm_x"]), torch.from_numpy(g["lcm_eps"])
    assert torch.allclose(S.scheduler_step_batch(fake, eps, x), torch.from_numpy(g["lcm_x0"]), atol=1e-6)
    assert torch.allclose(S.scheduler_step_batch(fake, eps[1:2], x[1:2], 1), torch.from_numpy(g["lcm_x0_idx1"]), atol=1e-6)
    assert torch.allclose(S.add_noise(fake, x[1:2], eps[1:2], 1), torch.from_numpy(g["add_noise_1"]), atol=1e-6)


def test_lcm_schedule_documented_values():
    """diffusers-0.25.0 LCMScheduler semantics (stub-pinned): t = 999 - 20 i for 50 steps; demo t_index [30,40]
    -> [399,199]; boundary scalings at t=399."""
    from live2diff_amd.scheduler import LCMSchedule
    s = LCMSchedule()
    ts = s.set_timesteps(50)
    assert ts[0] == 999 and ts[30] == 399 and ts[40] == 199 and ts[-1] == 19
    c_skip, c_out = s.get_scalings_for_boundary_condition_discrete(399)
    assert abs(float(c_skip) - 0.25 / (3990.0 ** 2 + 0.25)) < 1e-12
    assert abs(float(c_out) - 3990.0 / (3990.0 ** 2 + 0.25) ** 0.5) < 1e-7
    assert abs(float(s.alphas_cumprod[0]) - (1 - 0.00085)) < 1e-7
    # 0.25.0 picks floor(linspace(0, 50, n, endpoint=False)) of the reversed origin steps -- NOT origin[::-skip][:n];
    # the two differ whenever 50 % n != 0 (reference configs/pixart.yaml runs 4 steps with strength 0.6 -> [499, 259])
    assert s.set_timesteps(4).tolist() == [999, 759, 499, 259]
    assert s.set_timesteps(6).tolist() == [999, 839, 679, 499, 339, 179]
    assert s.set_timesteps(50).tolist() == [999 - 20 * i for i in range(50)]
    assert s.set_timesteps(1).tolist() == [999]
    with pytest.raises(ValueError):
        s.set_timesteps(51)


def test_param_count_and_layout():
    from live2diff_amd.config import motion_module_layout, sd15_config
    from live2diff_amd.weights import count_params
    cfg = sd15_config()
    assert count_params(cfg) == 1277745188            # reference classes instantiated at SD-1.5 widths: 1 277.7 M
    lay = motion_module_layout(cfg, 64, 64)
    assert len(lay) == 40
    assert [c for c, *_ in lay[:16]] == [320] * 4 + [640] * 4 + [1280] * 8
    assert [(hh, ww) for _, hh, ww, _ in lay[16:22]] == [(8, 8)] * 6 and lay[39][:3] == (320, 64, 64)
    kv_bytes = sum(2 * 2 * hh * ww * 16 * c * 2 for c, hh, ww, _ in lay)
    assert abs(kv_bytes / 1e9 - 3.04) < 0.02          # 3.04 GB at cfg-2


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly the functions include/l2d.h declares."""
    from live2diff_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "l2d.h")).read()
    body = hdr[hdr.index("typedef struct l2d_op"):]
    names = re.findall(r"\b(l2d_[a-z_]+)\s*\(", body)
    assert len(names) >= 9
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert _lib.lib.l2d_abi_version() == _lib.ABI_VERSION == 6
    assert ctypes.sizeof(_lib.L2dOp) == 312          # ABI v4: 16 pointers + 32 ints + 4 int64 + 4 floats (+ kind, tag)
    # error path without a device: refused loudly, no fallback
    ops = (_lib.L2dOp * 1)()
    ops[0].kind = 99
    assert _lib.lib.l2d_run_ops(ops, 1, None) == -1
    assert b"unknown op kind" in _lib.lib.l2d_last_error()


def test_weight_packers_cpu():
    from live2diff_amd import ops
    w = torch.randn(8, 5, 3, 3)
    p = ops.pack_conv3x3(w)
    assert p.shape == (8, 9 * 64) and p.dtype == torch.float16
    assert torch.equal(p.view(8, 9, 64)[:, 4, :5], w[:, :, 1, 1].half()) and (p.view(8, 9, 64)[:, :, 5:] == 0).all()
    perm = ops.geglu_perm(64, "cpu")
    assert perm[:16].tolist() == list(range(16)) and perm[16:32].tolist() == list(range(64, 80)) and perm[32] == 16
    assert sorted(perm.tolist()) == list(range(128))


@pytest.fixture
def dry_run():
    from live2diff_amd import _lib
    _lib.set_dry_run(True)
    yield
    _lib.set_dry_run(False)


@pytest.mark.parametrize("mode", ["stream", "warmup"])
def test_plan_builds_and_validates_without_gpu(dry_run, mode):
    """Build the whole UNet plan on CPU tensors and push every op through the C library's argument validation
    (validate-only mode): catches shape / stride / packing mistakes of the host code without a device."""
    from live2diff_amd.config import tiny_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 256, 256), cross_attention_dim=96)
    sd = random_state_dict(cfg, dtype=torch.float16)
    unet = HipStreamingUNet(sd, cfg, 16, 24, 2, device="cpu")
    kv = unet.prepare_cache(2)
    assert len(kv) == 40 and kv[0].shape == (2, 2, 16 * 24, 16, 64) and kv[16].shape == (2, 2, 2 * 3, 16, 256)
    st = unet._plan(mode, kv)
    st.pl.run(stream=0)                      # validate-only
    summ = unet.plan_summary(mode)
    from live2diff_amd import _lib
    assert summ["kinds"][_lib.OP_TATTN_STREAM if mode == "stream" else _lib.OP_TATTN_WARMUP] == 40
    # every GroupNorm is either its own (apply) launch or the prologue of the row GEMM behind it (rowgemm.hip, prologue 2);
    # no LayerNorm launch is left where the row GEMM applies (levels whose widths / token counts fit it)
    n_gn_pro = sum(1 for op in st.pl._ops if op.kind == _lib.OP_ROWGEMM and op.i[7] == 2)
    assert summ["kinds"][_lib.OP_FLASH_ATTN] == 32 and summ["kinds"][_lib.OP_GN_APPLY] + n_gn_pro == 22 * 2 + 16 + 20 + 1
    assert n_gn_pro > 0 and summ["kinds"].get(_lib.OP_ROWGEMM, 0) > 100
    # a broken op is refused with a message naming it
    bad = st.pl[5]
    old = bad.i[15]
    bad.i[15] = 3                            # ldo not a multiple of 4
    st.pl._arr = None
    with pytest.raises(_lib.L2DError):
        st.pl.run(stream=0)
    bad.i[15] = old


def test_plan_validates_sd15_shapes(dry_run):
    """SD-1.5 widths at 512x512 / N=2 / L=16 (cfg-2): the whole 1.28 B-parameter plan is built on CPU
    (zero weights) and every op passes the C library's validation."""
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import unet_param_spec
    cfg = sd15_config()
    sd = {k: torch.zeros(shp, dtype=torch.float16) for k, shp in unet_param_spec(cfg).items()}
    from live2diff_amd import _lib
    unet = HipStreamingUNet(sd, cfg, 64, 64, 2, device="cpu")
    del sd
    kv = unet.prepare_cache(2)
    st = unet._plan("stream", kv)
    st.pl.run(stream=0)
    assert 400 < st.n_ops <= 500             # round 2: 643 (LayerNorm / GroupNorm launches folded into the row GEMMs, q | k | V^T fused)
    kinds = unet.plan_summary()["kinds"]
    assert _lib.OP_LAYERNORM not in kinds and kinds[_lib.OP_GN_APPLY] == 22 * 2 + 1      # resnet norms + conv_norm_out only
    assert abs(unet.weight_bytes() / 1e9 - 2.6) < 0.2


@pytest.mark.parametrize("name,h,w,N,L,S", [("cfg-1", 32, 32, 1, 12, 4), ("cfg-3", 64, 96, 2, 24, 8), ("cfg-4", 64, 64, 4, 16, 8),
                                             ("cfg-5", 72, 128, 2, 40, 8),
                                             # tall / narrow latents: level 3 is 8 x 4 / 16 x 4 pixels -- narrower than the 8 pixels the
                                             # weight-streaming conv loader walks per DMA instruction (round-4 advisor finding: those
                                             # plans failed to build with EINVAL); their 3x3 convs keep the round-3 kernels
                                             ("tall 512x256", 64, 32, 2, 16, 8), ("wide 256x1024", 32, 128, 2, 16, 8)])
def test_plan_validates_other_baseline_configs(dry_run, name, h, w, N, L, S):
    """The other BASELINE.json configurations at SD-1.5 widths: stream and warm-up plans build from the heuristic schedule
    (their shapes are not in the cfg-2 table: deep split-K at the 4x4 / 8x8 levels, long windows on the chunked temporal kernel,
    ragged token counts) and every op passes the C library's validation; split launches beyond 16 splits keep the separate
    reduction launch (igemm.hip: one block per tile would serialise the fused form)."""
    from live2diff_amd import _lib, ops
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import unet_param_spec
    cfg = sd15_config(window_size=L, sink_size=S)
    sd = {k: torch.zeros(shp, dtype=torch.float16) for k, shp in unet_param_spec(cfg).items()}
    unet = HipStreamingUNet(sd, cfg, h, w, N, device="cpu")
    del sd
    kv = unet.prepare_cache(N)
    for mode in ("stream", "warmup"):
        st = unet._plan(mode, kv)
        st.pl.run(stream=0)
        st.cond_pl.run(stream=0)
        splits = [(op.i[21], bool(op.p[11])) for op in st.pl._ops if op.kind == _lib.OP_IGEMM and op.i[21] > 1]
        assert all(fused == (S_ <= ops.SPLITK_FUSED_MAX) for S_, fused in splits), (name, mode, splits)
        assert st.sk_used <= st.sk_cnt.numel()
    assert unet.plan_summary("stream")["gn_fused"] >= 40        # (levels whose tokens per sample are no multiple of the tile keep the statistics kernel)


def test_device_step_plan_validates_without_gpu(dry_run):
    """HipStreamStep (UNet plan + randn + stream_shift + ring_update as one op list) built on CPU
    tensors; every op passes the C library's validation, and malformed glue ops are refused."""
    from live2diff_amd import _lib, ops
    from live2diff_amd.config import tiny_config
    from live2diff_amd.stream_step_hip import HipStreamStep
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    N = 3
    unet = HipStreamingUNet(random_state_dict(cfg, dtype=torch.float16), cfg, 16, 16, N, device="cpu")
    kv = unet.prepare_cache(N)
    one = torch.ones(N, 1, 1, 1, 1, dtype=torch.float16)
    step = HipStreamStep(unet, kv, torch.tensor([399, 299, 199]), torch.zeros(N, 77, 64, dtype=torch.float16), one, one, one, one)
    step.pl.run(stream=0)                    # validate-only
    kinds = [step.pl[j].kind for j in range(len(step.pl))]
    assert kinds[-3:] == [_lib.OP_RANDN, _lib.OP_STREAM_SHIFT, _lib.OP_RING_UPDATE]
    assert len(step.pl) == len(step.st.pl) + 3

    def validate(opk):
        pl = _lib.OpList()
        pl.append(*opk)
        pl.run(stream=0)

    x = torch.zeros(9, 64, dtype=torch.float16)
    with pytest.raises(_lib.L2DError):       # more rows than the shift register supports
        validate(ops.stream_shift(x, x, torch.zeros(9, 4), x[0], N=9, per=64))
    b, p, u = torch.zeros(2, 16, dtype=torch.float16), torch.zeros(2, 16, dtype=torch.int64), torch.zeros(2, dtype=torch.int64)
    with pytest.raises(_lib.L2DError):       # sink must lie inside the window
        validate(ops.ring_update(b, p, u, N=2, L=16, sink=16))
    validate(ops.ring_update(b, p, u, N=2, L=16, sink=8))


def test_packed_weight_cache_round_trip(dry_run, tmp_path):
    """save_packed / load from path: identical packed tensors and layout tables, the plan built from
    the cache file validates, and a file packed for another window or format is refused."""
    from live2diff_amd.config import tiny_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    a = HipStreamingUNet(random_state_dict(cfg, dtype=torch.float16), cfg, 16, 16, 2, device="cpu")
    path = tmp_path / (HipStreamingUNet.packed_cache_name("sd15", "lcm", cfg.window_size, {"loras/style.safetensors": 0.8}, 16, 16, 2) + ".safetensors")
    assert path.name == f"sd15--lcm--style-0.8--16x16x2--L{cfg.window_size}--l2dpack{HipStreamingUNet.PACK_FORMAT}.safetensors"
    a.save_packed(path)
    b = HipStreamingUNet(path, cfg, 16, 16, 2, device="cpu")
    assert set(a.W) == set(b.W) and all(torch.equal(a.W[k], b.W[k]) and a.W[k].dtype == b.W[k].dtype for k in a.W)
    assert a.temb_offsets == b.temb_offsets and a.text_offsets == b.text_offsets
    assert (a.temb_total, a.text_total, a.text_kp, a.n_map_blocks) == (b.temb_total, b.text_total, b.text_kp, b.n_map_blocks)
    b._plan("stream", b.prepare_cache(2)).pl.run(stream=0)             # validate-only
    with pytest.raises(ValueError):
        HipStreamingUNet(path, tiny_config(window_size=24, channels=(64, 128, 128, 128), cross_attention_dim=64), 16, 16, 2, device="cpu")
    # which kernel serves a layer depends on the stream shape (levels with few tokens: weight-streaming forms; samples that are not
    # whole 32-token tiles: implicit-GEMM forms): a file packed for another layout is refused when it is LOADED, by name
    with pytest.raises(ValueError, match="re-pack"):
        HipStreamingUNet(path, cfg, 16, 32, 3, device="cpu")
    with pytest.raises(ValueError, match="re-pack"):
        HipStreamingUNet(path, cfg, 8, 24, 2, device="cpu")
    # ... and a shape change that leaves the layout alone reuses the file (the same levels take the same kernels)
    big = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    a2 = HipStreamingUNet(random_state_dict(big, dtype=torch.float16), big, 64, 64, 2, device="cpu")
    p2 = tmp_path / "big.safetensors"
    a2.save_packed(p2)
    if a2._pack_layout() == HipStreamingUNet(a2, big, 64, 64, 2, device="cpu")._pack_layout():
        HipStreamingUNet(p2, big, 64, 64, 2, device="cpu")


def test_plan_algorithmic_work_matches_survey(dry_run):
    """The roofline accounting of bench.py (`op_work`, per plan op) summed over the cfg-2 plan reproduces the work the
    survey derived from the REFERENCE's layer list: 2.227 TFLOP per UNet forward, 3.04 GB of KV cache
    streamed once, 40 temporal / 32 spatial attention launches."""
    import importlib.util
    import os

    from live2diff_amd import _lib
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import unet_param_spec
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = sd15_config()
    sd = {k: torch.zeros(shp, dtype=torch.float16) for k, shp in unet_param_spec(cfg).items()}
    unet = HipStreamingUNet(sd, cfg, 64, 64, 2, device="cpu")
    del sd
    kv = unet.prepare_cache(2)
    st = unet._plan("stream", kv)
    tot = {}
    s_ = unet.plan_summary()
    assert s_["gn_fused"] == 81 and s_["gn_stats_launches"] == 0      # every GroupNorm's statistics come from its producers
    assert len(st.cond_pl) == 6        # timestep sinusoid + 3 skinny GEMMs + text K / V^T: run when the conditioning changes
    for op in [st.cond_pl[j] for j in range(len(st.cond_pl))] + [st.pl[j] for j in range(len(st.pl))]:
        fl, by = bench.op_work(op, _lib)
        t = tot.setdefault(op.kind, [0, 0.0, 0.0])
        t[0] += 1; t[1] += fl; t[2] += by
    flops = sum(t[1] for t in tot.values())
    assert abs(flops / 2.227e12 - 1) < 5e-3, flops                     # 2.227 TFLOP at cfg-2
    kv_bytes = sum(c.numel() * 2 for c in kv)
    assert abs(kv_bytes / 3.04e9 - 1) < 2e-3                           # 3.04 GB
    n, fl, by = tot[_lib.OP_TATTN_STREAM]
    assert n == 40 and abs(by - (kv_bytes + 2 * kv_bytes / cfg.window_size)) < 1e-6 * by   # K+V once + (row write, q, out) = 8 N T C bytes
    # 380 GEMM launches in round 2; the 16 V^T projections now ride in the q | k | V^T row GEMMs
    # (21 level-0 / level-1 3x3 convs: patch kernel; the levels with <= 512 stream tokens: weight-streaming GEMM where the in-frame
    #  tuner found it faster, round 4)
    gemm = [tot[k_] for k_ in (_lib.OP_IGEMM, _lib.OP_ROWGEMM, _lib.OP_PCONV, _lib.OP_WSGEMM, _lib.OP_ROWCHAIN, _lib.OP_CCONV) if k_ in tot]
    # (round 5, rowchain.hip: the tail of each of the 10 level-0 transformer blocks -- to_out + residual, LayerNorm + GEGLU, FF2 +
    #  residual, proj_out + residual -- is ONE token-resident launch instead of four)
    #  ... and the two head segments of each block (proj_in behind the GroupNorm + q | k | v; to_out + residual + the next query /
    #  q | k | v) one launch each instead of two
    assert tot[_lib.OP_ROWCHAIN][0] == 10 + 20
    # (round 5: the level-1 3x3 convs whose contraction is long -- 9 of the 10 -- and the level-1 q | k | V^T / GEGLU layers moved to the
    #  weight-streaming kernel too, per measured shape: the `large` list of wsgemm_tuned.json)
    assert tot[_lib.OP_FLASH_ATTN][0] == 32 and sum(g_[0] for g_ in gemm) == 380 - 16 - 30 - 20 and tot.get(_lib.OP_PCONV, [0])[0] == 11       # (the 320 -> 640 conv of level 1: cconv since round 6)
    # (round 6, cconv.hip: the 3x3 convs of the 640- / 1280-wide levels with whole 8 x 16 patches -- levels 1 and 2 at cfg-2 -- and the
    #  up-samplers that produce such a level moved from the weight-streaming / implicit-GEMM kernels to the patch-resident,
    #  register-streamed form)
    assert tot.get(_lib.OP_CCONV, [0])[0] == 23 and tot.get(_lib.OP_WSGEMM, [0])[0] + tot.get(_lib.OP_CCONV, [0])[0] >= 100
    assert abs(sum(g_[1] for g_ in gemm) / 1.9723e12 - 1) < 1e-3          # the figure quoted in DESIGN.md section 3
    assert _lib.OP_LAYERNORM not in tot


def test_plain_k1280_linear_stays_on_igemm_where_the_chain_does_not_run(dry_run):
    """Round-5 advisor finding: the block chain's FF2 packing must not move the plain K = 4 C Linear to the row GEMM at a C = 320 level
    whose M / 32 blocks do not fill the chip (32 x 40 latent, N = 2: M = 2560, 80 blocks < ROWCHAIN_MIN_BLOCKS = 96; round 6 moved the
    threshold from 192 to 96 after measuring 154, 144 and 100 blocks) -- the measured rule keeps it on igemm."""
    from live2diff_amd import _lib
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import unet_param_spec
    cfg = sd15_config()
    sd = {k: torch.zeros(shp, dtype=torch.float16) for k, shp in unet_param_spec(cfg).items()}
    unet = HipStreamingUNet(sd, cfg, 32, 40, 2, device="cpu")
    del sd
    st = unet._plan("stream", unet.prepare_cache(2))
    ops_ = [st.pl[j] for j in range(len(st.pl))]
    assert not any(o.kind == _lib.OP_ROWCHAIN and o.i[6] == 0 for o in ops_)                   # no chain tail at this size
    assert not any(o.kind == _lib.OP_ROWGEMM and o.i[1] == 1280 and o.i[7] == 0 and o.i[0] == 2560 for o in ops_)
    assert sum(1 for o in ops_ if o.kind == _lib.OP_IGEMM and o.i[0] == 1 and o.i[13] == 2560 and o.i[1] == 1280 and o.i[14] == 320) == 10


def test_one_launch_groupnorm_rule_agrees_with_the_library(dry_run):
    """ops.gn_self_ok (which GroupNorms the plan gives to the one-launch form, norm.hip gn_self_kernel) must be exactly the set the C
    library accepts for gn_apply with nchunk = 0 and no accumulator: checked through the library's validate-only mode over the SD widths,
    their concats and pixel counts of tuned and untuned levels."""
    from live2diff_amd import _lib, ops
    n_ok = 0
    for C in (64, 160, 320, 640, 960, 1280, 1920, 2560):
        for T in (16, 36, 64, 100, 144, 256, 400, 576, 1024, 4096):
            x = torch.zeros(1, T, C, dtype=torch.float16)
            g = torch.ones(C, dtype=torch.float16)
            op = ops.gn_apply(x, None, g, g, torch.empty_like(x), eps=1e-5, silu=True, B=1, T=T, C1=C, ld1=C, G=32, nchunk=0)
            ok = True
            try:
                ops.run(op)
            except _lib.L2DError as e:
                ok = False
                assert "one-launch form" in str(e)
            assert ok == ops.gn_self_ok(T, C, 32), (C, T, ok)
            n_ok += ok
    assert 20 < n_ok < 70


def test_untuned_resolution_uses_the_fallback_rules(dry_run):
    """384 x 384 (48 x 48 latent, N = 2): no table holds its shapes.  The plan must take the chain kernel at 144 blocks, the one-launch
    GroupNorm at the 12 x 12 / 6 x 6 levels (no gn_stats launch left but the ones that do not fit), and stay under 480 launches
    (566 before round 6's fallback rules)."""
    from live2diff_amd import _lib
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import unet_param_spec
    cfg = sd15_config()
    sd = {k: torch.zeros(shp, dtype=torch.float16) for k, shp in unet_param_spec(cfg).items()}
    unet = HipStreamingUNet(sd, cfg, 48, 48, 2, device="cpu")
    del sd
    st = unet._plan("stream", unet.prepare_cache(2))
    st.pl.run(stream=0)
    summ = unet.plan_summary("stream")
    kinds = summ["kinds"]
    assert kinds.get(_lib.OP_ROWCHAIN, 0) == 30 and summ["gn_self_launches"] >= 38 and kinds.get(_lib.OP_GN_STATS, 0) <= 2
    assert st.n_ops <= 480, st.n_ops


def test_igemm_fallback_rule_follows_the_tuned_table():
    """Round 6: the rule behind igemm_tuned.json (`ops._igemm_heuristic_r6`, what every shape outside cfg-2 runs on) is fitted to the table's
    in-frame picks and must keep reproducing them -- (tile, split-K) of at least 50 of the 60 shapes (the round-1 rule: 30), every
    few-token Linear / 8 x 8 conv among them, and always a schedule the launcher accepts (fused split-K <= 16, >= 2 BK64 steps per split)."""
    import json

    from live2diff_amd import ops
    shapes = json.load(open(os.path.join(os.path.dirname(ops.__file__), "igemm_tuned.json")))["shapes"]
    hit_new = hit_old = 0
    for key, (tile, S, _v) in shapes.items():
        taps, M, N, Kp, epi, batch = (int(v) for v in key.split(","))
        t6, s6, v6 = ops._igemm_heuristic_r6(M, N, Kp, batch, epi)
        t1, s1, _ = ops._igemm_heuristic_r1(M, N, Kp, batch, epi)
        assert t6 in (1, 2) and 1 <= s6 <= ops.SPLITK_FUSED_MAX and Kp // 64 >= 2 * s6 and 0 <= v6 <= 10, key
        hit_new += (t6, s6) == (tile, S)
        hit_old += (t1, s1) == (tile, S)
        if M <= 128 and N == 1280:
            assert (t6, s6) == (tile, S), key                    # level 3 + mid of cfg-2: the launches the old rule lost most on
    assert hit_new >= 50 and hit_old <= 32, (hit_new, hit_old)
    # shapes no table holds: few tokens -> 64 x 64 tiles, K split towards ~240-480 blocks
    assert ops._igemm_heuristic_r6(288, 1280, 1280, 1, 0)[:2] == (2, 2) and ops._igemm_heuristic_r6(72, 1280, 11520, 1, 0)[:2] == (2, 6)


def test_pipeline_mirror_keeps_the_reference_api_surface():
    """names the Python counterpart of `StreamAnimateDiffusionDepth` must preserve (reference
    pipeline_stream_animation_depth.py:24-666), checked on a CPU instance driven by a stand-in UNet callable -- the
    boundary is duck-typed, exactly like the reference's TensorRT engine swap (wrapper.py:613-626)."""
    import inspect
    from types import SimpleNamespace

    import live2diff_amd.pipeline_stream_animation_depth as M
    S = M.StreamAnimateDiffusionDepth
    assert M.WARMUP_FRAMES == 8 and M.WINDOW_SIZE == 16
    for name in ("prepare_cache", "get_timesteps", "load_lora", "fuse_lora", "enable_similar_image_filter",
                 "disable_similar_image_filter", "prepare", "warmup_engine", "update_prompt", "add_noise",
                 "scheduler_step_batch", "initialize_attn_bias_pe_and_update_idx", "update_attn_bias", "unet_step",
                 "encode_image", "decode_image", "encode_depth", "predict_x0_batch", "__call__", "load_warmup_unet"):
        assert callable(getattr(S, name, None)), name
    params = inspect.signature(S.__init__).parameters
    for kw in ("pipe", "num_inference_steps", "t_index_list", "strength", "torch_dtype", "width", "height", "do_add_noise",
               "use_denoising_batch", "frame_buffer_size", "clip_skip", "cfg_type"):
        assert kw in params, kw                                        # reference __init__ kwargs (:25-39)

    class FakeUNet:                                                    # the duck-typed boundary object
        config = SimpleNamespace(in_channels=4)

        def prepare_cache(self, n):
            return [torch.zeros(n, 2, 16, 16, 64, dtype=torch.float16)]

        def __call__(self, sample, timestep, **kw):
            assert set(kw) >= {"encoder_hidden_states", "temporal_attention_mask", "depth_sample", "kv_cache", "pe_idx", "update_idx"}
            return {"sample": sample * 0.5, "kv_cache": kw["kv_cache"]}

    pipe = SimpleNamespace(device=torch.device("cpu"), vae_scale_factor=8, unet=FakeUNet(), scheduler=None)
    s = S(pipe, num_inference_steps=50, t_index_list=[30, 40], width=32, height=32, torch_dtype=torch.float32)
    for attr in ("batch_size", "trt_unet_batch_size", "t_list", "denoising_steps_num", "frame_bff_size", "inference_time_ema",
                 "inference_time_list", "depth_time_ema"):
        assert hasattr(s, attr), attr
    assert s.batch_size == 2 and s.denoising_steps_num == 2 and s.t_list == [30, 40]
    s.prepare_cache(32, 32, 2)
    assert isinstance(s.kv_cache_list, list)
    s.attn_bias, s.pe_idx, s.update_idx = s.initialize_attn_bias_pe_and_update_idx()
    assert s.attn_bias.shape == (2, 16) and s.pe_idx.shape == (2, 16) and s.update_idx.tolist() == [8, 9]
    # one stream-batch step through the duck-typed boundary with the reference's scheduler algebra
    s.sub_timesteps_tensor = torch.tensor([399, 199])
    s.prompt_embeds = torch.zeros(2, 77, 8)
    shp = (2, 1, 1, 1, 1)
    s.alpha_prod_t_sqrt, s.beta_prod_t_sqrt = torch.full(shp, 0.8), torch.full(shp, 0.6)
    s.c_skip, s.c_out = torch.full(shp, 0.1), torch.full(shp, 0.9)
    s.x_t_latent_buffer = torch.ones(1, 4, 1, 4, 4)
    s.depth_latent_buffer = torch.zeros(1, 4, 1, 4, 4)
    out = s.predict_x0_batch(torch.ones(1, 4, 1, 4, 4), torch.zeros(1, 4, 1, 4, 4), noise=torch.zeros(1, 4, 1, 4, 4))
    x0 = 0.9 * (1 - 0.6 * 0.5) / 0.8 + 0.1                            # c_out (x - beta eps) / alpha + c_skip x, x = 1, eps = 0.5
    assert out.shape == (1, 4, 1, 4, 4) and torch.allclose(out, torch.full_like(out, x0), atol=1e-6)
    assert torch.allclose(s.x_t_latent_buffer, torch.full((1, 4, 1, 4, 4), 0.8 * x0), atol=1e-6)
    assert s.update_idx.tolist() == [9, 8]                             # frame 1 of the reference trace (row 1 lags row 0 by a slot)


@pytest.mark.parametrize("n,t_index", [(2, [30, 40]), (3, [20, 30, 45])])
def test_prepare_and_call_chain_match_reference_capture(golden, monkeypatch, n, t_index):
    """A12 + the per-frame glue, pinned to the REFERENCE: tests/golden/pipeline_chain.npz holds prepare() (warm-up passes with the
    x0 -> re-noise chain between them, reference :171-344) and six `__call__`s (:625-666) of the reference's
    StreamAnimateDiffusionDepth run on deterministic mock models (tests/pipeline_mocks.py).  The mirror, driven with the same
    mocks, seeds and inputs, must reproduce outputs, latent / depth shift registers, ring-buffer state, every cache and the
    sequence of UNet arguments -- which also pins the order and shapes of the random draws (global RNG + `generator`)."""
    import pipeline_mocks as M
    from live2diff_amd import pipeline_stream_animation_depth as P
    g = golden("pipeline_chain")
    k = f"n{n}_"
    monkeypatch.setattr(torch.cuda, "Event", M.NoCudaEvent)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **kw: None)
    monkeypatch.setattr(P, "retrieve_latents", M.retrieve_latents)
    pipe = M.MockPipe()
    pipe.unet, pipe.vae, pipe.depth_model = M.MockStreamUNet(), M.MockVAE(), M.MockDepth()
    s = P.StreamAnimateDiffusionDepth(pipe, num_inference_steps=50, t_index_list=t_index, torch_dtype=torch.float32, width=M.W, height=M.H)
    s.scheduler = M.MockScheduler()
    s.timesteps = s.scheduler.timesteps
    s.image_processor = M.MockImageProcessor()
    s.unet_warmup = M.MockWarmupUNet()
    s.kv_cache_list = M.make_caches(n)
    torch.manual_seed(123)
    warm = s.prepare(M.frames(8, seed=7), "a prompt", seed=2)
    T = lambda name: torch.from_numpy(g[k + name])
    close = lambda a, b: torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6)
    assert torch.equal(s.sub_timesteps_tensor, T("sub_timesteps"))
    for mine, name in ((s.c_skip, "c_skip"), (s.c_out, "c_out"), (s.alpha_prod_t_sqrt, "alpha"), (s.beta_prod_t_sqrt, "beta")):
        assert close(mine, T(name)), name
    assert torch.equal(s.init_noise, T("init_noise"))                     # same draw from the same RNG stream
    assert close(warm, T("prepare_out"))
    assert close(torch.stack(s.kv_cache_list), T("prepare_caches"))
    for i, img in enumerate(M.frames(6, seed=11)):
        out = s(img)
        assert close(out, T("frame_out")[i]), f"frame {i}"
        assert close(s.x_t_latent_buffer, T("x_t_buffer")[i]) and close(s.depth_latent_buffer, T("depth_buffer")[i]), f"buffers {i}"
    assert close(torch.stack(s.kv_cache_list), T("caches"))
    assert torch.equal(torch.isinf(s.attn_bias), torch.isinf(T("bias"))) and torch.equal(s.pe_idx, T("pe_idx"))
    assert torch.equal(s.update_idx, T("update_idx"))
    assert torch.equal(torch.stack([c["t"] for c in pipe.unet.log]), T("unet_t"))
    assert torch.equal(torch.stack([c["update_idx"] for c in pipe.unet.log]), T("unet_update_idx"))


def test_float_assisted_division_is_exact():
    """igemm.hip `l2d_divf`: q = int(float(n) * (1.0f / d)), one correction step either way -- the kernels' tile / pixel index
    arithmetic relies on it being EXACT for 0 <= n < 2^24 (validated range).  Emulated here in IEEE fp32 (numpy), on the values the
    plans use (token counts, tile counts, image widths, split counts) and on random / adversarial operands."""
    rng = np.random.default_rng(0)

    def divf(n, d):
        inv = (np.float32(1.0) / d.astype(np.float32)).astype(np.float32)
        q = (n.astype(np.float32) * inv).astype(np.int64)                 # (int) truncation of a non-negative float
        r = n - q * d
        return q + (r >= d).astype(np.int64) - (r < 0).astype(np.int64)

    n = rng.integers(0, 1 << 24, size=2_000_000, dtype=np.int64)
    d = rng.integers(1, 1 << 20, size=n.size, dtype=np.int64)
    assert np.array_equal(divf(n, d), n // d)
    # adversarial: n = k*d - 1, k*d, k*d + 1 around every multiple, small and large divisors, and the largest numerators
    for dd in (1, 2, 3, 5, 7, 9, 10, 12, 24, 63, 64, 65, 96, 127, 128, 320, 576, 577, 1024, 4095, 4096, 4097, 9216, 65535, 65537, 1 << 20):
        k = np.arange(0, min((1 << 24) // dd, 200_000) + 1, dtype=np.int64)
        for off in (-1, 0, 1):
            nn = k * dd + off
            nn = nn[(nn >= 0) & (nn < (1 << 24))]
            assert np.array_equal(divf(nn, np.full_like(nn, dd)), nn // dd), dd
        top = np.arange((1 << 24) - 100_000, 1 << 24, dtype=np.int64)
        assert np.array_equal(divf(top, np.full_like(top, dd)), top // dd), dd


def test_frame_filter_matches_reference_trace():
    """`frame_filter.SimilarImageFilter` (the default behind enable_similar_image_filter) against pass / drop decisions
    captured from the reference's own class (tests/golden/gen_golden_filter.py) on the same seeded frames and draws."""
    import json
    import random

    from live2diff_amd.frame_filter import SimilarImageFilter
    sys.path.insert(0, GOLDEN)
    from gen_golden_filter import frame_sequence
    with open(os.path.join(GOLDEN, "frame_filter.json")) as fh:
        want = json.load(fh)
    assert any(0 < sum(v) < len(v) for v in want.values())          # the trace exercises both outcomes
    for key, dec in want.items():
        thr, ms = key.split(",")
        random.seed(4321)
        f = SimilarImageFilter()
        f.set_threshold(float(thr)); f.set_max_skip_frame(float(ms))
        got = [0 if f(x) is None else 1 for x in frame_sequence()]
        assert got == dec, key


def test_unet_instances_can_share_packed_weights(dry_run):
    """A second HipStreamingUNet built FROM a first one (serving several streams per GPU) shares its packed weights by reference
    and refuses another latent size / configuration (the packing depends on both)."""
    from live2diff_amd.config import tiny_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config()
    u = HipStreamingUNet(random_state_dict(cfg, dtype=torch.float16), cfg, 16, 16, 2, device="cpu")
    v = HipStreamingUNet(u, cfg, 16, 16, 2, device="cpu", use_graph=True)
    assert v.W is u.W and v.temb_offsets is u.temb_offsets
    n = len(v._plan("stream", v.prepare_cache(2)).pl)
    assert n == len(u._plan("stream", u.prepare_cache(2)).pl)
    with pytest.raises(ValueError):
        HipStreamingUNet(u, cfg, 8, 8, 2, device="cpu")
    with pytest.raises(ValueError):                       # the packing depends on the stream batch too (weight-streaming levels, skip list)
        HipStreamingUNet(u, cfg, 16, 16, 8, device="cpu")
    with pytest.raises(ValueError):
        HipStreamingUNet(u, tiny_config(window_size=12, sink_size=4), 16, 16, 2, device="cpu")


def test_wsgemm_packers_schedule_and_validation(dry_run):
    """Host side of the weight-streaming GEMM (csrc/wsgemm.hip) without a GPU: the fragment packing is a permutation that
    rowgemm's inverse undoes, the LayerNorm fold's column sums are the row sums of the ROUNDED folded weights, the conv packing
    orders k as (tap, channel), every default / tuned schedule passes the library's argument validation for the frame's shapes,
    and malformed launches are refused."""
    from live2diff_amd import _lib, ops
    g = torch.Generator().manual_seed(5)
    w = torch.randn(96, 128, generator=g)
    b, gm, bt = torch.randn(96, generator=g), 1 + 0.1 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
    wp, bp, cs = ops.pack_wsgemm(w, b, gm.half(), bt.half())
    wf = (w * gm.half().float()[None]).half()
    assert torch.equal(ops.unpack_rowgemm(wp, 96, 128), wf)                          # same fragment order as rowgemm
    assert torch.allclose(cs, wf.float().sum(1)) and torch.allclose(bp, b + w @ bt.half().float(), atol=1e-5)
    assert ops.pack_wsgemm(w, b)[2] is None
    cw = torch.randn(64, 70, 3, 3, generator=g)
    cp = ops.pack_wsgemm_conv3x3(cw)
    flat = ops.unpack_rowgemm(cp, 64, 9 * 128).view(64, 9, 128)
    assert torch.equal(flat[:, :, :70], cw.permute(0, 2, 3, 1).reshape(64, 9, 70).half()) and (flat[:, :, 70:] == 0).all()

    def validate(opk):
        pl = _lib.OpList()
        pl.append(*opk)
        pl.run(stream=0)

    cnt = torch.zeros(4096, dtype=torch.int32)
    for (taps, M, C1, C2, N, epi, pro, ntr) in ((1, 512, 1280, 0, 1280, 0, 0, 0), (1, 128, 1280, 0, 10240, 1, 1, 0), (1, 512, 1280, 0, 3840, 0, 1, 1280),
                                                 (1, 768, 5120, 0, 1280, 0, 0, 0), (9, 128, 1280, 1280, 1280, 0, 0, 0), (9, 1152, 640, 0, 1280, 0, 0, 0),
                                                 (1, 192, 1280, 1280, 1280, 0, 0, 0), (1, 300, 64, 0, 96, 0, 0, 0)):
        K = C1 + C2
        sched = ops.wsgemm_schedule(M, taps * K, N, ntr, epi, pro, taps)
        NW, NT, NL, S, ntw = sched
        assert (N // 32) % (NW * NT) == 0 and (ntr // 32) % (NW * NT) == 0 and 1 <= S <= taps * K // 64
        x1, x2 = torch.zeros(M, C1, dtype=torch.float16), (torch.zeros(M, C2, dtype=torch.float16) if C2 else None)
        wt = torch.zeros(N * taps * K, dtype=torch.float16)
        No = N // 2 if epi else N
        kw = {}
        if S > 1:
            n_ws, n_cnt = ops.wsgemm_sizes(M, N, NW, NT, S)
            kw = dict(ws=torch.zeros(n_ws), cnt=cnt)
        T = M // 2 if ntr == 0 else 256
        HW = {128: (8, 8), 1152: (18, 32)}.get(M, (1, 1))
        out_t = torch.zeros(M // T, ntr, T + 8, dtype=torch.float16) if ntr else None
        validate(ops.wsgemm(x1, wt, torch.zeros(M, (N - ntr) // (2 if epi else 1) or No, dtype=torch.float16), M=M, Nout=N, C1=C1, ldx1=C1, x2=x2, C2=C2,
                            ldx2=C2, ldo=(N - ntr) // (2 if epi else 1), bias=torch.zeros(N), colsum=(torch.zeros(N) if pro else None), taps=taps,
                            B=2, H=HW[0], W=HW[1], epi=epi, pro=pro, T=T, out_t=out_t, ntr=ntr, ldt=T + 8, st=ntr * (T + 8), sched=sched, **kw))
    x, wt, out = torch.zeros(128, 64, dtype=torch.float16), torch.zeros(64 * 64, dtype=torch.float16), torch.zeros(128, 64, dtype=torch.float16)
    with pytest.raises(_lib.L2DError):
        validate(ops.wsgemm(x, wt, out, M=128, Nout=64, C1=64, ldx1=64, ldo=64, sched=(3, 1, 1, 1, False)))     # 2 tiles over 3 waves
    with pytest.raises(_lib.L2DError):
        op, keep = ops.wsgemm(x, wt, out, M=128, Nout=64, C1=64, ldx1=64, ldo=64, sched=(2, 1, 1, 1, False))
        op.i[12] = 2                                                                                            # K slices without a workspace
        validate((op, keep))
    with pytest.raises(_lib.L2DError):
        op, keep = ops.wsgemm(x, wt, out, M=128, Nout=64, C1=64, ldx1=64, ldo=64, sched=(2, 1, 1, 1, False))
        op.i[20] = 1                                                                                            # LayerNorm fold without column sums
        validate((op, keep))


def test_wsgemm_table_lists_are_consistent_and_gate_the_packing():
    """wsgemm_tuned.json: `skip` (opt-out) holds few-token shapes only, `large` (opt-in, round 5) shapes above WS_SMALL_M only, every
    schedule is one the launcher accepts for its shape; ops.wsgemm_wanted follows the lists at every token count the tuner measured; at
    other token counts (round 6) the rule fitted to the lists decides: the 1280-wide few-token layers, LayerNorm + q|k|v / GEGLU at 640
    wide, long plain contractions from 3072 tokens on -- and nothing at the 320-wide level or beyond 4608 tokens."""
    import json

    from live2diff_amd import ops
    d = json.load(open(os.path.join(os.path.dirname(ops.__file__), "wsgemm_tuned.json")))
    M_of = lambda k: int(k.split(",")[1])
    assert d["large"] and all(M_of(k) > ops.WS_SMALL_M for k in d["large"]), "large = opt-in list of many-token shapes"
    assert all(M_of(k) <= ops.WS_SMALL_M for k in d["skip"]), "skip = opt-out list of few-token shapes"
    assert not set(d["large"]) & set(d["skip"])
    for key, (nw, nt, nl, S) in d["shapes"].items():
        taps, M, K, N, ntr, epi, pro = (int(v) for v in key.split(","))
        tiles = N // 32
        assert nt in (1, 2) and 1 <= nw <= (10 if nt == 1 else 4) and nl in (1, 2) and tiles % (nw * nt) == 0 and (ntr // 32) % (nw * nt) == 0, key
        assert 1 <= S <= max(1, K // 64) and not (ntr and S > 1), key
        assert ops.wsgemm_schedule(M, K, N, ntr, epi, pro, taps)[:4] == (nw, nt, nl, S), key
    for key in d["large"]:
        assert ops.wsgemm_wanted(*[int(v) for v in key.split(",")])
    for key in d["skip"]:
        assert not ops.wsgemm_wanted(*[int(v) for v in key.split(",")])
    assert ops.wsgemm_wanted(1, 640, 1280, 1280)              # untuned, few tokens: the weight-streaming kernel
    assert not ops.wsgemm_wanted(1, 2304, 1280, 1280)         # untuned, many tokens: the round-3 kernels
    assert not ops.wsgemm_wanted(9, 8192, 2880, 320)          # level 0 of cfg-2: measured, lost
    assert ops.wsgemm_wanted(1, 3200, 640, 1920, 0, 0, 1) and ops.wsgemm_wanted(1, 3200, 2560, 640)      # untuned level 1 (640 x 640 image)
    assert not ops.wsgemm_wanted(1, 3200, 1280, 320) and not ops.wsgemm_wanted(1, 1152, 640, 640)        # 320-wide level; C -> C at 640
    assert not ops.wsgemm_wanted(1, 2048, 2560, 640)          # a token count the tuner saw: the lists decide (igemm kept this one)


def test_cconv_patch_image_is_bank_conflict_free():
    """cconv.hip's LDS patch image: pixel p of the haloed 10 x 18 patch at p * 128 bytes, channel slot q (16 bytes) at position
    q ^ ((patch column >> 1) & 7).  An MFMA token tile is two patch rows of 16 pixels (lane l32 -> row l32 >> 4, column l32 & 15), lanes
    32..63 read the other k half (slot + 1).  gfx950 serves a ds_read_b128 in four 16-lane groups over 64 banks of 4 bytes
    (MI355X_MICROARCH.md, LDS table): every group must hit 16 distinct 16-byte bank slots, for every tap, k step and token tile."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups]
    for kk in range(4):
        for dy in range(3):
            for dx in range(3):
                for mt in range(4):
                    for g in groups:
                        slots = set()
                        for l in g:
                            l32, lh = l & 31, l >> 5
                            p = (mt * 2 + (l32 >> 4) + dy) * 18 + (l32 & 15) + dx
                            q = 2 * kk + lh
                            addr = p * 128 + ((q ^ (((p % 18) >> 1) & 7)) * 16)
                            slots.add((addr // 16) % 16)
                        assert len(slots) == 16, (kk, dy, dx, mt, g)
