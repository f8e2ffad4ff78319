"""The oracle (oracle/unet_ref.py, fp32 CPU) against golden vectors captured from the REFERENCE's own
classes (tests/golden/gen_golden.py).  Tolerance: fp32, max-abs <= 1e-4 relative to output scale
(SURVEY.md section 8c proposes 1e-5 absolute for O(1) tensors; outputs here reach |x|~30)."""
import json
import os

import numpy as np
import pytest
import torch

from live2diff_amd.config import tiny_config
from live2diff_amd.weights import _fill, unet_param_spec
from oracle import unet_ref as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=2e-5):
    a, b = a.double(), b.double()
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"max-abs {err:.3e} > {tol * scale:.3e}"


def sd_for(prefix, shapes):
    return {k: _fill(prefix + k, shp, 1.0) for k, shp in shapes.items()}


def test_pe_table(golden):
    g = golden("pe_table")
    close(O.sinusoid_pe(40, 64), T(g["pe"]), 1e-6)


def test_param_spec_matches_reference():
    with open(os.path.join(GOLDEN, "param_spec_tiny.json")) as f:
        ref = json.load(f)
    spec = unet_param_spec(tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64))
    assert set(ref) == set(spec)
    for k, shp in spec.items():
        assert list(shp) == ref[k], k


ATTN_SHAPES = lambda C: {"to_q.weight": (C, C), "to_k.weight": (C, C), "to_v.weight": (C, C),
                         "to_out.0.weight": (C, C), "to_out.0.bias": (C,)}


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_stream_temporal_attention(golden, ci):
    g = golden(f"stream_attn_{ci}")
    C, Tn, L, S, N = [int(v) for v in g["meta"]]
    cfg = tiny_config(window_size=L, sink_size=S)
    sd = sd_for(f"sta{ci}.", ATTN_SHAPES(C))
    cache = T(g["cache_in"]).clone()
    pe = O.sinusoid_pe(max(24, L), C)
    out = O.stream_temporal_attention(T(g["x"]), O._W(sd), cfg, cache, T(g["bias"]), T(g["pe_idx"]),
                                      T(g["update_idx"]), pe)
    close(out, T(g["out"]))
    close(cache, T(g["cache_out"]))
    # exactly one slot per row changed
    changed = (T(g["cache_in"]) != cache).any(dim=-1).any(dim=2).any(dim=1)   # [N,L]
    for n in range(N):
        assert changed[n].nonzero().flatten().tolist() == [int(g["update_idx"][n])]


def _mm_shapes(C):
    from live2diff_amd.weights import _motion
    spec = {}
    _motion(spec, "", C)
    return spec


def test_motion_module_stream(golden):
    g = golden("motion_module_stream")
    cfg = tiny_config()
    sd = sd_for("mm.", _mm_shapes(64))
    caches = [T(g["cache_in0"]).clone(), T(g["cache_in1"]).clone()]
    pe = O.sinusoid_pe(cfg.temporal_max_len, 64)
    x = T(g["x"])[:, :, 0]

    def attn(tokens, wa, idx):
        return O.stream_temporal_attention(tokens, wa, cfg, caches[idx], T(g["bias"]), T(g["pe_idx"]),
                                           T(g["update_idx"]), pe)

    out = O.motion_module(x, O._W(sd), cfg, attn, 0)
    close(out, T(g["out"])[:, :, 0])
    close(caches[0], T(g["cache_out0"]))
    close(caches[1], T(g["cache_out1"]))


def test_motion_module_warmup(golden):
    g = golden("motion_module_warmup")
    cfg = tiny_config()
    sd = sd_for("mm.", _mm_shapes(64))
    rows = [torch.zeros(2, 16, 16, 64), torch.zeros(2, 16, 16, 64)]
    pe = O.sinusoid_pe(cfg.temporal_max_len, 64)
    x = T(g["x"])[0].transpose(0, 1)   # [F,C,H,W]

    def attn(tokens, wa, idx):
        return O.warmup_temporal_attention(tokens, wa, cfg, rows[idx], pe)

    out = O.motion_module(x, O._W(sd), cfg, attn, 0)
    close(out, T(g["out"])[0].transpose(0, 1))
    close(rows[0], T(g["cache_out0"]))
    close(rows[1], T(g["cache_out1"]))
    assert rows[0][:, :, 8:].abs().max() == 0   # only the sink slots are written


@pytest.mark.parametrize("name,cin,cout", [("resnet_same", 64, 64), ("resnet_proj", 96, 64)])
def test_resnet(golden, name, cin, cout):
    from live2diff_amd.weights import _resnet
    g = golden(name)
    spec = {}
    _resnet(spec, "", cin, cout, 128)
    sd = sd_for(name + ".", spec)
    out = O.resnet_block(T(g["x"])[:, :, 0], T(g["temb"]), O._W(sd), tiny_config())
    close(out, T(g["out"])[:, :, 0])


def test_down_up_mapping(golden):
    import torch.nn.functional as F
    g = golden("downsample")
    sd = sd_for("down.", {"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)})
    close(O._conv(T(g["x"])[:, :, 0], O._W(sd), "conv", stride=2), T(g["out"])[:, :, 0])
    g = golden("upsample")
    sd = sd_for("up.", {"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)})
    x = F.interpolate(T(g["x"])[:, :, 0], scale_factor=2.0, mode="nearest")
    close(O._conv(x, O._W(sd), "conv"), T(g["out"])[:, :, 0])
    g = golden("mapping")
    shapes = {"conv_in.weight": (16, 4, 3, 3), "conv_in.bias": (16,), "conv_out.weight": (64, 256, 3, 3),
              "conv_out.bias": (64,)}
    mc = (16, 32, 96, 256)
    for i in range(3):
        shapes[f"blocks.{2 * i}.weight"] = (mc[i], mc[i], 3, 3)
        shapes[f"blocks.{2 * i}.bias"] = (mc[i],)
        shapes[f"blocks.{2 * i + 1}.weight"] = (mc[i + 1], mc[i], 3, 3)
        shapes[f"blocks.{2 * i + 1}.bias"] = (mc[i + 1],)
    sd = sd_for("map.", shapes)
    close(O.mapping_network(T(g["x"])[:, :, 0], O._W(sd)), T(g["out"])[:, :, 0])


def test_spatial_transformer_stub_pinned(golden):
    """diffusers-0.25.0 Attention/GEGLU semantics: stub-pinned (parity unpinned by the reference)."""
    from live2diff_amd.weights import _spatial
    g = golden("spatial_transformer")
    spec = {}
    _spatial(spec, "", 64, 96)
    sd = sd_for("sp.", spec)
    out = O.spatial_transformer(T(g["x"])[:, :, 0], T(g["enc"]), O._W(sd), tiny_config())
    close(out, T(g["out"])[:, :, 0])


def test_unet_rollout(golden):
    """Tiny full UNet: N warm-up passes (cache fill) then 12 streaming frames driven by the reference's
    ring-buffer trace; checks eps predictions and cache contents.  Tolerance 3e-4 of the output scale:
    fp32 on both sides, but at this test scale the deepest level is 1x1 pixels with 2 channels per
    GroupNorm group, which amplifies summation-order noise (measured rel-L2 2e-6 .. 8e-5 per frame)."""
    g = golden("unet_rollout")
    sm = golden("state_machine")
    h, w, N, FR = [int(v) for v in g["meta"]]
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    sd = {k: _fill(k, shp, 1.0) for k, shp in unet_param_spec(cfg).items()}
    kv = O.alloc_kv_cache(cfg, h, w, N)
    enc, ts = T(g["enc"]), T(g["tsteps"])
    for idx in range(N):
        o = O.unet_forward(sd, cfg, T(g["warm_x"])[idx:idx + 1], ts[idx:idx + 1], enc, T(g["warm_depth"]), kv,
                           mode="warmup", warmup_row=idx)
        close(o, T(g["warm_out"])[idx], 3e-4)
    for f in range(FR):
        o = O.unet_forward(sd, cfg, T(g["xs"])[f], ts, enc.repeat(N, 1, 1), T(g["ds"])[f], kv,
                           temporal_attention_mask=T(sm["bias_n2"])[f], pe_idx=T(sm["pe_idx_n2"])[f],
                           update_idx=T(sm["update_idx_n2"])[f])
        close(o, T(g["outs"])[f], 3e-4)
    cs = torch.stack([c.double().sum() for c in kv])
    cq = torch.stack([(c.double() ** 2).sum() for c in kv])
    numel = torch.tensor([float(c.numel()) for c in kv], dtype=torch.float64)
    assert ((cs - T(g["cache_sum"])).abs() <= 1e-4 * (cq * numel).sqrt()).all()   # |sum| <= sqrt(sumsq * n)
    assert torch.allclose(cq, T(g["cache_sq"]), rtol=5e-4)
    close(torch.stack([c[:, :, :1, :, :8] for c in kv]), T(g["cache_slice"]), 3e-4)
