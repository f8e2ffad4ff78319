"""Parameter inventory of the streaming UNet and a deterministic synthetic-weight generator.

The key names are the reference's `UNet3DConditionStreamingModel.state_dict()` names (so a real,
LoRA-fused fp16 state_dict produced by the reference loaders drops straight in); buffers that the
reference registers but that are pure functions of the config (`pos_encoder.pe`, `q_pe/k_pe/v_pe`)
are not part of the spec -- the backend recomputes them.

There is no network / no checkpoint in the build image, so tests and bench use *key-hashed* random
weights: every tensor is filled from a generator seeded with crc32(key).  Both the oracle and the
HIP backend consume the same dict, so only inputs/outputs ever need to be stored as fixtures.
"""
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import UNetConfig


def _resnet(spec, p, cin, cout, temb):
    spec[p + "norm1.weight"] = (cin,)
    spec[p + "norm1.bias"] = (cin,)
    spec[p + "conv1.weight"] = (cout, cin, 3, 3)
    spec[p + "conv1.bias"] = (cout,)
    spec[p + "time_emb_proj.weight"] = (cout, temb)
    spec[p + "time_emb_proj.bias"] = (cout,)
    spec[p + "norm2.weight"] = (cout,)
    spec[p + "norm2.bias"] = (cout,)
    spec[p + "conv2.weight"] = (cout, cout, 3, 3)
    spec[p + "conv2.bias"] = (cout,)
    if cin != cout:
        spec[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        spec[p + "conv_shortcut.bias"] = (cout,)


def _ff(spec, p, c):
    spec[p + "net.0.proj.weight"] = (8 * c, c)
    spec[p + "net.0.proj.bias"] = (8 * c,)
    spec[p + "net.2.weight"] = (c, 4 * c)
    spec[p + "net.2.bias"] = (c,)


def _spatial(spec, p, c, xdim):
    spec[p + "norm.weight"] = (c,)
    spec[p + "norm.bias"] = (c,)
    spec[p + "proj_in.weight"] = (c, c, 1, 1)
    spec[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    for a, kv in (("attn1", c), ("attn2", xdim)):
        spec[b + a + ".to_q.weight"] = (c, c)
        spec[b + a + ".to_k.weight"] = (c, kv)
        spec[b + a + ".to_v.weight"] = (c, kv)
        spec[b + a + ".to_out.0.weight"] = (c, c)
        spec[b + a + ".to_out.0.bias"] = (c,)
    for n in ("norm1", "norm2", "norm3"):
        spec[b + n + ".weight"] = (c,)
        spec[b + n + ".bias"] = (c,)
    _ff(spec, b + "ff.", c)
    spec[p + "proj_out.weight"] = (c, c, 1, 1)
    spec[p + "proj_out.bias"] = (c,)


def _motion(spec, p, c):
    t = p + "temporal_transformer."
    spec[t + "norm.weight"] = (c,)
    spec[t + "norm.bias"] = (c,)
    spec[t + "proj_in.weight"] = (c, c)
    spec[t + "proj_in.bias"] = (c,)
    b = t + "transformer_blocks.0."
    for j in range(2):
        a = b + f"attention_blocks.{j}."
        spec[a + "to_q.weight"] = (c, c)
        spec[a + "to_k.weight"] = (c, c)
        spec[a + "to_v.weight"] = (c, c)
        spec[a + "to_out.0.weight"] = (c, c)
        spec[a + "to_out.0.bias"] = (c,)
        spec[b + f"norms.{j}.weight"] = (c,)
        spec[b + f"norms.{j}.bias"] = (c,)
    _ff(spec, b + "ff.", c)
    spec[b + "ff_norm.weight"] = (c,)
    spec[b + "ff_norm.bias"] = (c,)
    spec[t + "proj_out.weight"] = (c, c)
    spec[t + "proj_out.bias"] = (c,)


def unet_param_spec(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for every learnable tensor of the streaming UNet (== warm-up UNet: the two share
    weights, reference unet_depth_warmup.py differs only in plumbing)."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ch = cfg.block_out_channels
    c0, temb = ch[0], cfg.time_embed_dim
    spec["conv_in.weight"] = (c0, cfg.in_channels, 3, 3)
    spec["conv_in.bias"] = (c0,)
    mc = cfg.mapping_channels
    spec["flow_conv_in.conv_in.weight"] = (mc[0], cfg.in_channels, 3, 3)
    spec["flow_conv_in.conv_in.bias"] = (mc[0],)
    k = 0
    for i in range(len(mc) - 1):
        spec[f"flow_conv_in.blocks.{k}.weight"] = (mc[i], mc[i], 3, 3)
        spec[f"flow_conv_in.blocks.{k}.bias"] = (mc[i],)
        spec[f"flow_conv_in.blocks.{k + 1}.weight"] = (mc[i + 1], mc[i], 3, 3)
        spec[f"flow_conv_in.blocks.{k + 1}.bias"] = (mc[i + 1],)
        k += 2
    spec["flow_conv_in.conv_out.weight"] = (c0, mc[-1], 3, 3)
    spec["flow_conv_in.conv_out.bias"] = (c0,)
    spec["time_embedding.linear_1.weight"] = (temb, c0)
    spec["time_embedding.linear_1.bias"] = (temb,)
    spec["time_embedding.linear_2.weight"] = (temb, temb)
    spec["time_embedding.linear_2.bias"] = (temb,)
    nl = cfg.num_levels
    cout = c0
    skip_ch = [c0]
    for i in range(nl):
        cin, cout = cout, ch[i]
        p = f"down_blocks.{i}."
        for j in range(cfg.layers_per_block):
            _resnet(spec, p + f"resnets.{j}.", cin if j == 0 else cout, cout, temb)
            if i != nl - 1:
                _spatial(spec, p + f"attentions.{j}.", cout, cfg.cross_attention_dim)
            _motion(spec, p + f"motion_modules.{j}.", cout)
            skip_ch.append(cout)
        if i != nl - 1:
            spec[p + "downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            spec[p + "downsamplers.0.conv.bias"] = (cout,)
            skip_ch.append(cout)
    cm = ch[-1]
    _resnet(spec, "mid_block.resnets.0.", cm, cm, temb)
    _spatial(spec, "mid_block.attentions.0.", cm, cfg.cross_attention_dim)
    _resnet(spec, "mid_block.resnets.1.", cm, cm, temb)
    rev = list(reversed(ch))
    prev = rev[0]
    for i in range(nl):
        cout = rev[i]
        p = f"up_blocks.{i}."
        for j in range(cfg.layers_per_block + 1):
            cin = (prev if j == 0 else cout) + skip_ch.pop()
            _resnet(spec, p + f"resnets.{j}.", cin, cout, temb)
            if i != 0:
                _spatial(spec, p + f"attentions.{j}.", cout, cfg.cross_attention_dim)
            _motion(spec, p + f"motion_modules.{j}.", cout)
        if i != nl - 1:
            spec[p + "upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            spec[p + "upsamplers.0.conv.bias"] = (cout,)
        prev = cout
    spec["conv_norm_out.weight"] = (c0,)
    spec["conv_norm_out.bias"] = (c0,)
    spec["conv_out.weight"] = (cfg.out_channels, c0, 3, 3)
    spec["conv_out.bias"] = (cfg.out_channels,)
    return spec


def _fill(key: str, shape, gain: float) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    leaf = key.rsplit(".", 1)[-1]
    is_norm = any(s in key for s in (".norm", "norms.", "ff_norm", "conv_norm_out"))
    if leaf == "bias":
        return 0.05 * x
    if is_norm:
        return 1.0 + 0.1 * x
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return x * (gain * fan_in ** -0.5)


def random_state_dict(cfg: UNetConfig, dtype=torch.float32, device="cpu", gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Key-hashed deterministic weights (seed = crc32(key); randn * gain * fan_in^-0.5; norm gains
    1 +- 0.1; biases 0.05 * randn).  The reference's zero-initialised layers (motion-module proj_out,
    mapping conv_out) are deliberately NON-zero here so that every path contributes to parity tests."""
    return OrderedDict((k, _fill(k, shp, gain).to(device=device, dtype=dtype)) for k, shp in unet_param_spec(cfg).items())


def device_random_state_dict(cfg: UNetConfig, device, dtype=torch.float16, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Same distribution as `random_state_dict`, generated directly on the device (seed = crc32(key) on the device
    generator): for benchmarks that do not need bit-identical CPU weights (seconds instead of ~1 min for 1.28 B)."""
    out = OrderedDict()
    g = torch.Generator(device=device)
    for k, shp in unet_param_spec(cfg).items():
        g.manual_seed(zlib.crc32(k.encode()) & 0x7FFFFFFF)
        x = torch.randn(shp, generator=g, dtype=torch.float32, device=device)
        leaf = k.rsplit(".", 1)[-1]
        is_norm = any(s in k for s in (".norm", "norms.", "ff_norm", "conv_norm_out"))
        if leaf == "bias":
            x = 0.05 * x
        elif is_norm:
            x = 1.0 + 0.1 * x
        else:
            fan_in = 1
            for s_ in shp[1:]:
                fan_in *= s_
            x = x * (gain * fan_in ** -0.5)
        out[k] = x.to(dtype)
    return out


def count_params(cfg: UNetConfig) -> int:
    n = 0
    for shp in unet_param_spec(cfg).values():
        m = 1
        for s in shp:
            m *= s
        n += m
    return n
