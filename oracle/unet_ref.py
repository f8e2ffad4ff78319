"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain PyTorch, fp32) of the reference's streaming UNet hot path:

  * `UNet3DConditionStreamingModel.forward`      reference unet_depth_streaming.py:429-627
  * `UNet3DConditionWarmupModel.forward`         reference unet_depth_warmup.py:407-590
  * block assemblies                             reference unet_blocks_streaming.py:253-280,381-445,
                                                            516-569,666-731,798-850
  * `ResnetBlock3D`, up/down-sample, mapping     reference resnet.py:44-54,94-127,145-153,229-259
  * `Transformer3DModel`/`BasicTransformerBlock` reference attention.py:91-135,221-270
  * temporal transformer + streaming attention   reference motion_module.py:256-299,401-435;
                                                            stream_motion_module.py:99-213
  * warm-up (bidirectional) temporal attention   reference motion_module.py:469-530

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module, and only as the *checker* / the timed CPU baseline.  The product (`live2diff_amd/`) never
imports it.

Pinning: checked against golden vectors captured from the reference's own classes imported in the
build container (tests/golden/gen_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).
The arithmetic that the reference delegates to `diffusers==0.25.0` (spatial `Attention`, GEGLU
`FeedForward`, `Timesteps`, `TimestepEmbedding`) is absent from /root/reference and not installed
here; those pieces are restated from the documented 0.25.0 semantics on BOTH sides of the golden
comparison, i.e. they are **parity unpinned** by the reference.

Layout: activations are NCHW `[B, C, H, W]` with B = denoising-batch N (streaming) or the F warm-up
frames; KV caches use the reference interchange layout `[N, 2, H*W, L, C]`.
The weights are a plain `state_dict` with the reference's key names.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- helpers
class _W:
    """state_dict accessor with a key prefix; everything is promoted to fp32."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name: str) -> torch.Tensor:
        return self.sd[self.prefix + name].float()

    def has(self, name: str) -> bool:
        return (self.prefix + name) in self.sd

    def sub(self, name: str) -> "_W":
        return _W(self.sd, self.prefix + name + ".")


def sinusoid_pe(max_len: int, dim: int) -> torch.Tensor:
    """reference positional_encoding.py:12-16 -> [max_len, dim]"""
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000.0) / dim))
    pe = torch.zeros(max_len, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def timestep_sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers 0.25.0 `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`
    (call site reference unet_depth_streaming.py:102,499)."""
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.float()[:, None] * freq[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _gn(x, w: _W, name, groups, eps, silu=False):
    y = F.group_norm(x, groups, w(name + ".weight"), w(name + ".bias"), eps)
    return F.silu(y) if silu else y


def _conv(x, w: _W, name, stride=1, padding=1):
    return F.conv2d(x, w(name + ".weight"), w(name + ".bias"), stride=stride, padding=padding)


def _tokens(x):  # [B,C,H,W] -> [B,HW,C]
    b, c, h, ww = x.shape
    return x.permute(0, 2, 3, 1).reshape(b, h * ww, c)


def _image(x, h, ww):  # [B,HW,C] -> [B,C,H,W]
    b, t, c = x.shape
    return x.reshape(b, h, ww, c).permute(0, 3, 1, 2)


def _lin(x, w: _W, name):
    b = w(name + ".bias") if w.has(name + ".bias") else None
    return F.linear(x, w(name + ".weight"), b)


def _mha(q, k, v, heads, bias=None):
    """softmax(q k^T / sqrt(d) + bias) v over the second-to-last axis; q [...,Tq,C], k/v [...,Tk,C]."""
    *lead, tq, c = q.shape
    tk = k.shape[-2]
    d = c // heads
    q = q.reshape(*lead, tq, heads, d).transpose(-2, -3)
    k = k.reshape(*lead, tk, heads, d).transpose(-2, -3)
    v = v.reshape(*lead, tk, heads, d).transpose(-2, -3)
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    if bias is not None:
        s = s + bias
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v)
    return o.transpose(-2, -3).reshape(*lead, tq, c)


def _geglu_ff(x, w: _W):
    """diffusers 0.25.0 FeedForward(activation_fn='geglu'): Linear(C,8C) -> h*gelu_erf(g) -> Linear(4C,C)."""
    h, g = _lin(x, w, "net.0.proj").chunk(2, dim=-1)
    return _lin(h * F.gelu(g), w, "net.2")


# ----------------------------------------------------------------------------- blocks
def resnet_block(x, temb, w: _W, cfg):
    """reference resnet.py:229-259 (time_embedding_norm='default', output_scale_factor=1)."""
    h = _gn(x, w, "norm1", cfg.norm_num_groups, cfg.norm_eps, silu=True)
    h = _conv(h, w, "conv1")
    h = h + _lin(F.silu(temb), w, "time_emb_proj")[:, :, None, None]
    h = _gn(h, w, "norm2", cfg.norm_num_groups, cfg.norm_eps, silu=True)
    h = _conv(h, w, "conv2")
    if w.has("conv_shortcut.weight"):
        x = _conv(x, w, "conv_shortcut", padding=0)
    return x + h


def spatial_transformer(x, enc, w: _W, cfg):
    """reference attention.py:91-135 + 221-270 (use_linear_projection=False, one block)."""
    b, c, h, ww = x.shape
    res = x
    y = _gn(x, w, "norm", cfg.norm_num_groups, cfg.transformer_norm_eps)
    y = _conv(y, w, "proj_in", padding=0)
    y = _tokens(y)
    blk = w.sub("transformer_blocks.0")
    n1 = F.layer_norm(y, (c,), blk("norm1.weight"), blk("norm1.bias"))
    a = blk.sub("attn1")
    y = _lin(_mha(_lin(n1, a, "to_q"), _lin(n1, a, "to_k"), _lin(n1, a, "to_v"), cfg.num_heads), a, "to_out.0") + y
    n2 = F.layer_norm(y, (c,), blk("norm2.weight"), blk("norm2.bias"))
    a = blk.sub("attn2")
    y = _lin(_mha(_lin(n2, a, "to_q"), _lin(enc, a, "to_k"), _lin(enc, a, "to_v"), cfg.num_heads), a, "to_out.0") + y
    n3 = F.layer_norm(y, (c,), blk("norm3.weight"), blk("norm3.bias"))
    y = _geglu_ff(n3, blk.sub("ff")) + y
    y = _image(y, h, ww)
    y = _conv(y, w, "proj_out", padding=0)
    return y + res


def stream_temporal_core(q, k, v, w: _W, cfg, cache, bias, pe_idx, update_idx, pe):
    """The cache/PE/softmax core of reference stream_motion_module.py:112-194 on already projected
    q,k,v [N,T,C]: returns the attention output BEFORE to_out.  cache [N,2,T,L,C] is mutated in place."""
    n, t, c = q.shape
    L = cfg.window_size
    for i in range(n):                                  # :117-119
        cache[i, 0, :, update_idx[i]] = k[i].to(cache.dtype)
        cache[i, 1, :, update_idx[i]] = v[i].to(cache.dtype)
    pe_l = pe[:L]                                       # prepare_pe_buffer :79-97
    q_pe, k_pe, v_pe = (F.linear(pe_l, w(nm + ".weight")) for nm in ("to_q", "to_k", "to_v"))
    q_idx = torch.stack([pe_idx[i, update_idx[i]] for i in range(n)])       # :125-127
    qf = q.float() + q_pe[q_idx][:, None, :]                                # :139
    kf = cache[:, 0].float() + k_pe[pe_idx][:, None]                        # :140  [N,T,L,C]
    vf = cache[:, 1].float() + v_pe[pe_idx][:, None]                        # :141
    o = _mha(qf[:, :, None, :], kf, vf, cfg.temporal_heads, bias=bias.float()[:, None, None, None, :])
    return o[:, :, 0, :]


def stream_temporal_attention(x, w: _W, cfg, cache, bias, pe_idx, update_idx, pe):
    """reference stream_motion_module.py:99-213.
    x [N,T,C] (layer-normed tokens), cache [N,2,T,L,C] MUTATED IN PLACE (pre-PE projections),
    bias [N,L] additive (0/-inf), pe_idx [N,L] int64, update_idx [N] int64, pe [max_len,C]."""
    q, k, v = _lin(x, w, "to_q"), _lin(x, w, "to_k"), _lin(x, w, "to_v")
    o = stream_temporal_core(q, k, v, w, cfg, cache, bias, pe_idx, update_idx, pe)
    return _lin(o, w, "to_out.0")


def warmup_temporal_core(q, k, v, w: _W, cfg, cache_row, pe):
    """Core of reference motion_module.py:492-510 on projected q,k,v [T,F,C] (per pixel sequences)."""
    f = q.shape[1]
    cache_row[0, :, :f, :] = k.to(cache_row.dtype)       # :492-493
    cache_row[1, :, :f, :] = v.to(cache_row.dtype)
    pe_f = pe[:f]
    q = q.float() + F.linear(pe_f, w("to_q.weight"))
    k = k.float() + F.linear(pe_f, w("to_k.weight"))
    v = v.float() + F.linear(pe_f, w("to_v.weight"))
    return _mha(q, k, v, cfg.temporal_heads)


def warmup_temporal_attention(x, w: _W, cfg, cache_row, pe):
    """reference motion_module.py:469-530 (VersatileAttention, no mask).
    x [F,T,C] warm-up frames; cache_row [2,T,L,C]: slots 0..F-1 receive the pre-PE K / V."""
    xt = x.transpose(0, 1)                               # "(b f) d c -> (b d) f c"  [T,F,C]
    q, k, v = _lin(xt, w, "to_q"), _lin(xt, w, "to_k"), _lin(xt, w, "to_v")
    o = warmup_temporal_core(q, k, v, w, cfg, cache_row, pe)
    return _lin(o, w, "to_out.0").transpose(0, 1)


def motion_module(x, w: _W, cfg, attn_fn, idx_base):
    """reference motion_module.py:256-299 (transformer) + :401-435 (block); attn_fn(tokens, w_attn, idx)."""
    b, c, h, ww = x.shape
    tw = w.sub("temporal_transformer")
    y = _gn(x, tw, "norm", cfg.norm_num_groups, cfg.transformer_norm_eps)
    y = _lin(_tokens(y), tw, "proj_in")
    blk = tw.sub("transformer_blocks.0")
    for j in range(2):
        nrm = F.layer_norm(y, (c,), blk(f"norms.{j}.weight"), blk(f"norms.{j}.bias"))
        y = attn_fn(nrm, blk.sub(f"attention_blocks.{j}"), idx_base + j) + y
    nrm = F.layer_norm(y, (c,), blk("ff_norm.weight"), blk("ff_norm.bias"))
    y = _geglu_ff(nrm, blk.sub("ff")) + y
    y = _lin(y, tw, "proj_out")
    return _image(y, h, ww) + x


def mapping_network(d, w: _W):
    """reference resnet.py:44-54"""
    e = F.silu(_conv(d, w, "conv_in"))
    i = 0
    while w.has(f"blocks.{i}.weight"):
        e = F.silu(_conv(e, w, f"blocks.{i}"))
        i += 1
    return _conv(e, w, "conv_out")


# ----------------------------------------------------------------------------- full UNet
@torch.no_grad()
def unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, depth_sample, kv_cache: List[torch.Tensor],
                 temporal_attention_mask: Optional[torch.Tensor] = None, pe_idx=None, update_idx=None,
                 mode: str = "stream", warmup_row: int = 0):
    """One UNet pass.

    mode == "stream": sample/depth [N,4,1,h,w], timestep [N], enc [N,77,D], mask [N,L], kv_cache list of
                      [N,2,T,L,C] mutated in place (reference unet_depth_streaming.py:429-627).
    mode == "warmup": sample/depth [1,4,F,h,w], timestep [1], enc [1,77,D]; kv_cache is the list of FULL
                      caches and `warmup_row` selects cache[warmup_row] (reference pipeline :317-328 passes
                      `[cache[idx] for cache in kv_cache_list]`).
    Returns the predicted noise with the input's shape.
    """
    w = _W(sd)
    if mode == "stream":
        x = sample[:, :, 0].float()
        d = depth_sample[:, :, 0].float()
        enc = encoder_hidden_states.float()
    else:
        x = sample[0].transpose(0, 1).float()            # [F,4,h,w]
        d = depth_sample[0].transpose(0, 1).float()
        enc = encoder_hidden_states.float().expand(x.shape[0], -1, -1)
    B = x.shape[0]
    c0 = cfg.block_out_channels[0]
    temb = timestep_sinusoid(timestep.reshape(-1), c0)
    temb = _lin(F.silu(_lin(temb, w, "time_embedding.linear_1")), w, "time_embedding.linear_2")
    temb = temb.expand(B, -1)
    pe_tables = {}

    def attn_fn(tokens, wa, idx):
        c = tokens.shape[-1]
        if c not in pe_tables:
            pe_tables[c] = sinusoid_pe(cfg.temporal_max_len, c)
        if mode == "stream":
            return stream_temporal_attention(tokens, wa, cfg, kv_cache[idx], temporal_attention_mask,
                                             pe_idx, update_idx, pe_tables[c])
        return warmup_temporal_attention(tokens, wa, cfg, kv_cache[idx][warmup_row], pe_tables[c])

    x = _conv(x, w, "conv_in") + mapping_network(d, w.sub("flow_conv_in"))
    skips = [x]
    mm = 0
    nl = cfg.num_levels
    for i in range(nl):
        bw = w.sub(f"down_blocks.{i}")
        for j in range(cfg.layers_per_block):
            x = resnet_block(x, temb, bw.sub(f"resnets.{j}"), cfg)
            if bw.has(f"attentions.{j}.norm.weight"):
                x = spatial_transformer(x, enc, bw.sub(f"attentions.{j}"), cfg)
            x = motion_module(x, bw.sub(f"motion_modules.{j}"), cfg, attn_fn, mm)
            mm += 2
            skips.append(x)
        if i != nl - 1:
            x = _conv(x, bw, "downsamplers.0.conv", stride=2)
            skips.append(x)
    mw = w.sub("mid_block")
    x = resnet_block(x, temb, mw.sub("resnets.0"), cfg)
    x = spatial_transformer(x, enc, mw.sub("attentions.0"), cfg)
    x = resnet_block(x, temb, mw.sub("resnets.1"), cfg)
    for i in range(nl):
        bw = w.sub(f"up_blocks.{i}")
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(x, temb, bw.sub(f"resnets.{j}"), cfg)
            if bw.has(f"attentions.{j}.norm.weight"):
                x = spatial_transformer(x, enc, bw.sub(f"attentions.{j}"), cfg)
            x = motion_module(x, bw.sub(f"motion_modules.{j}"), cfg, attn_fn, mm)
            mm += 2
        if i != nl - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(x, bw, "upsamplers.0.conv")
    x = _gn(x, w, "conv_norm_out", cfg.norm_num_groups, cfg.norm_eps, silu=True)
    x = _conv(x, w, "conv_out")
    if mode == "stream":
        return x[:, :, None]
    return x.transpose(0, 1)[None]


def alloc_kv_cache(cfg, h, w, n, dtype=torch.float32, device="cpu"):
    """reference stream_motion_module.py:57-77 + unet_depth_streaming.py:283-302: list of zero caches
    [N,2,h*w,L,C] in motion_module_idx order."""
    from live2diff_amd.config import motion_module_layout

    return [torch.zeros(n, 2, hh * ww, cfg.window_size, c, dtype=dtype, device=device)
            for (c, hh, ww, _lvl) in motion_module_layout(cfg, h, w)]
