"""Minimal stand-in for the `diffusers==0.25.0` symbols the reference model files import.

TEST INFRASTRUCTURE ONLY.  Used by `gen_golden.py` in the build container to import the
reference's *own* model code (`/root/reference/live2diff/animatediff/models/*.py`) so that golden
input/output vectors can be captured.  `diffusers` is not installed in the image and there is no
network, so the third-party arithmetic (reference row A13 in SURVEY.md section 8: `Attention`,
`FeedForward`/GEGLU, `Timesteps`, `TimestepEmbedding`) is restated here from its documented
0.25.0 semantics.  Consequence: everything that flows through these classes is *stub-pinned*
("parity unpinned" by the reference itself) -- the reference-owned arithmetic (temporal attention,
KV-cache, PE, resnet/conv/GN plumbing, UNet topology) is what the goldens really pin.

Nothing in the product path (live2diff_amd/) imports this file.
"""
import math
import sys
import types
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- utils
class BaseOutput(OrderedDict):
    """dataclass + dict hybrid: supports out.sample, out["sample"], out[0]."""

    def __post_init__(self):
        if is_dataclass(self):
            for f in fields(self):
                v = getattr(self, f.name)
                if v is not None:
                    OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return tuple(self.values())[k]


class _Logger:
    def info(self, *a, **k):
        pass

    warning = debug = error = info


class _logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def is_xformers_available():
    return False


# ----------------------------------------------------------------------------- config / model mixins
class _Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def register_to_config(init):
    import functools
    import inspect

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        self._internal_config = _Config(cfg)
        init(self, *args, **kwargs)

    return wrapper


class ConfigMixin:
    @property
    def config(self):
        return self._internal_config

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config)
        cfg.update(kwargs)
        import inspect

        names = set(inspect.signature(cls.__init__).parameters)
        return cls(**{k: v for k, v in cfg.items() if k in names})


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class UNet2DConditionLoadersMixin:
    pass


class AttentionProcessor:
    pass


# ----------------------------------------------------------------------------- embeddings
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0,
                           scale=1.0, max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos,
                                      self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# ----------------------------------------------------------------------------- attention / FF
class Attention(nn.Module):
    """diffusers 0.25.0 `Attention` with the default AttnProcessor2_0 (SDPA), no group-norm,
    no added-kv, residual_connection=False, rescale_output_factor=1."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, **kwargs):
        super().__init__()
        inner = heads * dim_head
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kwargs):
        b, t, _ = hidden_states.shape
        enc = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = self.to_q(hidden_states)
        k = self.to_k(enc)
        v = self.to_v(enc)
        h = self.heads
        d = q.shape[-1] // h
        q = q.view(b, -1, h, d).transpose(1, 2)
        k = k.view(b, -1, h, d).transpose(1, 2)
        v = v.view(b, -1, h, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, h * d).to(q.dtype)
        o = self.to_out[0](o)
        o = self.to_out[1](o)
        return o


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu"
        inner = int(dim * mult)
        dim_out = dim_out or dim
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out)])

    def forward(self, x, scale=1.0):
        for m in self.net:
            x = m(x)
        return x


class AdaLayerNorm(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("AdaLayerNorm is not on the Live2Diff path (num_embeds_ada_norm=None)")


# ----------------------------------------------------------------------------- install
def install():
    """Register the stub as `diffusers` (+ the submodules the reference imports) in sys.modules."""
    if "diffusers" in sys.modules and not getattr(sys.modules["diffusers"], "_l2d_stub", False):
        return  # a real diffusers is present: use it
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    root = mod("diffusers", _l2d_stub=True)
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    mod("diffusers.loaders", UNet2DConditionLoadersMixin=UNet2DConditionLoadersMixin)
    models = mod("diffusers.models", ModelMixin=ModelMixin)
    mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    mod("diffusers.models.attention_processor", AttentionProcessor=AttentionProcessor)
    mod("diffusers.models.embeddings", Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding)
    mod("diffusers.models.attention", Attention=Attention, FeedForward=FeedForward, AdaLayerNorm=AdaLayerNorm)
    utils = mod("diffusers.utils", BaseOutput=BaseOutput, logging=_logging)
    mod("diffusers.utils.import_utils", is_xformers_available=is_xformers_available)
    root.models = models
    root.utils = utils
    return root
