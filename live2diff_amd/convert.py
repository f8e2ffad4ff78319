"""Weight ingestion for the packed-weight cache: third-party style checkpoints -> the
reference-keyed UNet state dict that `HipStreamingUNet` packs (and `save_packed` caches on disk).

The reference does this on live `nn.Module`s before the accelerator is built (wrapper.py:417-466):
  * a DreamBooth checkpoint in LDM / CompVis key layout is renamed to diffusers keys and loaded over the spatial weights
    (animatediff/converter/convert.py:26-38 -> convert_from_ckpt.py:245-474 `convert_ldm_unet_checkpoint`);
  * kohya-style LoRA files are merged into the weights, `W += alpha * up @ down`
    (convert.py:72-88 -> convert_lora_safetensor_to_diffusers.py:22-101 `convert_lora_model_level`), for the streaming and
    the warm-up UNet alike (convert.py:106-134) -- which share ONE set of weights here.
This module does the same on plain state dicts (no module tree, no diffusers): key renaming from the UNet topology, and a
LoRA merge that resolves kohya's underscore-flattened module names against the state-dict keys.  Both are pinned against
the reference's own functions run in the build container (tests/golden/gen_golden_convert.py -> convert_*.json / .npz).
"""
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .config import UNetConfig

LDM_UNET_PREFIX = "model.diffusion_model."

_RESNET = (("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("emb_layers.1", "time_emb_proj"), ("out_layers.0", "norm2"),
           ("out_layers.3", "conv2"), ("skip_connection", "conv_shortcut"))


def ldm_unet_key_map(cfg: UNetConfig) -> Dict[str, str]:
    """{LDM module prefix -> diffusers module prefix} for the SD-1.x UNet topology described by `cfg`
    (what convert_from_ckpt.py:245-474 derives by scanning the checkpoint).  Prefixes, without `.weight` / `.bias`."""
    m = {"time_embed.0": "time_embedding.linear_1", "time_embed.2": "time_embedding.linear_2", "input_blocks.0.0": "conv_in",
         "out.0": "conv_norm_out", "out.2": "conv_out"}
    nl, lpb = cfg.num_levels, cfg.layers_per_block

    def resnet(old, new):
        for a, b in _RESNET:
            m[f"{old}.{a}"] = f"{new}.{b}"

    i = 1
    for lvl in range(nl):
        for j in range(lpb):
            resnet(f"input_blocks.{i}.0", f"down_blocks.{lvl}.resnets.{j}")
            if lvl != nl - 1:
                m[f"input_blocks.{i}.1"] = f"down_blocks.{lvl}.attentions.{j}"
            i += 1
        if lvl != nl - 1:
            m[f"input_blocks.{i}.0.op"] = f"down_blocks.{lvl}.downsamplers.0.conv"
            i += 1
    resnet("middle_block.0", "mid_block.resnets.0")
    m["middle_block.1"] = "mid_block.attentions.0"
    resnet("middle_block.2", "mid_block.resnets.1")
    i = 0
    for lvl in range(nl):
        for j in range(lpb + 1):
            resnet(f"output_blocks.{i}.0", f"up_blocks.{lvl}.resnets.{j}")
            has_attn = lvl != 0
            if has_attn:
                m[f"output_blocks.{i}.1"] = f"up_blocks.{lvl}.attentions.{j}"
            if j == lpb and lvl != nl - 1:
                m[f"output_blocks.{i}.{2 if has_attn else 1}.conv"] = f"up_blocks.{lvl}.upsamplers.0.conv"
            i += 1
    return m


def convert_ldm_unet_checkpoint(checkpoint: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """LDM-layout checkpoint (keys `model.diffusion_model.*`; other entries ignored) -> diffusers-keyed spatial UNet weights.
    Same result as the reference's convert_ldm_unet_checkpoint for the SD-1.x topology (non-EMA weights)."""
    kmap = ldm_unet_key_map(cfg)
    prefixes = sorted(kmap, key=len, reverse=True)
    out = {}
    for k, v in checkpoint.items():
        if not k.startswith(LDM_UNET_PREFIX):
            continue
        name = k[len(LDM_UNET_PREFIX):]
        for p in prefixes:
            if name == p or name.startswith(p + "."):
                out[kmap[p] + name[len(p):]] = v
                break
        else:
            raise KeyError(f"LDM UNet key {k!r} has no place in the SD-1.x topology of this config")
    return out


def _flatten(key: str) -> str:
    return key.replace(".", "_")


def lora_pairs(lora_sd: Dict[str, torch.Tensor], prefix: str = "lora_unet") -> Iterable[Tuple[str, torch.Tensor, torch.Tensor]]:
    """(flattened module name, up, down) for every UNet LoRA pair of a kohya-style state dict; `.alpha` entries are skipped
    like the reference does (convert_lora_safetensor_to_diffusers.py:35-36: the strength is the caller's `alpha`)."""
    for k in lora_sd:
        if ".alpha" in k or "lora_down" not in k or "text" in k or not k.startswith(prefix + "_"):
            continue
        up = k.replace("lora_down", "lora_up")
        yield k.split(".")[0][len(prefix) + 1:], lora_sd[up], lora_sd[k]


def merge_lora(sd: Dict[str, torch.Tensor], lora_sd: Dict[str, torch.Tensor], alpha: float = 0.6, strict: bool = False) -> List[str]:
    """In place: sd[weight] += alpha * up @ down for every UNet pair of a kohya LoRA (reference
    convert_lora_safetensor_to_diffusers.py:22-101).  Returns the touched keys.  Pairs whose module does not exist in `sd`
    raise with strict=True, else are skipped (the reference's attribute walk would raise)."""
    flat = {_flatten(k[: -len(".weight")]): k for k in sd if k.endswith(".weight")}
    touched = []
    for name, up, down in lora_pairs(lora_sd):
        key = flat.get(name)
        if key is None:
            if strict:
                raise KeyError(f"LoRA module {name!r} not found in the UNet state dict")
            continue
        w = sd[key]
        u, d = up.to(torch.float32), down.to(torch.float32)
        if "conv_in" in name:
            # the streaming UNet's conv_in may be wider than the LoRA's 4 input channels: only the first 4 are touched (:75-81)
            delta = (u.reshape(u.shape[0], -1) @ d.reshape(d.shape[0], -1)).reshape(w.shape[0], 4, *w.shape[2:])
            w32 = w.to(torch.float32).clone()          # (never the caller's tensor: `.to` is a no-op on fp32 weights)
            w32[:, :4] += alpha * delta
        elif "conv" in name:
            delta = (u.reshape(u.shape[0], -1) @ d.reshape(d.shape[0], -1)).reshape(w.shape)
            w32 = w.to(torch.float32) + alpha * delta
        elif up.dim() == 4:
            delta = (u.squeeze(3).squeeze(2) @ d.squeeze(3).squeeze(2)).unsqueeze(2).unsqueeze(3)
            w32 = w.to(torch.float32) + alpha * delta
        else:
            w32 = w.to(torch.float32) + alpha * (u @ d)
        sd[key] = w32.to(w.dtype)
        touched.append(key)
    return touched


def merge_motion_checkpoint(sd: Dict[str, torch.Tensor], ckpt: Dict) -> List[str]:
    """In place: the `live2diff.ckpt` merge of the reference (animatediff/pipeline/pipeline_animatediff_depth.py:281-290):
    take `ckpt["state_dict"]` when present, strip `module.` from the keys, drop the `grid` buffers, load the rest over the
    UNet weights (`load_state_dict(strict=False)` + the reference's `assert len(unexpected) == 0`).  Returns the keys set."""
    state = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    state = {k.replace("module.", ""): v for k, v in state.items() if "grid" not in k and torch.is_tensor(v)}
    unexpected = [k for k in state if k not in sd]
    if unexpected:
        raise KeyError(f"Find unexpected keys ({len(unexpected)}): {unexpected[:5]}")
    for k, v in state.items():
        if tuple(v.shape) != tuple(sd[k].shape):
            raise ValueError(f"{k}: checkpoint tensor {tuple(v.shape)} != UNet parameter {tuple(sd[k].shape)}")
        sd[k] = v.to(sd[k].dtype)
    return list(state)


_DOWN_TAGS = (".lora_down.weight", ".lora.down.weight", ".lora_A.weight", "_lora.down.weight")
_UP_OF = {".lora_down.weight": ".lora_up.weight", ".lora.down.weight": ".lora.up.weight", ".lora_A.weight": ".lora_B.weight",
          "_lora.down.weight": "_lora.up.weight"}


def few_step_lora_pairs(lora_sd: Dict[str, torch.Tensor]):
    """(state-dict weight key candidate, up, down, alpha or None) for every UNet pair of a few-step (LCM) LoRA file.  The
    published `latent-consistency/lcm-lora-sdv1-5` file is kohya-keyed (`lora_unet_<flattened module>.lora_down.weight` +
    `.alpha`); diffusers 0.25.0's loader (the third-party code behind wrapper.py:451-452 -> pipeline/loader.py:12-31,
    `LoraLoaderMixin.lora_state_dict`) also accepts its own layouts, which are resolved here too:
    `unet.<module>.lora.down.weight`, `unet.<module>.lora_A.weight` (peft) and the attention-processor form
    `unet.<block>.attn1.processor.to_q_lora.down.weight`.  The first tuple element is the flattened (underscore) module
    name for kohya keys and the dotted module path otherwise."""
    for k, down in lora_sd.items():
        tag = next((t for t in _DOWN_TAGS if k.endswith(t)), None)
        if tag is None or "text" in k.split(".")[0]:
            continue
        up = lora_sd[k[: -len(tag)] + _UP_OF[tag]]
        stem = k[: -len(tag)]
        if stem.startswith("lora_unet_"):
            yield ("flat", stem[len("lora_unet_"):], up, down, lora_sd.get(stem + ".alpha"))
            continue
        if stem.startswith("lora_te"):
            continue
        if stem.startswith("unet."):
            stem = stem[len("unet."):]
        stem = stem.replace(".processor.", ".")                     # attn1.processor.to_q(_lora) -> attn1.to_q
        if stem.endswith("to_out"):
            stem += ".0"                                           # processor form names the Sequential, the weight is its [0]
        yield ("dotted", stem, up, down, lora_sd.get(k[: -len(tag)] + ".alpha"))


def merge_few_step_lora(sd: Dict[str, torch.Tensor], lora_sd: Dict[str, torch.Tensor], lora_scale: float = 1.0,
                        strict: bool = True) -> List[str]:
    """In place: `W += lora_scale * (alpha / rank) * up @ down` for every UNet pair -- the arithmetic of diffusers 0.25.0's
    `fuse_lora` (`LoRACompatibleLinear._fuse_lora` / `LoRACompatibleConv._fuse_lora`: `w + lora_scale * bmm(up, down)` with
    `network_alpha / rank` folded into `up`), which the reference calls through `stream.load_lora(few_step_lora);
    stream.fuse_lora()` (wrapper.py:451-452) BEFORE `prepare_cache` projects the positional-encoding tables (:454-459) --
    so a few-step LoRA that targets motion-module projections must be merged before `HipStreamingUNet` packs the weights,
    which is exactly where `build_state_dict` puts it.  diffusers is not in /root/reference: **parity unpinned**, checked
    against the formula (tests/test_convert.py).  Unresolved modules raise with strict=True (default), like the loader."""
    flat = {_flatten(k[: -len(".weight")]): k for k in sd if k.endswith(".weight")}
    touched = []
    for kind, name, up, down, alpha in few_step_lora_pairs(lora_sd):
        key = flat.get(name) if kind == "flat" else (name + ".weight" if name + ".weight" in sd else None)
        if key is None:
            if strict:
                raise KeyError(f"few-step LoRA module {name!r} not found in the UNet state dict")
            continue
        w = sd[key]
        u, d = up.to(torch.float32), down.to(torch.float32)
        rank = d.shape[0]
        scale = lora_scale * (float(alpha) / rank if alpha is not None else 1.0)
        delta = (u.reshape(u.shape[0], -1) @ d.reshape(d.shape[0], -1)).reshape(w.shape)
        sd[key] = (w.to(torch.float32) + scale * delta).to(w.dtype)
        touched.append(key)
    if strict and not touched:
        raise ValueError("the few-step LoRA touched no UNet weight: unknown key layout")
    return touched


def build_state_dict(base_sd: Dict[str, torch.Tensor], cfg: UNetConfig, dreambooth: Optional[Dict[str, torch.Tensor]] = None,
                     loras: Optional[List[Tuple[Dict[str, torch.Tensor], float]]] = None,
                     motion_ckpt: Optional[Dict] = None, few_step_lora: Optional[Dict[str, torch.Tensor]] = None,
                     strict_lora: bool = True) -> Dict[str, torch.Tensor]:
    """The reference's ingestion order on state dicts, from raw checkpoints alone:
      1. base SD-1.5 UNet weights inflated into the streaming topology (`from_pretrained_2d`: the caller's `base_sd`);
      2. `motion_ckpt` = live2diff.ckpt over them (pipeline_animatediff_depth.py:281-290, `merge_motion_checkpoint`);
      3. DreamBooth spatial weights (`load_third_party_checkpoints`, :303; `load_state_dict(strict=False)`: motion modules
         keep their weights);
      4. the few-step (LCM) LoRA at scale 1 (wrapper.py:451-452) -- before the PE tables are projected (:454-459), i.e. before
         packing;
      5. each user LoRA at its strength (wrapper.py:461-466; kohya files, convert.py:72-88).
    A LoRA that resolves against NO weight raises (strict_lora=False: warns) instead of silently merging nothing.
    The result is what `HipStreamingUNet(state_dict, ...)` packs and `save_packed` caches
    (cache name: `HipStreamingUNet.packed_cache_name`)."""
    sd = dict(base_sd)
    if motion_ckpt is not None:
        merge_motion_checkpoint(sd, motion_ckpt)
    if dreambooth is not None:
        conv = convert_ldm_unet_checkpoint(dreambooth, cfg)
        unknown = [k for k in conv if k not in sd]
        if unknown:
            raise KeyError(f"{len(unknown)} converted DreamBooth keys are not UNet parameters, e.g. {unknown[:3]}")
        for k, v in conv.items():
            if tuple(v.shape) != tuple(sd[k].shape):
                raise ValueError(f"{k}: DreamBooth tensor {tuple(v.shape)} != UNet parameter {tuple(sd[k].shape)}")
            sd[k] = v.to(sd[k].dtype)
    if few_step_lora is not None:
        merge_few_step_lora(sd, few_step_lora, 1.0, strict=strict_lora)
    for lora_sd, alpha in (loras or []):
        touched = merge_lora(sd, lora_sd, alpha)
        n_pairs = sum(1 for _ in lora_pairs(lora_sd))
        if len(touched) < n_pairs or not touched:
            msg = (f"LoRA merge resolved {len(touched)} of {n_pairs} UNet pairs against the state dict (a peft / diffusers-"
                   "keyed file, or another topology?)")
            if strict_lora:
                raise KeyError(msg)
            import warnings
            warnings.warn(msg)
    return sd


def load_safetensors(path: str) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        return {k: f.get_tensor(k) for k in f.keys()}
