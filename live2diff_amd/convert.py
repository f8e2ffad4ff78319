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
            w32 = w.to(torch.float32)
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


def build_state_dict(base_sd: Dict[str, torch.Tensor], cfg: UNetConfig, dreambooth: Optional[Dict[str, torch.Tensor]] = None,
                     loras: Optional[List[Tuple[Dict[str, torch.Tensor], float]]] = None) -> Dict[str, torch.Tensor]:
    """The reference's ingestion order (wrapper.py:417-466 / convert.py:11-134) on state dicts: base Live2Diff weights,
    DreamBooth spatial weights over them (`load_state_dict(strict=False)`: motion modules keep the base weights), then each
    LoRA merged at its strength.  The result is what `HipStreamingUNet(state_dict, ...)` packs and `save_packed` caches
    (cache name: `HipStreamingUNet.packed_cache_name`)."""
    sd = dict(base_sd)
    if dreambooth is not None:
        conv = convert_ldm_unet_checkpoint(dreambooth, cfg)
        unknown = [k for k in conv if k not in sd]
        if unknown:
            raise KeyError(f"{len(unknown)} converted DreamBooth keys are not UNet parameters, e.g. {unknown[:3]}")
        for k, v in conv.items():
            if tuple(v.shape) != tuple(sd[k].shape):
                raise ValueError(f"{k}: DreamBooth tensor {tuple(v.shape)} != UNet parameter {tuple(sd[k].shape)}")
            sd[k] = v.to(sd[k].dtype)
    for lora_sd, alpha in (loras or []):
        merge_lora(sd, lora_sd, alpha)
    return sd


def load_safetensors(path: str) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        return {k: f.get_tensor(k) for k in f.keys()}
