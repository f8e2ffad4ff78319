"""DPT-Hybrid (MiDaS `DPTDepthModel(backbone="vitb_rn50_384")`, reference depth_utils.py:11-32) parameter inventory and
deterministic random weights: state-dict names and shapes in the MiDaS / timm key scheme.  Plain torch -- this module does not load
libl2d_hip.so, so fixture generators (tests/golden/gen_golden_midas.py) run on a fresh checkout before anything is built."""
from typing import Dict

import torch

STAGES = (3, 4, 9)
STAGE_CH = (256, 512, 1024)
DIM, DEPTH, HEADS, MLP = 768, 12, 12, 3072
HOOKS = (8, 11)
FEAT, G = 256, 32


def midas_param_spec(img: int = 384):
    """MiDaS / timm state-dict names -> shapes of DPT-Hybrid (same inventory as oracle/midas_ref.py, kept in the product so that
    it never imports the oracle; tests assert the two agree)."""
    s = {}
    bb = "pretrained.model.patch_embed.backbone."
    s[bb + "stem.conv.weight"] = (64, 3, 7, 7)
    s[bb + "stem.norm.weight"] = s[bb + "stem.norm.bias"] = (64,)
    cin = 64
    for si, (nb, cout) in enumerate(zip(STAGES, STAGE_CH)):
        mid = cout // 4
        for bi in range(nb):
            p = bb + f"stages.{si}.blocks.{bi}."
            if bi == 0:
                s[p + "downsample.conv.weight"] = (cout, cin, 1, 1)
                s[p + "downsample.norm.weight"] = s[p + "downsample.norm.bias"] = (cout,)
            s[p + "conv1.weight"] = (mid, cin, 1, 1)
            s[p + "norm1.weight"] = s[p + "norm1.bias"] = (mid,)
            s[p + "conv2.weight"] = (mid, mid, 3, 3)
            s[p + "norm2.weight"] = s[p + "norm2.bias"] = (mid,)
            s[p + "conv3.weight"] = (cout, mid, 1, 1)
            s[p + "norm3.weight"] = s[p + "norm3.bias"] = (cout,)
            cin = cout
    m = "pretrained.model."
    s[m + "patch_embed.proj.weight"] = (DIM, STAGE_CH[-1], 1, 1)
    s[m + "patch_embed.proj.bias"] = (DIM,)
    s[m + "cls_token"] = (1, 1, DIM)
    s[m + "pos_embed"] = (1, (img // 16) ** 2 + 1, DIM)
    for i in range(DEPTH):
        p = m + f"blocks.{i}."
        s.update({p + "norm1.weight": (DIM,), p + "norm1.bias": (DIM,), p + "attn.qkv.weight": (3 * DIM, DIM), p + "attn.qkv.bias": (3 * DIM,),
                  p + "attn.proj.weight": (DIM, DIM), p + "attn.proj.bias": (DIM,), p + "norm2.weight": (DIM,), p + "norm2.bias": (DIM,),
                  p + "mlp.fc1.weight": (MLP, DIM), p + "mlp.fc1.bias": (MLP,), p + "mlp.fc2.weight": (DIM, MLP), p + "mlp.fc2.bias": (DIM,)})
    for k in (3, 4):
        p = f"pretrained.act_postprocess{k}."
        s.update({p + "0.project.0.weight": (DIM, 2 * DIM), p + "0.project.0.bias": (DIM,), p + "3.weight": (DIM, DIM, 1, 1), p + "3.bias": (DIM,)})
    s["pretrained.act_postprocess4.4.weight"] = (DIM, DIM, 3, 3)
    s["pretrained.act_postprocess4.4.bias"] = (DIM,)
    for k, c in zip((1, 2, 3, 4), (256, 512, DIM, DIM)):
        s[f"scratch.layer{k}_rn.weight"] = (FEAT, c, 3, 3)
    for k in (1, 2, 3, 4):
        for u in (1, 2):
            for c in (1, 2):
                s[f"scratch.refinenet{k}.resConfUnit{u}.conv{c}.weight"] = (FEAT, FEAT, 3, 3)
                s[f"scratch.refinenet{k}.resConfUnit{u}.conv{c}.bias"] = (FEAT,)
        s[f"scratch.refinenet{k}.out_conv.weight"] = (FEAT, FEAT, 1, 1)
        s[f"scratch.refinenet{k}.out_conv.bias"] = (FEAT,)
    s.update({"scratch.output_conv.0.weight": (FEAT // 2, FEAT, 3, 3), "scratch.output_conv.0.bias": (FEAT // 2,),
              "scratch.output_conv.2.weight": (32, FEAT // 2, 3, 3), "scratch.output_conv.2.bias": (32,),
              "scratch.output_conv.4.weight": (1, 32, 1, 1), "scratch.output_conv.4.bias": (1,)})
    return s


def random_midas_state_dict(dtype=torch.float16, device="cpu", img: int = 384) -> Dict[str, torch.Tensor]:
    """Key-hashed deterministic weights (seed = crc32(key)); token embeddings small, norms 1 +- 0.1, like the UNet's recipe."""
    from .weights import _fill
    out = {}
    for k, shp in midas_param_spec(img).items():
        if k.endswith(("cls_token", "pos_embed")):
            out[k] = (0.02 * _fill("midas." + k + ".bias", shp, 1.0) / 0.05).to(device=device, dtype=dtype)
        else:
            out[k] = _fill("midas." + k, shp, 1.0).to(device=device, dtype=dtype)
    # the head ends in ReLU(conv1x1): with zero-mean random weights 4/5 of the synthetic depth map would be clipped to 0
    out["scratch.output_conv.4.bias"] = out["scratch.output_conv.4.bias"] + 6.0
    return out
