"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain PyTorch, fp32) of the depth detector the reference runs once per frame (SURVEY.md section 8f row
F2): `MidasDetector` = `DPTDepthModel(backbone="vitb_rn50_384", non_negative=True)` (live2diff/animatediff/models/
depth_utils.py:11-32), called from pipeline_stream_animation_depth.py:563 on a 384x384 image batch and returning inverse
depth `[B, 384, 384]`.

**Pinned to an independent implementation, not to the reference's own code.**  The MiDaS repository (un-vendored git submodule
`live2diff/MiDaS`, commit unknown) and the `timm` backbone it builds on (`vit_base_resnet50_384`) are neither under
/root/reference nor installed here, and the reference holds no test vector for them.  What is restated is the published
DPT-Hybrid architecture (Ranftl et al., "Vision Transformers for Dense Prediction", MiDaS v3 `DPTDepthModel(backbone=
"vitb_rn50_384", non_negative=True)`), and it is CHECKED against a second, published implementation of that architecture:
Hugging Face `transformers` `DPTForDepthEstimation(DPTConfig(is_hybrid=True))` (the class the converted MiDaS `dpt_hybrid`
checkpoint runs on) on the same key-hashed weights -- 12 stage taps and the depth map at 128^2 and 384^2 agree to rel-L2
<= 1e-4 in fp32 (tests/golden/gen_golden_midas.py -> midas_hf.npz, tests/test_midas_cpu.py).  Relative to the REFERENCE the
row therefore stays "parity unpinned" (SURVEY 8c); relative to the published architecture it is pinned.

  backbone   ResNetV2-50 stem + stages (3, 4, 9 bottlenecks; weight-standardised convs with TF-"SAME" padding, GroupNorm(32))
             -> 1x1 projection to 768 -> 24 x 24 patch tokens + class token + position embedding -> 12 ViT-B blocks
  taps       ResNet stage 0 (256 ch, 96^2), stage 1 (512 ch, 48^2), ViT blocks 8 and 11 (readout "project": the class token
             is concatenated to every patch token and projected back to 768 with GELU), 1x1 conv; tap 4 also 3x3 stride 2
  decoder    four 3x3 "layer_rn" convs to 256 channels, four feature-fusion blocks (two pre-activation residual conv units,
             bilinear x2 with align_corners=True, 1x1 conv), head: conv3x3 256->128, bilinear x2, conv3x3 128->32, ReLU,
             conv1x1 32->1, ReLU

State-dict keys follow MiDaS / timm naming (`pretrained.model.*`, `pretrained.act_postprocess*`, `scratch.*`) so that real
DPT-Hybrid weights drop in.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

STAGES = (3, 4, 9)
STAGE_CH = (256, 512, 1024)
VIT_DIM, VIT_DEPTH, VIT_HEADS, VIT_MLP = 768, 12, 12, 3072
HOOKS = (8, 11)                  # ViT blocks tapped for layers 3 and 4 (MiDaS hooks [0, 1, 8, 11]: 0 / 1 are the ResNet stages)
FEATURES = 256
GN_GROUPS = 32


def midas_param_spec(img: int = 384) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    bb = "pretrained.model.patch_embed.backbone."
    s[bb + "stem.conv.weight"] = (64, 3, 7, 7)
    s[bb + "stem.norm.weight"] = (64,)
    s[bb + "stem.norm.bias"] = (64,)
    cin = 64
    for si, (nb, cout) in enumerate(zip(STAGES, STAGE_CH)):
        mid = cout // 4
        for bi in range(nb):
            p = bb + f"stages.{si}.blocks.{bi}."
            if bi == 0:
                s[p + "downsample.conv.weight"] = (cout, cin, 1, 1)
                s[p + "downsample.norm.weight"] = (cout,)
                s[p + "downsample.norm.bias"] = (cout,)
            s[p + "conv1.weight"] = (mid, cin, 1, 1)
            s[p + "norm1.weight"] = (mid,)
            s[p + "norm1.bias"] = (mid,)
            s[p + "conv2.weight"] = (mid, mid, 3, 3)
            s[p + "norm2.weight"] = (mid,)
            s[p + "norm2.bias"] = (mid,)
            s[p + "conv3.weight"] = (cout, mid, 1, 1)
            s[p + "norm3.weight"] = (cout,)
            s[p + "norm3.bias"] = (cout,)
            cin = cout
    m = "pretrained.model."
    s[m + "patch_embed.proj.weight"] = (VIT_DIM, STAGE_CH[-1], 1, 1)
    s[m + "patch_embed.proj.bias"] = (VIT_DIM,)
    s[m + "cls_token"] = (1, 1, VIT_DIM)
    s[m + "pos_embed"] = (1, (img // 16) ** 2 + 1, VIT_DIM)
    for i in range(VIT_DEPTH):
        p = m + f"blocks.{i}."
        for n, shp in (("norm1.weight", (VIT_DIM,)), ("norm1.bias", (VIT_DIM,)), ("attn.qkv.weight", (3 * VIT_DIM, VIT_DIM)),
                       ("attn.qkv.bias", (3 * VIT_DIM,)), ("attn.proj.weight", (VIT_DIM, VIT_DIM)), ("attn.proj.bias", (VIT_DIM,)),
                       ("norm2.weight", (VIT_DIM,)), ("norm2.bias", (VIT_DIM,)), ("mlp.fc1.weight", (VIT_MLP, VIT_DIM)),
                       ("mlp.fc1.bias", (VIT_MLP,)), ("mlp.fc2.weight", (VIT_DIM, VIT_MLP)), ("mlp.fc2.bias", (VIT_DIM,))):
            s[p + n] = shp
    for k in (3, 4):
        p = f"pretrained.act_postprocess{k}."
        s[p + "0.project.0.weight"] = (VIT_DIM, 2 * VIT_DIM)
        s[p + "0.project.0.bias"] = (VIT_DIM,)
        s[p + "3.weight"] = (VIT_DIM, VIT_DIM, 1, 1)
        s[p + "3.bias"] = (VIT_DIM,)
    s["pretrained.act_postprocess4.4.weight"] = (VIT_DIM, VIT_DIM, 3, 3)
    s["pretrained.act_postprocess4.4.bias"] = (VIT_DIM,)
    for k, c in zip((1, 2, 3, 4), (256, 512, VIT_DIM, VIT_DIM)):
        s[f"scratch.layer{k}_rn.weight"] = (FEATURES, c, 3, 3)
    for k in (1, 2, 3, 4):
        for u in (1, 2):
            if k == 4 and u == 1:
                pass                      # refinenet4 gets a single input: resConfUnit1 exists in the checkpoint but is never used
            for c in (1, 2):
                s[f"scratch.refinenet{k}.resConfUnit{u}.conv{c}.weight"] = (FEATURES, FEATURES, 3, 3)
                s[f"scratch.refinenet{k}.resConfUnit{u}.conv{c}.bias"] = (FEATURES,)
        s[f"scratch.refinenet{k}.out_conv.weight"] = (FEATURES, FEATURES, 1, 1)
        s[f"scratch.refinenet{k}.out_conv.bias"] = (FEATURES,)
    s["scratch.output_conv.0.weight"] = (FEATURES // 2, FEATURES, 3, 3)
    s["scratch.output_conv.0.bias"] = (FEATURES // 2,)
    s["scratch.output_conv.2.weight"] = (32, FEATURES // 2, 3, 3)
    s["scratch.output_conv.2.bias"] = (32,)
    s["scratch.output_conv.4.weight"] = (1, 32, 1, 1)
    s["scratch.output_conv.4.bias"] = (1,)
    return s


# ----------------------------------------------------------------------------- building blocks
_ROUND = [lambda t: t]


class fp16_activations:
    """Context manager: round the output of every convolution and GroupNorm of the backbone to fp16 (storage emulation of
    an op-by-op fp16 graph, which is how the reference runs the detector).  Tests use the distance between this and the
    fp32 forward as the noise floor that an fp16 implementation is allowed."""

    def __enter__(self):
        _ROUND.append(lambda t: t.half().float())

    def __exit__(self, *a):
        _ROUND.pop()


def same_pad(n: int, k: int, s: int) -> Tuple[int, int]:
    """TF-"SAME" padding of one axis (timm `pad_same`): total = max((ceil(n/s) - 1) * s + k - n, 0), low = total // 2."""
    total = max((math.ceil(n / s) - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def std_conv_same(x, w, stride=1, eps=1e-8):
    """timm StdConv2dSame: weights standardised per output channel (biased variance, eps inside the sqrt), SAME padding."""
    wf = w.float()
    mean = wf.mean(dim=(1, 2, 3), keepdim=True)
    var = wf.var(dim=(1, 2, 3), keepdim=True, unbiased=False)
    ws = (wf - mean) / torch.sqrt(var + eps)
    k = w.shape[-1]
    pt, pb = same_pad(x.shape[-2], k, stride)
    pl, pr = same_pad(x.shape[-1], k, stride)
    return _ROUND[-1](F.conv2d(F.pad(x, (pl, pr, pt, pb)), ws, None, stride=stride))


def gn(x, sd, p, relu=True):
    y = _ROUND[-1](F.group_norm(x, GN_GROUPS, sd[p + "weight"].float(), sd[p + "bias"].float(), 1e-5))
    return F.relu(y) if relu else y


def bottleneck(x, sd, p, stride):
    """timm ResNetV2 `Bottleneck` (non pre-activation): conv-norm-act x2, conv-norm, + shortcut, act."""
    sc = x
    if (p + "downsample.conv.weight") in sd:
        sc = gn(std_conv_same(x, sd[p + "downsample.conv.weight"], stride), sd, p + "downsample.norm.", relu=False)
    y = gn(std_conv_same(x, sd[p + "conv1.weight"]), sd, p + "norm1.")
    y = gn(std_conv_same(y, sd[p + "conv2.weight"], stride), sd, p + "norm2.")
    y = gn(std_conv_same(y, sd[p + "conv3.weight"]), sd, p + "norm3.", relu=False)
    return F.relu(y + sc)


def vit_block(x, sd, p):
    B, T, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"].float(), sd[p + "norm1.bias"].float(), 1e-6)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"].float(), sd[p + "attn.qkv.bias"].float()).reshape(B, T, 3, VIT_HEADS, C // VIT_HEADS)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    att = torch.softmax(q @ k.transpose(-1, -2) * (C // VIT_HEADS) ** -0.5, dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, T, C)
    x = x + F.linear(o, sd[p + "attn.proj.weight"].float(), sd[p + "attn.proj.bias"].float())
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"].float(), sd[p + "norm2.bias"].float(), 1e-6)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"].float(), sd[p + "mlp.fc1.bias"].float()))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"].float(), sd[p + "mlp.fc2.bias"].float())


def readout_project(tokens, sd, p, g):
    """MiDaS ProjectReadout(start_index=1) + Transpose + Unflatten: [B, 1+g*g, C] -> [B, C, g, g]."""
    patches, cls = tokens[:, 1:], tokens[:, :1]
    f = torch.cat([patches, cls.expand_as(patches)], dim=-1)
    f = F.gelu(F.linear(f, sd[p + "0.project.0.weight"].float(), sd[p + "0.project.0.bias"].float()))
    return f.transpose(1, 2).reshape(f.shape[0], -1, g, g)


def rcu(x, sd, p):
    """ResidualConvUnit_custom (bn=False, activation=ReLU): conv2(relu(conv1(relu(x)))) + x."""
    y = F.conv2d(F.relu(x), sd[p + "conv1.weight"].float(), sd[p + "conv1.bias"].float(), padding=1)
    y = F.conv2d(F.relu(y), sd[p + "conv2.weight"].float(), sd[p + "conv2.bias"].float(), padding=1)
    return y + x


def fusion(sd, k, x, skip=None):
    """FeatureFusionBlock_custom (deconv=False, bn=False, expand=False, align_corners=True)."""
    p = f"scratch.refinenet{k}."
    out = x
    if skip is not None:
        out = out + rcu(skip, sd, p + "resConfUnit1.")
    out = rcu(out, sd, p + "resConfUnit2.")
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(out, sd[p + "out_conv.weight"].float(), sd[p + "out_conv.bias"].float())


def midas_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], taps: dict = None) -> torch.Tensor:
    """x [B,3,H,W] (H, W multiples of 32; 384 in the reference) -> inverse depth [B,H,W] >= 0.  `taps` (optional dict) receives
    named intermediates for stage-by-stage comparison."""
    x = x.float()
    bb = "pretrained.model.patch_embed.backbone."
    t = taps if taps is not None else {}
    y = gn(std_conv_same(x, sd[bb + "stem.conv.weight"], 2), sd, bb + "stem.norm.")
    pt, pb = same_pad(y.shape[-2], 3, 2)
    pl, pr = same_pad(y.shape[-1], 3, 2)
    y = F.max_pool2d(F.pad(y, (pl, pr, pt, pb), value=float("-inf")), 3, 2)
    t["stem"] = y
    feats = []
    for si, nb in enumerate(STAGES):
        for bi in range(nb):
            y = bottleneck(y, sd, bb + f"stages.{si}.blocks.{bi}.", 2 if (bi == 0 and si > 0) else 1)
        feats.append(y)
        t[f"stage{si}"] = y
    m = "pretrained.model."
    tok = F.conv2d(y, sd[m + "patch_embed.proj.weight"].float(), sd[m + "patch_embed.proj.bias"].float())
    B, C, gh, gw = tok.shape
    tok = tok.flatten(2).transpose(1, 2)
    tok = torch.cat([sd[m + "cls_token"].float().expand(B, -1, -1), tok], dim=1) + sd[m + "pos_embed"].float()
    hooked = {}
    for i in range(VIT_DEPTH):
        tok = vit_block(tok, sd, m + f"blocks.{i}.")
        if i in HOOKS:
            hooked[i] = tok
            t[f"vit{i}"] = tok
    l1, l2 = feats[0], feats[1]
    p3, p4 = "pretrained.act_postprocess3.", "pretrained.act_postprocess4."
    l3 = F.conv2d(readout_project(hooked[HOOKS[0]], sd, p3, gh), sd[p3 + "3.weight"].float(), sd[p3 + "3.bias"].float())
    l4 = F.conv2d(readout_project(hooked[HOOKS[1]], sd, p4, gh), sd[p4 + "3.weight"].float(), sd[p4 + "3.bias"].float())
    l4 = F.conv2d(l4, sd[p4 + "4.weight"].float(), sd[p4 + "4.bias"].float(), stride=2, padding=1)
    t["l3"], t["l4"] = l3, l4
    r1, r2, r3, r4 = (F.conv2d(l, sd[f"scratch.layer{k}_rn.weight"].float(), None, padding=1) for k, l in ((1, l1), (2, l2), (3, l3), (4, l4)))
    path = fusion(sd, 4, r4)
    t["path4"] = path
    path = fusion(sd, 3, path, r3)
    t["path3"] = path
    path = fusion(sd, 2, path, r2)
    t["path2"] = path
    path = fusion(sd, 1, path, r1)
    t["path1"] = path
    o = F.conv2d(path, sd["scratch.output_conv.0.weight"].float(), sd["scratch.output_conv.0.bias"].float(), padding=1)
    o = F.interpolate(o, scale_factor=2, mode="bilinear", align_corners=True)
    o = F.relu(F.conv2d(o, sd["scratch.output_conv.2.weight"].float(), sd["scratch.output_conv.2.bias"].float(), padding=1))
    o = F.relu(F.conv2d(o, sd["scratch.output_conv.4.weight"].float(), sd["scratch.output_conv.4.bias"].float()))
    return o[:, 0]
