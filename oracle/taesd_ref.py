"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain PyTorch, fp32) of the tiny VAE the reference puts either side of the UNet in every frame
(SURVEY.md section 8f row F1): `AutoencoderTiny.from_pretrained("madebyollin/taesd")`, swapped into `stream.vae` at
live2diff/utils/wrapper.py:468-470 and called from the pipeline at pipeline_stream_animation_depth.py:526 (encode image),
:569 (encode depth map), :541 (decode).

**Parity unpinned.**  `AutoencoderTiny` lives in diffusers==0.25.0 (pin: reference setup.py:5), which is neither under
/root/reference nor installed here, and the reference holds no test vector for it.  What is restated is the documented 0.25.0
topology (`EncoderTiny` / `DecoderTiny` / `AutoencoderTinyBlock` in diffusers/models/vae.py and autoencoder_tiny.py):

  AutoencoderTinyBlock(c):  relu( conv3(relu(conv3(relu(conv3(x))))) + x )            (skip = identity: in == out channels)
  EncoderTiny:  x -> (x + 1) / 2 -> conv3(3->64) -> block
                -> [conv3 stride 2 (no bias) -> block x 3] x 3 -> conv3(64->4)
  DecoderTiny:  z -> tanh(z / 3) * 3 -> conv3(4->64) -> relu
                -> [block x 3 -> nearest x2 -> conv3 (no bias)] x 3 -> block -> conv3(64->3) -> * 2 - 1
  encode() returns the encoder output as `.latents`; decode() returns the decoder output; `config.scaling_factor` = 1.0.

State-dict keys are diffusers' (`encoder.layers.N...`, `decoder.layers.N...`), so real TAESD weights drop in.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

ENC_BLOCKS = (1, 3, 3, 3)
DEC_BLOCKS = (3, 3, 3, 1)
WIDTH = 64


def taesd_layer_plan(width: int = WIDTH):
    """[(kind, layer_index, cin, cout)] for encoder and decoder in `nn.Sequential` index order.
    kinds: conv (bias), conv_s2 (stride 2, no bias), conv_nb (no bias), block, up (nearest x2), relu."""
    enc, i = [], 0
    for lvl, nb in enumerate(ENC_BLOCKS):
        enc.append(("conv", i, 3, width) if lvl == 0 else ("conv_s2", i, width, width))
        i += 1
        for _ in range(nb):
            enc.append(("block", i, width, width))
            i += 1
    enc.append(("conv", i, width, 4))
    dec = [("conv", 0, 4, width), ("relu", 1, width, width)]
    i = 2
    for lvl, nb in enumerate(DEC_BLOCKS):
        final = lvl == len(DEC_BLOCKS) - 1
        for _ in range(nb):
            dec.append(("block", i, width, width))
            i += 1
        if not final:
            dec.append(("up", i, width, width))
            i += 1
        dec.append(("conv", i, width, 3) if final else ("conv_nb", i, width, width))
        i += 1
    return enc, dec


def taesd_param_spec(width: int = WIDTH) -> "OrderedDict[str, Tuple[int, ...]]":
    spec = OrderedDict()
    for side, plan in zip(("encoder", "decoder"), taesd_layer_plan(width)):
        for kind, i, cin, cout in plan:
            p = f"{side}.layers.{i}."
            if kind in ("conv", "conv_s2", "conv_nb"):
                spec[p + "weight"] = (cout, cin, 3, 3)
                if kind == "conv":
                    spec[p + "bias"] = (cout,)
            elif kind == "block":
                for j in (0, 2, 4):
                    spec[p + f"conv.{j}.weight"] = (cout, cin, 3, 3)
                    spec[p + f"conv.{j}.bias"] = (cout,)
    return spec


def _block(x, sd, p):
    h = x
    for j in (0, 2, 4):
        h = F.conv2d(h, sd[p + f"conv.{j}.weight"].float(), sd[p + f"conv.{j}.bias"].float(), padding=1)
        if j != 4:
            h = F.relu(h)
    return F.relu(h + x)


def _run(x, sd: Dict[str, torch.Tensor], side: str, plan):
    for kind, i, _cin, _cout in plan:
        p = f"{side}.layers.{i}."
        if kind == "conv":
            x = F.conv2d(x, sd[p + "weight"].float(), sd[p + "bias"].float(), padding=1)
        elif kind == "conv_nb":
            x = F.conv2d(x, sd[p + "weight"].float(), None, padding=1)
        elif kind == "conv_s2":
            x = F.conv2d(x, sd[p + "weight"].float(), None, stride=2, padding=1)
        elif kind == "block":
            x = _block(x, sd, p)
        elif kind == "up":
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        elif kind == "relu":
            x = F.relu(x)
    return x


def taesd_encode(x: torch.Tensor, sd: Dict[str, torch.Tensor], width: int = WIDTH) -> torch.Tensor:
    """x [B,3,H,W] in [-1,1] -> latents [B,4,H/8,W/8] (EncoderTiny.forward)."""
    return _run((x.float() + 1.0) / 2.0, sd, "encoder", taesd_layer_plan(width)[0])


def taesd_decode(z: torch.Tensor, sd: Dict[str, torch.Tensor], width: int = WIDTH) -> torch.Tensor:
    """z [B,4,h,w] -> image [B,3,8h,8w] (DecoderTiny.forward)."""
    x = torch.tanh(z.float() / 3.0) * 3.0
    return _run(x, sd, "decoder", taesd_layer_plan(width)[1]) * 2.0 - 1.0


def depth_glue(depth_map: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """The arithmetic between the depth detector and the VAE in the reference's encode_depth
    (pipeline_stream_animation_depth.py:560-567): min-max normalise over the WHOLE batch tensor, repeat to 3 channels,
    map to [-1,1], bilinear resize (align_corners=False) to the image size.  depth_map [B,Hd,Wd] -> [B,3,h,w]."""
    d = depth_map.float()
    dn = (d - d.min()) / (d.max() - d.min())
    dn = dn[:, None].repeat(1, 3, 1, 1) * 2 - 1
    return F.interpolate(dn, (h, w), mode="bilinear", align_corners=False)


def resize_bilinear(x: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """F.interpolate(x, (h, w), mode="bilinear", align_corners=False) (reference :553, the 384x384 depth-detector input)."""
    return F.interpolate(x.float(), (h, w), mode="bilinear", align_corners=False)
