"""HipTinyVAE / HipDepthGlue -- the objects either side of the UNet in every frame.

Boundary (reference): `stream.vae`, an `AutoencoderTiny` ("madebyollin/taesd") installed at live2diff/utils/wrapper.py:468-470
and swappable exactly like the UNet (TensorRT twin: acceleration/tensorrt/engine.py:71-109, swap at wrapper.py:616-624).  The
pipeline calls `vae.encode(x)` -> object with `.latents` (pipeline_stream_animation_depth.py:526 image, :569 depth map) and
`vae.decode(z, return_dict=False)[0]` (:541), and reads `vae.config.scaling_factor` and `vae.dtype`.

The network (diffusers 0.25.0 `EncoderTiny` / `DecoderTiny`, 2.45 M parameters, third-party: parity unpinned, see
oracle/taesd_ref.py) is 3x3 convolutions with ReLU and identity-skip blocks, so it runs on the UNet's implicit-GEMM kernel:
  * channels-last fp16 activations `[B*H*W, 64]`; the 3-channel image / 4-channel latent are padded to 8 channels by the
    layout kernel, which also applies the network's input map ((x+1)/2, tanh(z/3)*3) and, on the way out, 2x-1;
  * conv + bias + ReLU, and conv + bias + skip + ReLU (`relu(conv(x) + x)`, epilogue mode 4) are single launches; the
    nearest-x2 upsample is folded into the next conv's gather, the stride-2 convs are the gather's stride;
  * one static plan per (batch, H, W) and direction, replayed through the C ABI like the UNet's.
`HipDepthGlue` is the arithmetic between the depth detector and the VAE (reference :553, :560-567) as three ops.
"""
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Tuple

import torch

from . import _lib, ops
from .unet_hip import _Arena

ENC_BLOCKS = (1, 3, 3, 3)
DEC_BLOCKS = (3, 3, 3, 1)


def taesd_param_spec(width: int = 64) -> "OrderedDict[str, Tuple[int, ...]]":
    """diffusers `AutoencoderTiny.state_dict()` names -> shapes (encoder.layers.N / decoder.layers.N in nn.Sequential order)."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(p, cin, cout, bias=True):
        spec[p + "weight"] = (cout, cin, 3, 3)
        if bias:
            spec[p + "bias"] = (cout,)

    def block(p, c):
        for j in (0, 2, 4):
            conv(p + f"conv.{j}.", c, c)

    i = 0
    for lvl, nb in enumerate(ENC_BLOCKS):
        conv(f"encoder.layers.{i}.", 3 if lvl == 0 else width, width, bias=(lvl == 0))
        i += 1
        for _ in range(nb):
            block(f"encoder.layers.{i}.", width)
            i += 1
    conv(f"encoder.layers.{i}.", width, 4)
    conv("decoder.layers.0.", 4, width)
    i = 2                                            # layers.1 is the ReLU
    for lvl, nb in enumerate(DEC_BLOCKS):
        final = lvl == len(DEC_BLOCKS) - 1
        for _ in range(nb):
            block(f"decoder.layers.{i}.", width)
            i += 1
        if not final:
            i += 1                                   # nn.Upsample
        conv(f"decoder.layers.{i}.", width, 3 if final else width, bias=final)
        i += 1
    return spec


def random_taesd_state_dict(width: int = 64, dtype=torch.float16, device="cpu") -> Dict[str, torch.Tensor]:
    """Key-hashed deterministic weights (same recipe as weights.random_state_dict: seed = crc32(key))."""
    from .weights import _fill
    return OrderedDict((k, _fill("taesd." + k, shp, 1.0).to(device=device, dtype=dtype)) for k, shp in taesd_param_spec(width).items())


class _Out:
    """`.latents` / `.sample` holder (diffusers AutoencoderTinyOutput / DecoderOutput)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class HipTinyVAE:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", width: int = 64):
        self.device = torch.device(device)
        self.dtype = torch.float16
        self.width = width
        self.config = SimpleNamespace(scaling_factor=1.0, latent_channels=4, in_channels=3, out_channels=3)
        self.device_name = "dry-run" if ops.DRY_RUN else _lib.device_name()
        spec = taesd_param_spec(width)
        missing = [k for k in spec if k not in state_dict]
        if missing:
            raise KeyError(f"TAESD state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
        self.W = {}
        for k, shp in spec.items():
            t = state_dict[k].to(self.device)
            assert tuple(t.shape) == tuple(shp), (k, tuple(t.shape), shp)
            if k.endswith("weight"):
                self.W[k] = ops.pack_conv3x3(t)       # [Cout, 9 * 64]: Cin 3 / 4 / 64 all pad to one 64-wide K slice per tap
            else:
                self.W[k] = ops.f32(t)
        self._plans = {}

    # duck-typed members of the reference's vae
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ plans
    def _build(self, side: str, B: int, H: int, W_: int):
        """side 'enc': static input [B,3,H,W] -> static output [B,4,H/8,W/8]; 'dec': [B,4,H,W] latent -> [B,3,8H,8W]."""
        dev, W = self.device, self.W
        ar = _Arena(dev)
        pl = _lib.OpList()
        cin0 = 3 if side == "enc" else 4
        st = SimpleNamespace(pl=pl, arena=ar)
        st.inp = torch.zeros(B, cin0, H * W_, dtype=torch.float16, device=dev)
        h, w = H, W_
        x = ar.alloc(B * h * w * 8)
        if side == "enc":
            pl.append(*ops.nchw_to_nhwc(st.inp, x, B=B, C=3, HW=h * w, Cpad=8, mode=ops.MAP_ADD_SCALE, a=0.5, b=1.0))
        else:
            pl.append(*ops.nchw_to_nhwc(st.inp, x, B=B, C=4, HW=h * w, Cpad=8, mode=ops.MAP_TANH3))
        C = 8

        def conv(x, C, name, h, w, stride=1, ups=0, epi=0, res=None):
            wt = W[name + "weight"]
            cout = wt.shape[0]
            ho, wo = (h * 2, w * 2) if ups else (((h - 1) // 2 + 1, (w - 1) // 2 + 1) if stride == 2 else (h, w))
            ldo = max(4, cout)                                   # 3-channel image rows are stored 4 wide
            out = ar.alloc(B * ho * wo * ldo)
            M, Kp = B * ho * wo, wt.shape[1]
            patch = ops.pconv_patch(B, h, w, cout, C) if (stride == 1 and not ups and Kp == 9 * C) else None
            if patch is not None:
                # 64 -> 64 convs at the upper resolutions: the activation patch stays in LDS for all nine taps (csrc/pconv.hip)
                pl.append(*ops.pconv(x, wt, out, B=B, H=h, W=w, C1=C, ldx1=C, CinP=C, Nout=cout, ldo=ldo, patch=patch,
                                     bias=W.get(name + "bias"), res=res, ldr=(cout if res is not None else 0), epi=epi))
                return out, cout, ho, wo
            tile, S, variant = ops.igemm_schedule(M, cout, Kp, 1, epi, 9)
            pl.append(*ops.igemm(x, wt, out, M=M, Nout=cout, C1=C, ldx1=C, CinP=Kp // 9, ldo=ldo, bias=W.get(name + "bias"),
                                 res=res, ldr=(cout if res is not None else 0), taps=9, B=B, Hin=h, Win=w, Hout=ho, Wout=wo,
                                 stride=stride, ups=ups, epi=epi, splitk=1, tile=tile, variant=variant))
            return out, cout, ho, wo

        def block(x, C, name, h, w):
            a, _, _, _ = conv(x, C, name + "conv.0.", h, w, epi=3)
            b, _, _, _ = conv(a, C, name + "conv.2.", h, w, epi=3)
            ar.release(a)
            o, _, _, _ = conv(b, C, name + "conv.4.", h, w, epi=4, res=x)      # relu(conv(x) + x)
            ar.release(b)
            ar.release(x)
            return o

        i = 0
        if side == "enc":
            for lvl, nb in enumerate(ENC_BLOCKS):
                y, C, h, w = conv(x, C, f"encoder.layers.{i}.", h, w, stride=(1 if lvl == 0 else 2))
                ar.release(x)
                x = y
                i += 1
                for _ in range(nb):
                    x = block(x, C, f"encoder.layers.{i}.", h, w)
                    i += 1
            y, C, h, w = conv(x, C, f"encoder.layers.{i}.", h, w)
            ar.release(x)
            st.out = torch.zeros(B, 4, h * w, dtype=torch.float16, device=dev)
            pl.append(*ops.nhwc_to_nchw(y, st.out, B=B, C=4, HW=h * w, ld=4))
            st.out_shape = (B, 4, h, w)
        else:
            y, C, h, w = conv(x, C, "decoder.layers.0.", h, w, epi=3)
            ar.release(x)
            x = y
            i = 2
            for lvl, nb in enumerate(DEC_BLOCKS):
                final = lvl == len(DEC_BLOCKS) - 1
                for _ in range(nb):
                    x = block(x, C, f"decoder.layers.{i}.", h, w)
                    i += 1
                if not final:
                    i += 1
                y, C, h, w = conv(x, C, f"decoder.layers.{i}.", h, w, ups=(0 if final else 1))
                ar.release(x)
                x = y
                i += 1
            st.out = torch.zeros(B, 3, h * w, dtype=torch.float16, device=dev)
            pl.append(*ops.nhwc_to_nchw(x, st.out, B=B, C=3, HW=h * w, ld=4, mode=ops.MAP_SCALE_ADD, a=2.0, b=-1.0))
            st.out_shape = (B, 3, h, w)
        st.arena_bytes = ar.nbytes()
        return st

    def _plan(self, side, B, H, W_):
        key = (side, B, H, W_)
        st = self._plans.get(key)
        if st is None:
            if H % 8 or W_ % 8:
                if side == "enc":
                    raise ValueError(f"image size {H}x{W_} must be divisible by 8")
            st = self._plans[key] = self._build(side, B, H, W_)
        return st

    # ------------------------------------------------------------------ the boundary calls
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x [B,3,H,W] in [-1,1] -> `.latents` [B,4,H/8,W/8] (a view of the plan's static output, valid until the next encode
        of the same shape)."""
        B, C, H, W_ = x.shape
        if C != 3:
            raise ValueError(f"encode expects [B,3,H,W], got {tuple(x.shape)}")
        st = self._plan("enc", B, H, W_)
        st.inp.copy_(x.reshape(B, 3, H * W_))
        st.pl.run()
        lat = st.out.view(st.out_shape)
        return _Out(latents=lat) if return_dict else (lat,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, generator=None, return_dict: bool = True):
        B, C, h, w = z.shape
        if C != 4:
            raise ValueError(f"decode expects [B,4,h,w], got {tuple(z.shape)}")
        st = self._plan("dec", B, h, w)
        st.inp.copy_(z.reshape(B, 4, h * w))
        st.pl.run()
        img = st.out.view(st.out_shape)
        return _Out(sample=img) if return_dict else (img,)

    def plan_summary(self):
        return {k: dict(n_ops=len(st.pl), arena_bytes=st.arena_bytes) for k, st in self._plans.items()}


class HipDepthGlue:
    """The depth path's arithmetic around the (caller-owned) depth detector, on the device without host syncs."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self._scratch = torch.zeros(2 * 256, dtype=torch.float32, device=self.device)
        self._mm = torch.zeros(2, dtype=torch.float32, device=self.device)

    @torch.no_grad()
    def resize(self, x: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """F.interpolate(x, (h, w), mode="bilinear", align_corners=False) for [B,C,H,W] fp16 (reference :553)."""
        B, C, H, W_ = x.shape
        x = x.contiguous()
        out = torch.empty(B, C, h, w, dtype=torch.float16, device=x.device)
        ops.run(ops.resize_bilinear(x, out, planes=B * C, Hin=H, Win=W_, Hout=h, Wout=w))
        return out

    @torch.no_grad()
    def normalize_resize(self, depth_map: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """depth_map [B,Hd,Wd] fp16 -> min-max normalised, 3 channels, [-1,1], bilinear-resized [B,3,h,w] (reference :560-567)."""
        B, Hd, Wd = depth_map.shape
        d = depth_map.contiguous()
        out = torch.empty(B, 3, h, w, dtype=torch.float16, device=d.device)
        pl = _lib.OpList()
        pl.append(*ops.minmax(d, self._scratch, self._mm, n=d.numel(), nb=min(256, max(1, d.numel() // 2048))))
        pl.append(*ops.depth_norm_resize(d, self._mm, out, B=B, Hd=Hd, Wd=Wd, H=h, W=w))
        pl.run()
        return out
