"""Pins oracle/midas_ref.py (a restatement of the published DPT-Hybrid architecture: the MiDaS submodule is absent from
/root/reference) against an INDEPENDENT published implementation: Hugging Face `transformers` DPTForDepthEstimation with
`DPTConfig(is_hybrid=True)` (BiT ResNetV2-50 stem + 3/4/9 stages, ViT-B/16, readout "project", neck 256/512/768/768,
fusion 256) -- the class the published `Intel/dpt-hybrid-midas` checkpoint (converted from the MiDaS `dpt_hybrid` weights) runs on.

Run in the build container (needs `transformers`; not needed on the GPU box):  python tests/golden/gen_golden_midas.py
Both sides get the same key-hashed weights (`random_midas_state_dict`, MiDaS / timm key names), moved to the HF module
through the MiDaS -> HF key map of the published conversion (qkv split into query / key / value, refinenet4..1 -> fusion layers
0..3, act_postprocess3/4 -> reassemble layers 2/3, scratch.layerN_rn -> neck.convs, output_conv -> head).  Stored: the input
seeds, the full depth map at 128^2 (fp32), and for 12 stage taps + the depth map at 128^2 and 384^2 the values at 4096
fixed sample positions plus each tap's mean and L2 norm (the full taps at 384^2 would be ~60 MB).  tests/test_midas_cpu.py
compares oracle/midas_ref.py against them (fp32 vs fp32: rel-L2 <= 1e-4)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

TAPS = ("stem", "stage0", "stage1", "stage2", "vit8", "vit11", "l3", "l4", "path4", "path3", "path2", "path1")
NSAMPLE = 4096


def sample_index(numel: int, name: str) -> torch.Tensor:
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    return torch.randint(0, numel, (min(NSAMPLE, numel),), generator=g)


def midas_to_hf(sd, img):
    """MiDaS / timm keyed state dict -> HF DPTForDepthEstimation(is_hybrid=True) keys (the published conversion's map)."""
    out = {}
    bb, hb = "pretrained.model.patch_embed.backbone.", "dpt.embeddings.backbone.bit."
    out[hb + "embedder.convolution.weight"] = sd[bb + "stem.conv.weight"]
    out[hb + "embedder.norm.weight"], out[hb + "embedder.norm.bias"] = sd[bb + "stem.norm.weight"], sd[bb + "stem.norm.bias"]
    for k, v in sd.items():
        if k.startswith(bb + "stages."):
            out[hb + "encoder." + k[len(bb):].replace(".blocks.", ".layers.")] = v
    m = "pretrained.model."
    out["dpt.embeddings.projection.weight"], out["dpt.embeddings.projection.bias"] = sd[m + "patch_embed.proj.weight"], sd[m + "patch_embed.proj.bias"]
    out["dpt.embeddings.cls_token"], out["dpt.embeddings.position_embeddings"] = sd[m + "cls_token"], sd[m + "pos_embed"]
    C = 768
    for i in range(12):
        p, h = m + f"blocks.{i}.", f"dpt.encoder.layer.{i}."
        w, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, nm in enumerate(("query", "key", "value")):
            out[h + f"attention.attention.{nm}.weight"], out[h + f"attention.attention.{nm}.bias"] = w[j * C:(j + 1) * C], b[j * C:(j + 1) * C]
        for a, bname in (("attn.proj", "attention.output.dense"), ("norm1", "layernorm_before"), ("norm2", "layernorm_after"),
                         ("mlp.fc1", "intermediate.dense"), ("mlp.fc2", "output.dense")):
            out[h + bname + ".weight"], out[h + bname + ".bias"] = sd[p + a + ".weight"], sd[p + a + ".bias"]
    for k, li in ((3, 2), (4, 3)):
        p = f"pretrained.act_postprocess{k}."
        out[f"neck.reassemble_stage.readout_projects.{li}.0.weight"], out[f"neck.reassemble_stage.readout_projects.{li}.0.bias"] = sd[p + "0.project.0.weight"], sd[p + "0.project.0.bias"]
        out[f"neck.reassemble_stage.layers.{li}.projection.weight"], out[f"neck.reassemble_stage.layers.{li}.projection.bias"] = sd[p + "3.weight"], sd[p + "3.bias"]
    out["neck.reassemble_stage.layers.3.resize.weight"], out["neck.reassemble_stage.layers.3.resize.bias"] = sd["pretrained.act_postprocess4.4.weight"], sd["pretrained.act_postprocess4.4.bias"]
    for k in (1, 2, 3, 4):
        out[f"neck.convs.{k - 1}.weight"] = sd[f"scratch.layer{k}_rn.weight"]
        f, r = f"neck.fusion_stage.layers.{4 - k}.", f"scratch.refinenet{k}."
        out[f + "projection.weight"], out[f + "projection.bias"] = sd[r + "out_conv.weight"], sd[r + "out_conv.bias"]
        for u in (1, 2):
            for c in (1, 2):
                out[f + f"residual_layer{u}.convolution{c}.weight"] = sd[r + f"resConfUnit{u}.conv{c}.weight"]
                out[f + f"residual_layer{u}.convolution{c}.bias"] = sd[r + f"resConfUnit{u}.conv{c}.bias"]
    for j in (0, 2, 4):
        out[f"head.head.{j}.weight"], out[f"head.head.{j}.bias"] = sd[f"scratch.output_conv.{j}.weight"], sd[f"scratch.output_conv.{j}.bias"]
    return out


def hf_forward(sd, x, img):
    from transformers import DPTConfig, DPTForDepthEstimation
    g = img // 16
    cfg = DPTConfig(is_hybrid=True, image_size=img, patch_size=16, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                    intermediate_size=3072, hidden_act="gelu", layer_norm_eps=1e-6, qkv_bias=True, backbone_out_indices=[2, 5, 8, 11],
                    readout_type="project", reassemble_factors=[4, 2, 1, 0.5], neck_hidden_sizes=[256, 512, 768, 768],
                    fusion_hidden_size=256, head_in_index=-1, add_projection=False, backbone_featmap_shape=[1, 1024, g, g],
                    use_batch_norm_in_fusion_residual=False)
    model = DPTForDepthEstimation(cfg).eval().float()
    hf = midas_to_hf({k: v.float() for k, v in sd.items()}, img)
    ref = model.state_dict()
    extra = {k: ref[k] for k in ("dpt.layernorm.weight", "dpt.layernorm.bias")}      # final ViT norm: not on the depth path
    missing = [k for k in ref if k not in hf and k not in extra]
    unexpected = [k for k in hf if k not in ref]
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    for k in hf:
        assert tuple(hf[k].shape) == tuple(ref[k].shape), (k, tuple(hf[k].shape), tuple(ref[k].shape))
    model.load_state_dict({**hf, **extra}, strict=True)
    taps = {}
    bit = model.dpt.embeddings.backbone.bit
    hooks = [bit.embedder.register_forward_hook(lambda m_, i, o: taps.__setitem__("stem", o))]
    for si in range(3):
        hooks.append(bit.encoder.stages[si].register_forward_hook(lambda m_, i, o, si=si: taps.__setitem__(f"stage{si}", o)))
    for li in (8, 11):
        hooks.append(model.dpt.encoder.layer[li].register_forward_hook(
            lambda m_, i, o, li=li: taps.__setitem__(f"vit{li}", o[0] if isinstance(o, tuple) else o)))
    def h_reassemble(m_, i, o):        # (a hook that returns something replaces the module's output: return None)
        taps["l3"], taps["l4"] = o[2], o[3]

    def h_fusion(m_, i, o):
        for j in range(4):
            taps[f"path{4 - j}"] = o[j]

    hooks.append(model.neck.reassemble_stage.register_forward_hook(h_reassemble))
    hooks.append(model.neck.fusion_stage.register_forward_hook(h_fusion))
    with torch.no_grad():
        depth = model(pixel_values=x).predicted_depth
    for h in hooks:
        h.remove()
    return depth, taps


def main():
    from live2diff_amd.midas_spec import random_midas_state_dict      # (plain torch: no HIP library needed)
    out = {}
    for img, B in ((128, 2), (384, 1)):
        sd = random_midas_state_dict(dtype=torch.float32, img=img)
        g = torch.Generator().manual_seed(900 + img)
        x = torch.randn(B, 3, img, img, generator=g)
        depth, taps = hf_forward(sd, x, img)
        assert set(taps) == set(TAPS), sorted(taps)
        print(img, "depth", tuple(depth.shape), float(depth.mean()), "taps", {k: tuple(v.shape) for k, v in taps.items()})
        for name, t in list(taps.items()) + [("depth", depth)]:
            flat = t.reshape(-1).float()
            idx = sample_index(flat.numel(), f"{img}.{name}")
            out[f"{img}.{name}.samples"] = flat[idx].numpy()
            out[f"{img}.{name}.stats"] = np.array([flat.double().mean().item(), flat.double().norm().item(), flat.numel()], dtype=np.float64)
        if img == 128:
            out["128.depth.full"] = depth.numpy()
    np.savez_compressed(os.path.join(HERE, "midas_hf.npz"), **out)
    print("wrote", os.path.join(HERE, "midas_hf.npz"), os.path.getsize(os.path.join(HERE, "midas_hf.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
