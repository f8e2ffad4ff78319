"""CPU tests (-m "not gpu") of weight ingestion against fixtures produced by the REFERENCE's own converter
functions (tests/golden/gen_golden_convert.py): the LDM -> diffusers key layout of an SD-1.5 DreamBooth checkpoint, and a
kohya LoRA merged by `convert_lora_model_level` into the reference's streaming UNet."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ldm_key_layout_matches_reference_converter():
    from live2diff_amd.config import sd15_config
    from live2diff_amd.convert import convert_ldm_unet_checkpoint
    from live2diff_amd.weights import unet_param_spec
    mapping = json.load(open(os.path.join(GOLDEN, "convert_ldm_keys.json")))
    assert len(mapping) == 686                                       # SD-1.5 UNet: 686 tensors
    ckpt = {k: torch.tensor([float(i)]) for i, k in enumerate(mapping)}
    ckpt["first_stage_model.decoder.conv_in.weight"] = torch.zeros(1)           # VAE / text-encoder entries are ignored
    out = convert_ldm_unet_checkpoint(ckpt, sd15_config())
    got = {k: out_k for out_k, v in out.items() for k in [list(mapping)[int(v.item())]]}
    assert got == mapping
    spec = unet_param_spec(sd15_config())
    assert set(out) == {k for k in spec if "motion_modules" not in k and "flow_conv_in" not in k}   # exactly the spatial weights
    with pytest.raises(KeyError):
        convert_ldm_unet_checkpoint({"model.diffusion_model.input_blocks.99.0.in_layers.0.weight": torch.zeros(1)}, sd15_config())


def test_lora_merge_matches_reference():
    from live2diff_amd.config import tiny_config
    from live2diff_amd.convert import merge_lora
    from live2diff_amd.weights import _fill, unet_param_spec
    g = dict(np.load(os.path.join(GOLDEN, "convert_lora.npz")))
    lora = {k[len("lora::"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("lora::")}
    after = {k[len("after::"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("after::")}
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    sd = {k: _fill(k, shp, 1.0) for k, shp in unet_param_spec(cfg).items()}
    before = {k: v.clone() for k, v in sd.items()}
    touched = merge_lora(sd, lora, alpha=float(g["alpha"]), strict=True)
    assert sorted(touched) == sorted(after) and len(after) == 14
    for k in sd:
        if k in after:
            assert torch.allclose(sd[k], after[k], rtol=0, atol=1e-6), k
            assert not torch.equal(sd[k], before[k])
        else:
            assert torch.equal(sd[k], before[k]), k
    # a module the UNet does not have: skipped unless strict
    bogus = {"lora_unet_down_blocks_9_foo.lora_down.weight": torch.zeros(4, 8), "lora_unet_down_blocks_9_foo.lora_up.weight": torch.zeros(8, 4)}
    assert merge_lora(sd, bogus, 1.0) == []
    with pytest.raises(KeyError):
        merge_lora(sd, bogus, 1.0, strict=True)
    # text-encoder pairs are not the UNet's business
    assert merge_lora(sd, {"lora_te_text_model_encoder_layers_0_self_attn_k_proj.lora_down.weight": torch.zeros(4, 8),
                           "lora_te_text_model_encoder_layers_0_self_attn_k_proj.lora_up.weight": torch.zeros(8, 4)}, 1.0, strict=True) == []


def test_build_state_dict_feeds_the_packed_cache(tmp_path):
    """DreamBooth over the base weights (motion modules untouched), LoRA on top, then the packing pass + cache file."""
    from live2diff_amd import _lib
    from live2diff_amd.config import tiny_config
    from live2diff_amd.convert import LDM_UNET_PREFIX, build_state_dict, ldm_unet_key_map
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    base = random_state_dict(cfg, dtype=torch.float16)
    inv = {v: k for k, v in ldm_unet_key_map(cfg).items()}
    db = {}
    for k, v in base.items():
        if "motion_modules" in k or "flow_conv_in" in k:
            continue
        mod, leaf = k.rsplit(".", 1)
        pre = max((p for p in inv if mod == p or mod.startswith(p + ".")), key=len)
        db[LDM_UNET_PREFIX + inv[pre] + mod[len(pre):] + "." + leaf] = (v.float() * 0.5).half()
    name = "down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q"
    lora = {f"lora_unet_{name}.lora_down.weight": torch.ones(2, 64) * 0.01, f"lora_unet_{name}.lora_up.weight": torch.ones(64, 2)}
    sd = build_state_dict(base, cfg, dreambooth=db, loras=[(lora, 0.5)])
    k = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"
    assert torch.allclose(sd[k].float(), base[k].float() * 0.5 + 0.5 * 0.02, atol=2e-3)
    assert torch.equal(sd["mid_block.resnets.0.conv1.weight"], (base["mid_block.resnets.0.conv1.weight"].float() * 0.5).half())
    mk = next(x for x in base if "motion_modules" in x and x.endswith("to_q.weight"))
    assert torch.equal(sd[mk], base[mk])
    _lib.set_dry_run(True)
    try:
        u = HipStreamingUNet(sd, cfg, 16, 16, 2, device="cpu")
        path = tmp_path / (HipStreamingUNet.packed_cache_name("sd15-db", "lcm", cfg.window_size, {"style.safetensors": 0.5}) + ".safetensors")
        u.save_packed(path)
        v = HipStreamingUNet(path, cfg, 16, 16, 2, device="cpu")
        assert all(torch.equal(u.W[x], v.W[x]) for x in u.W)
    finally:
        _lib.set_dry_run(False)
    with pytest.raises(ValueError):
        bad = dict(db)
        kk = next(iter(bad))
        bad[kk] = torch.zeros(3, 3)
        build_state_dict(base, cfg, dreambooth=bad)
