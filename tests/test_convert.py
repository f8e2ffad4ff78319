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


def test_raw_checkpoint_ingestion_motion_ckpt_and_few_step_lora():
    """F4 remainder: the live2diff.ckpt merge (reference pipeline_animatediff_depth.py:281-290) and the few-step (LCM) LoRA
    fuse that every run applies before the PE tables are projected (wrapper.py:451-459), on plain state dicts.  The fuse
    arithmetic is diffusers 0.25.0's (not under /root/reference: parity unpinned) -- checked against the formula."""
    from live2diff_amd.config import tiny_config
    from live2diff_amd.convert import build_state_dict, merge_few_step_lora, merge_lora, merge_motion_checkpoint
    from live2diff_amd.weights import random_state_dict
    cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
    base = random_state_dict(cfg, dtype=torch.float16)
    # --- motion checkpoint: DDP-prefixed keys, a `grid` buffer to drop, wrapped in {"state_dict": ...}
    mkeys = [k for k in base if "motion_modules" in k][:40]
    ck = {"global_step": 7, "state_dict": {"module." + k: (base[k].float() + 1.0).half() for k in mkeys}}
    ck["state_dict"]["module.flow_conv_in.grid"] = torch.zeros(3)
    sd = dict(base)
    assert sorted(merge_motion_checkpoint(sd, ck)) == sorted(mkeys)
    assert all(torch.equal(sd[k], (base[k].float() + 1.0).half()) for k in mkeys)
    assert all(torch.equal(sd[k], base[k]) for k in base if k not in mkeys)
    with pytest.raises(KeyError):
        merge_motion_checkpoint(dict(base), {"module.not_a_parameter.weight": torch.zeros(1)})
    # --- few-step LoRA, three key layouts of the same pairs: kohya (+alpha), diffusers `lora.down`, peft `lora_A`
    lin = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"
    out = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0"
    conv = "down_blocks.0.resnets.0.conv1"
    mot = next(k[: -len(".weight")] for k in base if "motion_modules" in k and k.endswith("attention_blocks.0.to_q.weight"))
    g = torch.Generator().manual_seed(5)
    pairs = {}
    for m in (lin, out, conv, mot):
        w = base[m + ".weight"]
        r = 4
        down = torch.randn(r, w.shape[1], *w.shape[2:], generator=g) * 0.05
        up = torch.randn(w.shape[0], r, *([1] * (w.dim() - 2)), generator=g) * 0.05
        pairs[m] = (up, down)
    alpha = 2.0
    want = {m + ".weight": (base[m + ".weight"].float() + 1.0 * (alpha / 4) *
                            (u.reshape(u.shape[0], -1) @ d.reshape(4, -1)).reshape(base[m + ".weight"].shape)).half()
            for m, (u, d) in pairs.items()}
    kohya, dfs, peft = {}, {}, {}
    for m, (u, d) in pairs.items():
        f = "lora_unet_" + m.replace(".", "_")
        kohya[f + ".lora_down.weight"], kohya[f + ".lora_up.weight"], kohya[f + ".alpha"] = d, u, torch.tensor(alpha)
        dfs["unet." + m + ".lora.down.weight"], dfs["unet." + m + ".lora.up.weight"] = d, u * (alpha / 4)
        peft["unet." + m + ".lora_A.weight"], peft["unet." + m + ".lora_B.weight"] = d, u * (alpha / 4)
    kohya["lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight"] = torch.zeros(4, 8)       # not the UNet's business
    kohya["lora_te_text_model_encoder_layers_0_mlp_fc1.lora_up.weight"] = torch.zeros(8, 4)
    for name, lora in (("kohya", kohya), ("diffusers", dfs), ("peft", peft)):
        sd = dict(base)
        touched = merge_few_step_lora(sd, lora)
        assert sorted(touched) == sorted(want), name
        for k in base:
            assert torch.equal(sd[k], want.get(k, base[k])), (name, k)
    # attention-processor form of the linear pair
    proc = {"unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_q_lora.down.weight": pairs[lin][1],
            "unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_q_lora.up.weight": pairs[lin][0] * (alpha / 4)}
    sd = dict(base)
    assert merge_few_step_lora(sd, proc) == [lin + ".weight"] and torch.equal(sd[lin + ".weight"], want[lin + ".weight"])
    with pytest.raises((KeyError, ValueError)):
        merge_few_step_lora(dict(base), {"unet.nope.lora.down.weight": torch.zeros(4, 8), "unet.nope.lora.up.weight": torch.zeros(8, 4)})
    # --- the whole order from raw checkpoints; the motion-module pair is merged BEFORE packing projects the PE tables
    sd = build_state_dict(base, cfg, motion_ckpt=ck, few_step_lora=kohya)
    k = mot + ".weight"
    assert k in ["%s" % x for x in sd] and not torch.equal(sd[k], base[k])
    # --- a LoRA that resolves against nothing is an error, not a silent no-op; and conv_in merges never touch the caller's tensors
    with pytest.raises(KeyError):
        build_state_dict(base, cfg, loras=[({"lora_unet_down_blocks_9_foo.lora_down.weight": torch.zeros(4, 8),
                                             "lora_unet_down_blocks_9_foo.lora_up.weight": torch.zeros(8, 4)}, 1.0)])
    b32 = {k: v.float() for k, v in base.items()}
    keep = b32["conv_in.weight"].clone()
    ci = {"lora_unet_conv_in.lora_down.weight": torch.ones(2, 4, 3, 3) * 0.1, "lora_unet_conv_in.lora_up.weight": torch.ones(b32["conv_in.weight"].shape[0], 2, 1, 1)}
    s1 = build_state_dict(b32, cfg, loras=[(ci, 1.0)])
    s2 = build_state_dict(b32, cfg, loras=[(ci, 1.0)])
    assert torch.equal(b32["conv_in.weight"], keep) and torch.equal(s1["conv_in.weight"], s2["conv_in.weight"])
    assert not torch.equal(s1["conv_in.weight"], keep)
