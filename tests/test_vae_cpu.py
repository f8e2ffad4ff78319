"""CPU tests (-m "not gpu") of the tiny-VAE / depth-glue host side: parameter inventory, plan validation without a device,
and the oracle's own invariants (oracle/taesd_ref.py is third-party topology restated: parity unpinned)."""
import pytest
import torch
import torch.nn.functional as F


@pytest.fixture()
def dry_run():
    from live2diff_amd import _lib
    _lib.set_dry_run(True)
    yield
    _lib.set_dry_run(False)


def test_taesd_spec_matches_oracle_and_published_size():
    from live2diff_amd.vae_hip import taesd_param_spec
    from oracle import taesd_ref as T
    spec = taesd_param_spec()
    assert list(spec.items()) == list(T.taesd_param_spec().items())      # product and oracle agree on every key and shape
    n = sum(torch.Size(s).numel() for s in spec.values())
    assert n == 2445063                                                   # madebyollin/taesd: ~2.4 M parameters
    # diffusers nn.Sequential indices: encoder ends with layers.14 (64 -> 4), decoder with layers.18 (64 -> 3, bias)
    assert spec["encoder.layers.14.weight"] == (4, 64, 3, 3) and spec["decoder.layers.18.weight"] == (3, 64, 3, 3)
    assert "decoder.layers.18.bias" in spec and "decoder.layers.6.bias" not in spec and "encoder.layers.2.bias" not in spec
    assert not any(k.startswith("decoder.layers.1.") or k.startswith("decoder.layers.5.") for k in spec)   # ReLU / Upsample


@pytest.mark.parametrize("B,H,W", [(1, 512, 512), (8, 256, 256), (1, 576, 1024)])
def test_vae_plans_validate_without_gpu(dry_run, B, H, W):
    from live2diff_amd.vae_hip import HipTinyVAE, taesd_param_spec
    sd = {k: torch.zeros(s, dtype=torch.float16) for k, s in taesd_param_spec().items()}
    v = HipTinyVAE(sd, device="cpu")
    lat = v.encode(torch.zeros(B, 3, H, W, dtype=torch.float16)).latents
    assert lat.shape == (B, 4, H // 8, W // 8)
    img = v.decode(torch.zeros(B, 4, H // 8, W // 8, dtype=torch.float16), return_dict=False)[0]
    assert img.shape == (B, 3, H, W)
    assert v.config.scaling_factor == 1.0 and v.dtype == torch.float16 and v.to("cuda") is v
    s = v.plan_summary()
    assert s[("enc", B, H, W)]["n_ops"] == 37 and s[("dec", B, H // 8, W // 8)]["n_ops"] == 37      # 35 convs + 2 layout ops each
    with pytest.raises(KeyError):
        HipTinyVAE({}, device="cpu")
    with pytest.raises(ValueError):
        v.encode(torch.zeros(1, 4, 64, 64, dtype=torch.float16))


def test_taesd_oracle_invariants():
    from live2diff_amd.vae_hip import random_taesd_state_dict
    from oracle import taesd_ref as T
    sd = {k: v.float() for k, v in random_taesd_state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 64, 96, generator=g) * 2 - 1
    z = T.taesd_encode(x, sd)
    assert z.shape == (2, 4, 8, 12)
    y = T.taesd_decode(z, sd)
    assert y.shape == x.shape and torch.isfinite(y).all()
    # samples of a batch do not interact
    assert torch.allclose(T.taesd_encode(x[1:], sd), z[1:], atol=1e-5) and torch.allclose(T.taesd_decode(z[:1], sd), y[:1], atol=1e-5)
    # the encoder sees (x + 1) / 2: an image of -1 everywhere is a zero image, whose encoding is bias-only and flat in the interior
    zz = T.taesd_encode(torch.full((1, 3, 256, 256), -1.0), sd)
    assert (zz[..., 14:18, 14:18] - zz[..., 15:16, 15:16]).abs().max() < 1e-4
    # decoder input is squashed by tanh(z/3)*3: huge latents saturate
    big = torch.full((1, 4, 4, 4), 1e4)
    assert torch.allclose(T.taesd_decode(big, sd), T.taesd_decode(big * 10, sd), atol=1e-5)


def test_depth_glue_oracle_matches_the_reference_expressions():
    """oracle.depth_glue vs the literal expressions of reference pipeline_stream_animation_depth.py:560-567."""
    from oracle.taesd_ref import depth_glue
    g = torch.Generator().manual_seed(1)
    d = torch.rand(3, 384, 384, generator=g) * 7 + 2
    dn = (d - d.min()) / (d.max() - d.min())
    dn = dn[:, None].repeat(1, 3, 1, 1) * 2 - 1
    ref = F.interpolate(dn, (512, 768), mode="bilinear", align_corners=False)
    assert torch.equal(depth_glue(d, 512, 768), ref)
