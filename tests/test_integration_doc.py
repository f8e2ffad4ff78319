"""CPU test (-m "not gpu"): the code blocks INTEGRATION.md marks `# [executable]` are run verbatim against a stand-in for the
reference's `stream` object (plans are built and validated, never launched: l2d_set_dry_run), following the reference's
own warm-up loop (pipeline_stream_animation_depth.py:317-328) and per-frame call (:456-466)."""
import os
import re
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if b.lstrip().startswith("# [executable]")]


@pytest.fixture()
def dry_run():
    from live2diff_amd import _lib
    _lib.set_dry_run(True)
    yield
    _lib.set_dry_run(False)


def test_integration_md_snippets_run(dry_run, monkeypatch):
    import live2diff_amd.config as C
    from live2diff_amd.weights import unet_param_spec
    blocks = _blocks()
    assert len(blocks) >= 4, "INTEGRATION.md lost its executable blocks"
    ns = {}
    for b in blocks:
        exec(compile(b, "INTEGRATION.md", "exec"), ns)
    # a small-width UNet stands in for SD-1.5 so the CPU test stays light; the snippet's own call is untouched
    tiny = lambda window_size=16, sink_size=8, **k: C.tiny_config(window_size=window_size, sink_size=sink_size,
                                                                  channels=(64, 128, 128, 128), cross_attention_dim=64)
    monkeypatch.setattr(C, "sd15_config", tiny)
    cfg = tiny()
    sd = {k: torch.zeros(shp, dtype=torch.float16) for k, shp in unet_param_spec(cfg).items()}
    stream = SimpleNamespace(pipe=SimpleNamespace(unet=SimpleNamespace(state_dict=lambda: sd)))
    t_index_list = [30, 40]
    unet = ns["install_hip_backend"](stream, 128, 128, t_index_list, "cpu")
    assert stream.unet is unet and len(stream.kv_cache_list) == 40 and stream.unet_warmup.full_cache is stream.kv_cache_list
    # the reference's warm-up loop (:317-328): the second module is called with the ROW SLICES of the caches
    N, F_, h, w = 2, cfg.sink_size, 16, 16
    x = torch.zeros(1, 4, F_, h, w, dtype=torch.float16)
    enc = torch.zeros(N, 77, cfg.cross_attention_dim, dtype=torch.float16)
    seen = []
    real = unet.warmup
    monkeypatch.setattr(unet, "warmup", lambda *a, **k: (seen.append(k["row"]), real(*a, **k))[1])
    for idx, t in enumerate(torch.tensor([399, 199])):
        out = stream.unet_warmup(x, t.view(1), temporal_attention_mask=None, depth_sample=x, encoder_hidden_states=enc[0:1],
                                 kv_cache=[c[idx] for c in stream.kv_cache_list], return_dict=True)
        assert out["sample"].shape == (1, 4, F_, h, w)
    assert seen == [0, 1]
    # the per-frame call (:456-466)
    out = stream.unet(torch.zeros(N, 4, 1, h, w, dtype=torch.float16), torch.tensor([399, 199]), depth_sample=torch.zeros(N, 4, 1, h, w, dtype=torch.float16),
                      encoder_hidden_states=enc, temporal_attention_mask=torch.zeros(N, 16, dtype=torch.float16),
                      kv_cache=stream.kv_cache_list, pe_idx=torch.zeros(N, 16, dtype=torch.int64),
                      update_idx=torch.tensor([8, 9]), return_dict=True)
    assert out["sample"].shape == (N, 4, 1, h, w) and out["kv_cache"] is stream.kv_cache_list
    # the sibling vae slot
    from live2diff_amd.vae_hip import taesd_param_spec
    vsd = {k: torch.zeros(shp, dtype=torch.float16) for k, shp in taesd_param_spec().items()}
    stream.vae = SimpleNamespace(state_dict=lambda: vsd)
    vae = ns["install_hip_vae"](stream, "cpu")
    assert stream.vae is vae and vae.config.scaling_factor == 1.0 and vae.dtype == torch.float16
    assert vae.encode(torch.zeros(1, 3, 128, 128, dtype=torch.float16)).latents.shape == (1, 4, 16, 16)
    assert vae.decode(torch.zeros(1, 4, 16, 16, dtype=torch.float16), return_dict=False)[0].shape == (1, 3, 128, 128)
    # the depth-detector slot: MidasDetector.state_dict() carries the wrapper's `model.` prefix
    from live2diff_amd.midas_hip import midas_param_spec
    dsd = {"model." + k: torch.zeros(shp, dtype=torch.float16) for k, shp in midas_param_spec().items()}
    stream.depth_detector = SimpleNamespace(state_dict=lambda: dsd)
    det = ns["install_hip_depth"](stream, "cpu")
    assert stream.depth_detector is det and det.dtype == torch.float16
    assert det(torch.zeros(1, 3, 384, 384, dtype=torch.float16)).shape == (1, 384, 384)
