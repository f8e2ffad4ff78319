"""Deterministic stand-ins for the models around the pipeline class (UNet, warm-up UNet, VAE, depth detector, scheduler, image
processor, prompt encoder), shared by tests/golden/gen_golden.py (driving the REFERENCE's StreamAnimateDiffusionDepth methods
on a fake `self`) and tests/test_host_logic.py (driving this repo's mirror).  Each mock is a cheap function of EVERY input it is
handed -- including kv_cache contents, pe_idx, update_idx and the attention bias -- and the UNet mocks mutate the caches, so a
wrong argument, a missed buffer shift or a different order of random draws changes the captured numbers."""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

H = W = 64          # image size; latent 8 x 8 (vae_scale_factor 8)
N_CACHES = 3


class MockScheduler:
    """alphas_cumprod table + LCM boundary scalings (any smooth functions will do: both sides read the same object)"""

    def __init__(self):
        self.alphas_cumprod = torch.linspace(0.9991, 0.0047, 1000) ** 2
        self.timesteps = torch.arange(999, -1, -20)            # 50 steps
        self.config = {}

    def get_scalings_for_boundary_condition_discrete(self, t):
        s = torch.as_tensor(float(t)) * 10.0 / 1000.0
        c_skip = 0.25 / (s ** 2 + 0.25)
        c_out = s / (s ** 2 + 0.25) ** 0.5
        return c_skip, c_out


class MockImageProcessor:
    def preprocess(self, image, height, width):
        image = image if image.ndim == 4 else image[None]
        return 2.0 * image - 1.0


class MockVAE:
    dtype = torch.float32
    config = SimpleNamespace(scaling_factor=0.5)

    def encode(self, x):
        lat = F.avg_pool2d(x, 8)
        return SimpleNamespace(latents=torch.cat([lat, lat.mean(1, keepdim=True)], 1) * 1.5)      # 4 channels

    def decode(self, z, return_dict=False):
        return (F.interpolate(z[:, :3] * 0.02 + z[:, 3:4] * 0.004, scale_factor=8),)


def retrieve_latents(enc, generator=None):
    return enc.latents


class MockDepth:
    dtype = torch.float32

    def __call__(self, x):
        return (x.mean(1) * 3.0 + x[:, 0] ** 2 + 4.0)            # [B, 384, 384], not constant


class MockPipe:
    vae_scale_factor = 8
    device = torch.device("cpu")

    def __init__(self):
        self.calls = []

    def _encode_prompt(self, prompt, **kw):
        self.calls.append((prompt, tuple(sorted(kw))))
        g = torch.Generator().manual_seed(sum(map(ord, prompt)))
        return [torch.randn(1, 77, 16, generator=g), torch.zeros(1, 77, 16)]


class MockStreamUNet:
    """per-frame UNet: eps = f(sample, timestep, depth, prompt, bias, pe_idx, update_idx, caches); writes slot update_idx[n] of
    every cache row n (like the real one) and returns the same list object."""

    def __init__(self):
        self.log = []

    def __call__(self, x, t, depth_sample=None, encoder_hidden_states=None, temporal_attention_mask=None, kv_cache=None,
                 pe_idx=None, update_idx=None, return_dict=True, **kw):
        n = x.shape[0]
        self.log.append(dict(t=t.clone(), bias=temporal_attention_mask.clone(), pe_idx=pe_idx.clone(), update_idx=update_idx.clone()))
        live = (temporal_attention_mask == 0).float()                                             # [n, L]
        ctx = torch.zeros(n)
        for c in kv_cache:                                                                        # [n, 2, T, L, C]
            ctx = ctx + (c.mean(dim=(1, 2, 4)) * live * (1.0 + 0.01 * pe_idx.float())).sum(1)
        eps = (0.6 * x - 0.3 * depth_sample + 0.001 * t.float().view(n, 1, 1, 1, 1)
               + 0.05 * encoder_hidden_states.mean(dim=(1, 2)).view(n, 1, 1, 1, 1) + 0.02 * ctx.view(n, 1, 1, 1, 1))
        for c in kv_cache:
            for r in range(n):
                c[r, :, :, int(update_idx[r])] = x[r].mean() + 0.1 * depth_sample[r].mean()
        return {"sample": eps, "kv_cache": kv_cache}


class MockWarmupUNet:
    """the reference's second module: called with the ROW SLICES cache[idx] ([2, T, L, C]); fills the first F slots."""

    def to(self, *a, **k):
        return self

    def __call__(self, x, t, temporal_attention_mask=None, depth_sample=None, encoder_hidden_states=None, kv_cache=None,
                 return_dict=True):
        f = x.shape[2]
        for c in kv_cache:
            c[:, :, :f] = (x[0].mean(dim=(0, 2, 3)) + 0.01 * t.float()).view(1, 1, f, 1)
        eps = 0.4 * x + 0.2 * depth_sample - 0.0005 * t.float().view(-1, 1, 1, 1, 1) + 0.03 * encoder_hidden_states.mean()
        return {"sample": eps}


def make_caches(n, L=16):
    return [torch.zeros(n, 2, 4, L, 8) for _ in range(N_CACHES)]


def frames(k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(k, 3, H, W, generator=g)


class NoCudaEvent:
    """torch.cuda.Event stand-in for CPU runs of `__call__` (both pipeline classes time themselves with CUDA events)"""

    def __init__(self, *a, **k):
        pass

    def record(self, *a):
        pass

    def elapsed_time(self, other):
        return 0.0
