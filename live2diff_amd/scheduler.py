"""LCM scheduler scalars used by the stream pipeline.

The reference takes them from `diffusers.LCMScheduler` (pipeline_stream_animation_depth.py:54-55,263,
278-279; config configs/base_config.yaml:30-36).  diffusers is third-party and absent from the build image,
so the documented 0.25.0 semantics are restated here (parity unpinned by the reference):
  * betas: "linear" -> linspace(beta_start, beta_end, T); "scaled_linear" -> linspace(sqrt, sqrt)^2
  * timesteps: LCM origin steps (k * T/50 - 1, k = 1..50) reversed, then `num_inference_steps` of them picked at
    indices floor(linspace(0, 50, n, endpoint=False)) (0.25.0's rule; equal to the older stride rule only when
    50 % n == 0: n = 4 gives [999, 759, 499, 259], the stride rule would give [999, 759, 519, 279])
  * boundary-condition scalings: c_skip = s^2/((10 t)^2 + s^2), c_out = 10 t / sqrt((10 t)^2 + s^2), s = 0.5
"""
import numpy as np
import torch


class LCMSchedule:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                 original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5, **_ignored):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise ValueError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps = num_train_timesteps
        self.original_inference_steps = original_inference_steps
        self.timestep_scaling = timestep_scaling
        self.sigma_data = sigma_data
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        c = self.num_train_timesteps // self.original_inference_steps
        origin = np.asarray(list(range(1, self.original_inference_steps + 1))) * c - 1
        if num_inference_steps > len(origin):
            raise ValueError("num_inference_steps must be <= original_inference_steps")
        lcm = origin[::-1]
        idx = np.floor(np.linspace(0, len(lcm), num=num_inference_steps, endpoint=False)).astype(np.int64)
        ts = lcm[idx]
        self.timesteps = torch.from_numpy(ts.copy()).long()
        if device is not None:
            self.timesteps = self.timesteps.to(device)
        return self.timesteps

    def get_scalings_for_boundary_condition_discrete(self, timestep):
        t = torch.as_tensor(timestep, dtype=torch.float32).cpu() * self.timestep_scaling
        s2 = self.sigma_data ** 2
        return s2 / (t ** 2 + s2), t / (t ** 2 + s2) ** 0.5
