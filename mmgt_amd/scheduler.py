"""DDIM scheduler for the Stage-2 sampler: host-side coefficient tables only (the tensor update is the HIP kernel
mmgt_cfg_ddim_step).

Restates diffusers==0.24.0 `DDIMScheduler` (un-vendored dependency of the reference, requirements.txt:36) under the
reference's settings config/prompts/animation.yaml:80-89: linear betas 0.00085..0.012, rescale_betas_zero_snr,
v_prediction, trailing timestep spacing, clip_sample False, set_alpha_to_one True, eta = 0.
"""
import math

import numpy as np
import torch


def _rescale_zero_terminal_snr(betas):
    alphas_bar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0 = alphas_bar_sqrt[0].clone()
    aT = alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt = (alphas_bar_sqrt - aT) * (a0 / (a0 - aT))
    alphas_bar = alphas_bar_sqrt ** 2
    alphas = torch.cat([alphas_bar[0:1], alphas_bar[1:] / alphas_bar[:-1]])
    return 1 - alphas


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                 clip_sample=False, set_alpha_to_one=True, steps_offset=1, prediction_type="v_prediction",
                 rescale_betas_zero_snr=True, timestep_spacing="trailing", **unused):
        if beta_schedule != "linear":
            raise NotImplementedError(f"beta_schedule {beta_schedule}")
        if timestep_spacing != "trailing" or prediction_type != "v_prediction" or clip_sample:
            raise NotImplementedError("only the reference's sampler settings are implemented (animation.yaml:80-89)")
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        if rescale_betas_zero_snr:
            betas = _rescale_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps / num_inference_steps
        ts = np.round(np.arange(self.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step_coefficients(self, timestep):
        """(sqrt(abar_t), sqrt(1 - abar_t), sqrt(abar_prev), sqrt(1 - abar_prev)) as fp32-rounded Python floats."""
        t = int(timestep)
        prev = t - self.num_train_timesteps // self.num_inference_steps   # the reference's N=30 quirk included
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else torch.tensor(self.final_alpha_cumprod, dtype=torch.float32)
        f = lambda x: float(x.to(torch.float32))
        return f(a_t ** 0.5), f((1 - a_t) ** 0.5), f(a_p ** 0.5), f((1 - a_p) ** 0.5)
