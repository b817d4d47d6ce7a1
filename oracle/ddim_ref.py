"""ORACLE (test infrastructure): diffusers==0.24.0 DDIMScheduler restated in plain torch fp32 for the reference's
settings (config/prompts/animation.yaml:80-89).  diffusers is an un-vendored dependency
(requirements.txt:36) and the reference holds no test for it: PARITY UNPINNED, anchored on analytic known answers
(tests/test_host_logic.py: abar_999 == 0 after the zero-SNR rescale, trailing timesteps 999, 959, ..., 39 for N=25)."""
import numpy as np
import torch


class DDIMRef:
    def __init__(self, beta_start=0.00085, beta_end=0.012, n_train=1000):
        betas = torch.linspace(beta_start, beta_end, n_train, dtype=torch.float32)
        # rescale_zero_terminal_snr
        alphas = 1.0 - betas
        alphas_bar_sqrt = torch.cumprod(alphas, dim=0).sqrt()
        a0, aT = alphas_bar_sqrt[0].clone(), alphas_bar_sqrt[-1].clone()
        alphas_bar_sqrt -= aT
        alphas_bar_sqrt *= a0 / (a0 - aT)
        alphas_bar = alphas_bar_sqrt ** 2
        alphas = torch.cat([alphas_bar[0:1], alphas_bar[1:] / alphas_bar[:-1]])
        self.alphas_cumprod = torch.cumprod(alphas, dim=0)
        self.n_train = n_train
        self.final_alpha_cumprod = torch.tensor(1.0)

    def set_timesteps(self, n):
        self.n = n
        self.timesteps = torch.from_numpy((np.round(np.arange(self.n_train, 0, -self.n_train / n)) - 1).astype(np.int64))

    def step(self, model_output, t, sample):
        t = int(t)
        prev = t - self.n_train // self.n
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output          # v-prediction
        eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps                    # eta = 0
