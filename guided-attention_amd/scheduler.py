"""DDIM scheduler with the SD-1.x / SD-2.x configuration the reference forces
(pipeline_guided_attention.py:883-887: `DDIMScheduler.from_config(...)`, `set_timesteps(50)`,
`step(noise_pred, t, latents)`, `alphas_cumprod`, `scale_model_input`).  diffusers 0.12.1 is not
vendored in the reference; this follows its published algorithm (Song et al., DDIM, eq. 12, eta = 0):
scaled-linear betas 0.00085 -> 0.012 over 1000 train steps, steps_offset = 1, set_alpha_to_one = False,
clip_sample = False, which yields the timesteps 981, 961, ..., 1 the reference records
(utils/shared_state.py:8).  Parity of this file is "unpinned" (third-party arithmetic).
"""
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch


@dataclass
class DDIMOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False, prediction_type="epsilon"):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule="scaled_linear", steps_offset=steps_offset,
                                      set_alpha_to_one=set_alpha_to_one, clip_sample=False,
                                      prediction_type=prediction_type)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        if isinstance(config, dict):
            return cls(**{k: v for k, v in config.items() if k in ("num_train_timesteps", "beta_start", "beta_end",
                                                                   "steps_offset", "set_alpha_to_one",
                                                                   "prediction_type")})
        return cls(config.num_train_timesteps, config.beta_start, config.beta_end, config.steps_offset,
                   config.set_alpha_to_one, getattr(config, "prediction_type", "epsilon"))

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def alphas_for(self, timestep):
        """(alpha_bar_t, alpha_bar_prev) as Python floats for one inference step."""
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def step(self, model_output, timestep, sample, eta=0.0, **kwargs):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is outside the guided-attention path (the reference always passes 0)")
        a_t, a_prev = self.alphas_for(timestep)
        if self.config.prediction_type == "epsilon":
            eps = model_output
            x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        elif self.config.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - (1 - a_t) ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + (1 - a_t) ** 0.5 * sample
        else:
            raise ValueError(self.config.prediction_type)
        prev = a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps
        return DDIMOutput(prev_sample=prev, pred_original_sample=x0)
