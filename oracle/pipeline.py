"""Oracle (TEST INFRASTRUCTURE ONLY): the guided-attention denoising loop restated on CPU in fp32.

Follows (paths relative to the reference checkout):
  pipeline_guided_attention.py:905-1053  __call__ loop body, recurse / re-noise
  pipeline_guided_attention.py:475-581   _perform_iterative_refinement_step
  pipeline_guided_attention.py:456-470   _update_latent
  pipeline_guided_attention.py:298-354   _aggregate_and_get_max_attention_per_token
The UNet is any module with the package's `UNet2DConditionModel` interface, run here with the
oracle's plain-PyTorch attention processors (oracle/attention.py); the DDIM update is restated in
`ddim_step` from the published diffusers-0.12.1 algorithm (third-party: parity unpinned).
Pinned by tests/golden/g9_loop.npz: the reference's own __call__ driven on a reduced-width UNet.
"""
import math

import numpy as np
import torch

from . import attention as oattn
from . import loss as oloss


def alphas_cumprod(num_train=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(steps, num_train=1000, offset=1):
    return [int(v) + offset for v in (np.arange(0, steps) * (num_train // steps)).round()[::-1]]


def ddim_step(eps, t, x, acp, steps, num_train=1000):
    prev = t - num_train // steps
    a_t = float(acp[t])
    a_prev = float(acp[prev]) if prev >= 0 else float(acp[0])
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def install_processors(unet, store, pww=None):
    """utils/ptp_utils.py:149-175 register_attention_control with the oracle processor (pww: see OracleAttnProcessor)."""
    procs = {}
    for name in unet.attn_processors.keys():
        place = "mid" if name.startswith("mid_block") else ("up" if name.startswith("up_blocks") else "down")
        procs[name] = oattn.OracleAttnProcessor(store, place, pww)
    unet.set_attn_processor(procs)
    store.num_att_layers = len(procs)


class GuidedSampler:
    def __init__(self, unet, plan, *, thresholds, config_thresholds=None, only_update_on_threshold_steps=True,
                 max_iter_to_alter=25, run_standard_sd=False, guidance_scale=7.5, steps=50, scale_factor=20,
                 scale_range=(1.0, 0.5), smooth=True, sigma=0.5, kernel_size=3, attention_res=16,
                 normalize_eot=False, n_prompt_tokens=None, max_refinement_steps=10, paint_with_words=None):
        """paint_with_words: None or dict(stop=, weight=): the additive box mask of utils/ptp_utils.py:113-138 for the
        BOX tokens of `plan` while the step index is below `stop`."""
        self.unet, self.plan = unet, plan
        self.thresholds = dict(thresholds) if len(thresholds) else {0: float("inf")}
        self.config_thresholds = dict(config_thresholds if config_thresholds is not None else thresholds)
        self.only_thr = only_update_on_threshold_steps
        self.max_iter_to_alter = max_iter_to_alter
        self.run_standard_sd = run_standard_sd
        self.gs, self.steps = guidance_scale, steps
        self.scale_factor = scale_factor
        self.scale_range = np.linspace(scale_range[0], scale_range[1], steps)
        self.loss_kw = dict(smooth=smooth, sigma=sigma, kernel_size=kernel_size, normalize_eot=normalize_eot,
                            n_prompt_tokens=n_prompt_tokens)
        self.res = attention_res
        self.max_ref = max_refinement_steps
        self.recurse_steps = max(plan.hyper.get("recurse_steps", 1), 1)
        self.recurse_until = plan.hyper.get("recurse_until", 20)
        self.store = oattn.OracleStore()
        self.acp = alphas_cumprod()
        self.cur_i, self.cur_t = 0, 0
        pww = None
        if paint_with_words:
            boxes = {e["index"]: e["geom"] for e in plan.entries if e["kind"] == "BOX"}

            def pww(n_pixels):  # state.cur_time_step_iter < stop; sigma_t = sqrt((1 - a_t) / a_t)  (shared_state.get_sigma)
                if not self.cur_i < paint_with_words["stop"]:
                    return None
                a = float(self.acp[self.cur_t])
                return (oattn.paint_with_words_mask(boxes, n_pixels, plan.hyper["shrink_factor"], paint_with_words["weight"]),
                        math.log(1 + ((1 - a) / a) ** 0.5))
        install_processors(unet, self.store, pww)
        self.calls = {"fwd_b1_grad": 0, "bwd": 0, "fwd_b2": 0, "loss_evals": 0}
        self.trace = []

    # one guidance evaluation: forward with grad + aggregated maps + loss
    def _evaluate(self, latents, t, cond):
        latents = latents.clone().detach().requires_grad_(True)
        self.unet(latents, t, encoder_hidden_states=cond)
        self.calls["fwd_b1_grad"] += 1
        A = oattn.aggregate(self.store.attention_store, self.res, ("up", "down", "mid"), True)
        r = oloss.loss_torch(A, self.plan, **self.loss_kw)
        self.calls["loss_evals"] += 1
        sums = oloss.subprompt_sums(self.plan, r["unscaled"])
        return latents, r, sums

    def _update(self, latents, loss, step):
        (g,) = torch.autograd.grad(loss, [latents], retain_graph=True)
        self.calls["bwd"] += 1
        return latents.detach() - step * g

    def _refine(self, latents, t, cond, step, i):
        it = 0
        sums = None
        while sums is None or not oloss.meets_threshold(i, self.config_thresholds, sums):
            it += 1
            latents, r, sums = self._evaluate(latents, t, cond)
            self.trace.append(("refine", i, it, float(r["loss"])))
            if float(r["loss"]) != 0:
                latents = self._update(latents, r["loss"], step)
            if it >= self.max_ref:
                break
        latents, r, sums = self._evaluate(latents, t, cond)
        self.trace.append(("refine_final", i, it, float(r["loss"])))
        return latents, r

    @torch.no_grad()
    def sample(self, latents, prompt_embeds, renoise_noise=()):
        """prompt_embeds (2, n_tok, dim) = [uncond, cond]; latents (1, 4, h, w) fp32; returns final latents."""
        renoise = list(renoise_noise)
        cond = prompt_embeds[1:2]
        latents = latents.clone()
        for i, t in enumerate(ddim_timesteps(self.steps)):
            self.cur_i, self.cur_t = i, t
            for rstep in range(self.recurse_steps):
                updated = False
                with torch.enable_grad():
                    latents, r, sums = self._evaluate(latents, t, cond)
                    self.trace.append(("eval", i, rstep, float(r["loss"])))
                    if not self.run_standard_sd:
                        step = self.scale_factor * math.sqrt(self.scale_range[i])
                        if not oloss.meets_threshold(i, self.thresholds, sums):
                            updated = True
                            latents, r = self._refine(latents, t, cond, step, i)
                        if (not self.only_thr and i < self.max_iter_to_alter) or (i in self.config_thresholds):
                            # `sums` is deliberately the PRE-refinement value (reference :999)
                            if not oloss.meets_threshold(-1, self.config_thresholds, sums):
                                updated = True
                                if float(r["loss"]) != 0:
                                    latents = self._update(latents, r["loss"], step)
                latents = latents.detach()
                eps = self.unet(torch.cat([latents] * 2), t, encoder_hidden_states=prompt_embeds).sample
                self.calls["fwd_b2"] += 1
                eps_u, eps_t = eps.chunk(2)
                eps = eps_u + self.gs * (eps_t - eps_u)
                latents = ddim_step(eps, t, latents, self.acp, self.steps)
                self.trace.append(("ddim", i, rstep, float(latents.abs().mean())))
                if i > self.recurse_until or not updated:
                    break
                if rstep != self.recurse_steps - 1:
                    prev_t = t - 1000 // self.steps
                    if prev_t > 0:
                        Bt = self.acp[t] / self.acp[prev_t]
                        latents = Bt.sqrt() * latents + (1 - Bt).sqrt() * renoise.pop(0)
        return latents
