"""`GuidedAttention` — the per-step guided-attention sampling loop with the reference's interface
(pipeline_guided_attention.py:37-1123), host code in Python over the HIP kernels of libga_hip.so:

  per denoising step (reference :925-1053)
    A. guidance pass (autograd on):  UNet(latents, t, cond) with the capture processors  -> K1 x 32
       aggregate the 16x16 cross maps (K2) -> smoothed box loss (K3+K4) -> thresholds (host)
       iterative refinement / gradient step: autograd back to the latents (K1 bwd, K3+K4 bwd),
       latents <- latents - step * grad (K5)
    B. CFG pass (no grad):  UNet([latents]*2, t, [uncond, cond]) -> fused CFG combine + DDIM step
    C. recurse: re-noise back to level t (K6) and repeat while the step keeps updating

Kept from the reference: class name (`GuidedAttention`, alias `GuidedAttentionPipeline`), the
`__call__` keyword surface, the method names of the loss path and their return conventions, the use
of `utils.shared_state` globals, the control flow including its quirks (the second threshold test
reads the pre-refinement losses, :999).  Not reproduced (side effects off the path, declared in
DESIGN.md): per-token PNG dumps, predicted-x0 PNGs for steps 0-2, latent statistics logging,
deep-feature optimisation, SGD-momentum refinement, the safety checker.
"""
import math
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import ops
from ._lib import GaError
from .scheduler import DDIMScheduler
from .utils import helpers
from .utils import shared_state as state
from .utils.ptp_utils import AttentionStore, aggregate_attention, stored_maps

TERM = {"max_loss": 0, "col": 1, "row": 2, "inside_loss": 3, "outside_loss": 4, "token_loss": 5, "unscaled": 6}


class PipelineOutput(SimpleNamespace):
    """`.images`, `.nsfw_content_detected`, plus `.latents` and `.unet_calls` (run-time call counters)."""


class GuidedAttention:
    vae_scale_factor = 8

    def __init__(self, unet, scheduler=None, vae=None, text_encoder=None, tokenizer=None):
        self.unet = unet
        self.scheduler = scheduler or DDIMScheduler()
        self.vae = vae
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        self.prompt = None
        self.inside_iterative_refinement = False
        # "full": run the whole UNet in the guidance pass, as the reference does.  "truncated": stop after the
        # last attention map the loss reads (up_blocks.1): identical latents, fewer FLOPs (declared when used).
        self.guidance_forward = "full"
        # False keeps the reference's behaviour of evaluating the guidance pass + loss on every step even when
        # no update can follow (steps outside `thresholds` with only_update_on_threshold_steps): log-only work.
        self.skip_unused_guidance = False
        # True: the two UNet passes are captured once into hipGraphs (guidance forward + loss, its backward, the CFG
        # forward) and replayed — same kernels, no per-launch host work.  False: eager launches.
        self.use_graphs = False
        # With hipGraphs: on steps where no latent update can follow the guidance evaluation (its loss is only
        # logged), run the guidance forward (cond) and the CFG pair (uncond, cond) — three independent evaluations of
        # the same latents — as ONE batch-3 pass.  Nothing is skipped; False runs them as two passes (B=1, B=2).
        self.batch_loss_only_guidance = True
        # The reference writes per-token attention-map PNGs at EVERY loss evaluation (:243-246), predicted-x0 PNGs for the
        # steps in shared_state.always_save_iter (:1036-1037) and latent statistics at every step (:1031), whatever
        # config.diagnostic_level says.  Here those side effects are opt-in: True reproduces them (each costs device
        # syncs and host I/O; hipGraph replay is bypassed for such a run).  config.diagnostic_level > 0 switches them on
        # as well.  Off by default, never inside a timed benchmark region.
        self.reference_side_effects = False
        self.library_kernels = frozenset()   # see .to()
        # With hipGraphs, inside the iterative refinement: enqueue an iteration's backward, latent update and the NEXT guidance
        # evaluation before reading the iteration's loss table back.  Nothing is speculative about WHAT runs — the update is
        # unconditional whenever the loss is not exactly 0 (reference :551), and whatever the threshold test says afterwards the
        # next launch is an evaluation of the updated latents (the next iteration's, or the final one, :566) — only about the
        # `loss != 0` test, whose other outcome discards and repeats (never counted).  False: enqueue, read, decide, enqueue.
        self.speculative_refinement = True
        self._runner = None
        self._graph_cache = {}
        self.unet_calls = {"fwd_b1_grad": 0, "bwd": 0, "fwd_b2": 0, "loss_evals": 0, "joint_b3": 0}
        self._plan_key = None
        self._plan = None

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, name_or_path, revision=None, torch_dtype=None, random_init=False, unet_config=None,
                        seed=0, weights=True, **kwargs):
        """Local diffusers-layout folder -> weights loaded by name.  There is no network here: an id that is
        not a local folder raises unless random_init=True, which builds seeded random weights of the named
        architecture (SD-1.x for anything but '*2-1*').  weights=False builds the architecture only (no file read,
        no random draw): what the ranks other than 0 do before the weight broadcast (run.load_model)."""
        from pathlib import Path
        from .text import SyntheticTextEncoder, WordTokenizer, load_clip
        from .unet import UNet2DConditionModel, UNetConfig
        from .vae import AutoencoderKLDecoder
        sd21 = "2-1" in str(name_or_path)
        cfg = unet_config or (UNetConfig.sd21(sample_size=64) if sd21 else UNetConfig.sd15())
        folder = Path(str(name_or_path))
        if folder.is_dir() and (folder / "unet").is_dir():
            from safetensors.torch import load_file
            unet = UNet2DConditionModel(cfg)
            if weights:
                unet.load_diffusers_state(load_file(str(folder / "unet" / "diffusion_pytorch_model.safetensors")))
            vae = AutoencoderKLDecoder()
            vae_file = folder / "vae" / "diffusion_pytorch_model.safetensors"
            if vae_file.exists() and weights:
                sd = {k: v for k, v in load_file(str(vae_file)).items()
                      if k.startswith(("decoder.", "post_quant_conv."))}
                vae.load_state_dict(sd, strict=False)
            clip = load_clip(folder)
            tok, enc = clip if clip else (WordTokenizer(), SyntheticTextEncoder(cfg.cross_attention_dim))
        elif random_init:
            unet = UNet2DConditionModel(cfg)
            small = cfg.block_out_channels[0] < 128
            vae = AutoencoderKLDecoder.tiny() if small else AutoencoderKLDecoder()
            if weights:
                unet.init_weights_(seed)
                vae.init_weights_(seed + 1)
            tok, enc = WordTokenizer(pad_token_id=0 if sd21 else 49407), SyntheticTextEncoder(cfg.cross_attention_dim)
        else:
            raise FileNotFoundError(f"{name_or_path!r} is not a local checkpoint folder and there is no network; "
                                    "pass random_init=True for seeded random weights of that architecture")
        pipe = cls(unet, DDIMScheduler(), vae, enc, tok)
        dtype = torch_dtype or (torch.float16 if revision == "fp16" else None)
        if dtype is not None:
            pipe.to(dtype=dtype)
        return pipe

    def to(self, device=None, dtype=None):
        for m in (self.unet, self.vae, self.text_encoder):
            if m is not None:
                m.to(device=device, dtype=dtype)
        for p in self.unet.parameters():
            p.requires_grad_(False)  # only the latents are differentiated (reference :466)
        if self.unet.device.type == "cuda":
            # MI355X layout: activations and conv weights channels-last end to end (MIOpen's NHWC kernels without
            # transposes) and the fused NHWC GroupNorm(+SiLU) HIP kernels in every norm layer
            self.unet.to(memory_format=torch.channels_last)
            self.unet.set_norm_impl(ops.group_norm_act)
            # `library_kernels` (A/B runs and the own-vs-library tests set it before .to()): the kinds named there — "conv",
            # "linear", "cat" — stay on MIOpen / hipBLASLt + the separate LayerNorm, GEGLU and add launches / torch.cat; default:
            # the package's own kernels for all three (16-bit dtypes; fp32 keeps the library)
            from . import fused_linear
            lib = set(self.library_kernels)
            conv = None if "conv" in lib else ops.conv3x3
            linear = None if "linear" in lib else fused_linear
            ops.prepare_device(self.unet.device)   # split-K slabs / tickets exist before any hipGraph capture
            cat = None if "cat" in lib else ops.cat_channels
            self.unet.set_fused_impl(ops.geglu, ops.bias_residual_add, (ops.layer_norm, ops.add_layer_norm), conv, linear, cat)
            # conv_in / conv_out and conv_in's backward to the latents: own kernels (csrc/thin_conv.hip) with the "conv" kind
            self.unet.edge_conv_impl = (ops.conv3x3_thin_apply if conv is not None and
                                        self.unet.dtype in (torch.float16, torch.bfloat16) else None)
            # MIOpen's exhaustive search (cudnn.benchmark) stays OFF.  It executes every candidate solver once per shape, and a
            # candidate of the backward-data search for conv_in (4 <- 64 channels, 32 x 32, fp16) reads past its operands: a GPU
            # memory access fault that killed the process whenever the tensors happened to sit at the end of a mapped segment
            # (round 4, deterministic in `pytest tests/test_pipeline_gpu.py`, the worker thread in a native autograd node:
            # profiles/r4_fault_half_precision_graphs_wide.log; round 3 had seen "the fp32 96x96 backward-data search abort
            # the process once").  Only the three stride-2 backward convolutions (and the edge convolutions of shapes the own
            # kernels do not serve) are still on the library.
            torch.backends.cudnn.benchmark = False
        return self

    @property
    def device(self):
        return self.unet.device

    _execution_device = device

    # ------------------------------------------------------------------ prompt side
    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance,
                       negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None):
        """-> (text_inputs, cat[negative, positive] embeddings)  (reference :64-199)."""
        text_inputs = None
        batch_size = 1 if isinstance(prompt, str) else (len(prompt) if prompt is not None else prompt_embeds.shape[0])
        if prompt_embeds is None:
            text_inputs = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length,
                                         truncation=True, return_tensors="pt")
            prompt_embeds = self.text_encoder(text_inputs.input_ids.to(device))[0]
        prompt_embeds = prompt_embeds.to(dtype=self.unet.dtype, device=device)
        b, n, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, n, -1)
        if do_classifier_free_guidance:
            if negative_prompt_embeds is None:
                if negative_prompt is None:
                    uncond = [""] * batch_size
                elif isinstance(negative_prompt, str):
                    uncond = [negative_prompt]
                else:
                    if batch_size != len(negative_prompt):
                        raise ValueError("`negative_prompt` batch size does not match `prompt`")
                    uncond = negative_prompt
                ids = self.tokenizer(uncond, padding="max_length", max_length=prompt_embeds.shape[1], truncation=True,
                                     return_tensors="pt").input_ids
                negative_prompt_embeds = self.text_encoder(ids.to(device))[0]
            negative_prompt_embeds = negative_prompt_embeds.to(dtype=self.unet.dtype, device=device)
            n = negative_prompt_embeds.shape[1]
            negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(
                batch_size * num_images_per_prompt, n, -1)
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
        return text_inputs, prompt_embeds

    def check_inputs(self, prompt, height, width, callback_steps, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {shape}")
            latents = latents.to(device=device, dtype=dtype)
        return latents * self.scheduler.init_noise_sigma

    def decode_latents(self, latents):
        image = self.vae.decode(latents.to(self.unet.dtype) / self.vae.scaling_factor)
        image = (image / 2 + 0.5).clamp(0, 1)
        return image.detach().cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None]
        return [Image.fromarray(im) for im in (images * 255).round().astype("uint8")]

    def get_token(self, index):
        return self.tokenizer.decode(self.tokenizer(state.config.prompt)["input_ids"][index])

    # ------------------------------------------------------------------ the loss path (HIP)
    def _loss_plan(self, smooth, sigma, kernel_size):
        td = state.config.token_dict
        hp = state.curHyperParams
        entries = []
        for idx, info in td.items():
            if info["loss_type"] == helpers.AnnotationType.BOX:
                entries.append({"index": idx, "kind": "BOX", "geom": tuple(float(v) for v in info["loss"].as_tuple()),
                                "subprompt": info["subprompt"]})
            elif info["loss_type"] == helpers.AnnotationType.COOR:
                entries.append({"index": idx, "kind": "COOR", "geom": tuple(float(v) for v in info["loss"]),
                                "subprompt": info["subprompt"]})
            else:  # KEYWORD tokens only mark sub-prompts for custom losses: no built-in term
                continue
        # keyed on CONTENT (token positions, kinds, geometry, sub-prompts, the hyper-parameters the plan reads): an
        # in-place edit of a Rect invalidates the plan (the captured hipGraphs bake the geometry in by value), and a
        # fresh but equal token_dict — run.execute builds one per (seed, hyper-parameter state) — reuses plan and graphs
        key = (tuple((e["index"], e["kind"], e["geom"], e["subprompt"]) for e in entries),
               tuple((idx, info["loss_type"].name) for idx, info in td.items()),
               smooth, sigma, kernel_size, state.config.sub_prompt_avg_within,
               tuple(sorted((k, str(v)) for k, v in hp.items() if k in ops.LossPlan.HYPER_KEYS)))
        if key != self._plan_key:
            self._plan = ops.LossPlan(entries, hp, smooth, sigma, kernel_size, state.config.sub_prompt_avg_within)
            self._plan_key = key
        return self._plan

    def _last_text_index(self, n_tok, normalize_eot):
        if normalize_eot:
            prompt = self.prompt[0] if isinstance(self.prompt, list) else self.prompt
            return len(self.tokenizer(prompt)["input_ids"]) - 1
        return n_tok - 1

    def _aggregate_loss_device(self, attention_store, attention_res, smooth_attentions, sigma, kernel_size, normalize_eot):
        """aggregate_attention + the loss evaluation, device half.  The common case — built-in box / coordinate terms
        only, no diagnostics — is ONE launch (ga_aggregate_loss_fwd: the head-map mean never makes its own pass and its
        backward is one launch as well); custom Python losses and the PNG dumps need the aggregate as a differentiable
        tensor of its own and take the two-step form.  Results are identical (GPU test)."""
        plan = self._loss_plan(smooth_attentions, sigma, kernel_size)
        custom = getattr(state.config, "custom_loss", None)
        if plan.T > 0 and not custom and not self._dump and self.fused_aggregate_loss:
            maps = stored_maps(attention_store, attention_res, ("up", "down", "mid"), True, 0)
            last_idx = self._last_text_index(maps[0].shape[-1], normalize_eot)
            _, terms, loss = ops.AggregateSmoothLoss.apply(attention_res, 1, last_idx, plan, *maps)
            packed = torch.cat([terms.detach().reshape(-1), loss.detach(), loss.new_zeros(1)])
            return terms, loss, None, plan, packed
        attention_maps = aggregate_attention(attention_store=attention_store, res=attention_res,
                                             from_where=("up", "down", "mid"), is_cross=True, select=0)
        if self._dump and getattr(state.config, "save_individual_CA_maps", False) and state.cur_time_step_iter == 12:
            self._dump_individual_ca_maps(attention_store, attention_maps, ("up", "down", "mid"))
        return self._loss_device(attention_maps, smooth_attentions, sigma, kernel_size, normalize_eot)

    fused_aggregate_loss = True   # False: always aggregate_attention + loss as two launches (A/B and parity tests)

    def _loss_device(self, attention_maps, smooth_attentions, sigma, kernel_size, normalize_eot):
        """Device half of the loss evaluation (graph-capturable: no host sync): -> (terms, loss, custom, plan)."""
        res, n_tok = attention_maps.shape[0], attention_maps.shape[-1]
        last_idx = self._last_text_index(n_tok, normalize_eot)
        plan = self._loss_plan(smooth_attentions, sigma, kernel_size)
        if plan.T > 0:
            terms, loss = ops.SmoothLoss.apply(attention_maps.reshape(res * res, n_tok), res, 1, last_idx, plan)
        else:  # every annotated token is a KEYWORD of a custom loss (reference: no built-in term, :409-438)
            terms = attention_maps.new_zeros((0, 8))
            loss = attention_maps.new_zeros(1)
        custom = None
        if self._dump:
            self._dump_token_maps(attention_maps, last_idx)
        if hasattr(state.config, "custom_loss") and state.config.custom_loss:
            text_maps = torch.softmax(attention_maps[:, :, 1:last_idx] * 100, dim=-1)
            for _name, (fn, args) in state.config.custom_loss.items():
                v = fn.calc_loss(text_maps, args)
                custom = v if custom is None else custom + v
        # everything the host tests in ONE device->host copy: the term table, the fused loss and the custom-loss value
        cval = custom.detach().float().reshape(-1)[:1] if custom is not None else loss.new_zeros(1)
        packed = torch.cat([terms.detach().reshape(-1), loss.detach(), cval])
        return terms, loss, custom, plan, packed

    def _loss_host(self, terms, loss, custom, plan, packed):
        """Host half: the one device sync per evaluation (the thresholds need the values), then the reference's
        `losses_dict` keys (one entry per guided token) plus the fused results."""
        self.unet_calls["loss_evals"] += 1
        host = packed.cpu()
        host_terms = host[:-2].view(plan.T, 8)
        losses_dict = {k: [terms[t, c] for t in range(plan.T)] for k, c in TERM.items() if c < 5}
        # host_total = the value the reference tests with `loss != 0` (:551, :1002): box terms plus custom loss
        losses_dict["_fused"] = {"loss": loss, "host_terms": host_terms, "host_loss": host[-2:-1], "plan": plan,
                                 "host_custom": host[-1:], "host_total": host[-2:-1] + host[-1:]}
        if custom is not None:
            losses_dict["custom_loss"] = custom
        for t, e in enumerate(plan.entries):
            word = state.config.token_dict[e["index"]]["word"]
            helpers.log(f"{word}: weighted center col {host_terms[t, 1].item()} row {host_terms[t, 2].item()}")
        return losses_dict

    def _compute_max_attention_per_index(self, attention_maps, smooth_attentions=False, sigma=0.5, kernel_size=3,
                                         normalize_eot=False):
        """attention_maps (res, res, n_tokens) f32 -> losses_dict with the reference's keys plus the fused
        results of ga_smooth_loss_fwd (reference :201-296)."""
        return self._loss_host(*self._loss_device(attention_maps, smooth_attentions, sigma, kernel_size, normalize_eot))

    _dump = False  # True while a run reproduces the reference's PNG / log side effects (set per __call__)

    # ------------------------------------------------------------------ diagnostics (reference :243-246, 316-346, 1090-1123)
    def get_innermost_folder(self):
        return str(state.cur_seed)

    def _dump_dir(self):
        d = state.config.output_path / helpers.get_inner_folder_name() / self.get_innermost_folder()
        d.mkdir(exist_ok=True, parents=True)
        return d

    def save_viridis(self, tensor1, tag):
        """Min-max normalised map as a viridis PNG (reference :1096-1103)."""
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        with torch.no_grad():
            x = tensor1.float() - tensor1.float().min()
            x = x / x.max()
            fname = (tag + "_" + helpers.get_meta_prompt_clean() + state.get_name() + "_subiter_" +
                     "{:02d}".format(state.sub_iteration) + ".png")
            plt.imsave(self._dump_dir() / fname, x.detach().cpu().numpy())

    def save_numpy(self, tensor1, tag):
        fname = tag + "_" + helpers.get_meta_prompt_clean() + "_" + str(state.cur_time_step_iter)
        np.save(self._dump_dir() / fname, tensor1.detach().float().cpu().numpy())

    def save_image(self, latent, tag):
        """Decode latents and save the (annotated) image (reference :1114-1123)."""
        image = self.numpy_to_pil(self.decode_latents(latent.detach()))
        fname = helpers.get_meta_prompt_clean() + state.get_name() + "_" + tag
        for ch in "[]:.":
            fname = fname.replace(ch, "_")
        helpers.annotate_image(image[0])
        image[0].save(self._dump_dir() / (fname + ".png"))

    def _dump_token_maps(self, attention_maps, last_idx):
        """Per-token maps of softmax(100 * A[:, :, 1:last]) as PNGs + their sums in the log (reference :238-246)."""
        with torch.no_grad():
            text = torch.softmax(attention_maps.detach()[:, :, 1:last_idx] * 100, dim=-1)
            if getattr(state.config, "save_all_maps", False):
                indices = range(0, len(self.tokenizer(state.config.prompt)["input_ids"][1:-1]))
            else:
                indices = [index - 1 for index in state.config.token_dict.keys()]
            for index_i in indices:
                image = text[:, :, index_i]
                self.save_viridis(image, "_attnmap_" + self.get_token(index_i + 1))
                helpers.log(self.get_token(index_i + 1) + ": " + str(image.sum().item()))

    def _dump_individual_ca_maps(self, attention_store, attention_maps, from_where):
        """config.save_individual_CA_maps at step 12: every head of every stored cross map (token 1), the head mean
        and the aggregate (reference :316-346)."""
        for location in from_where:
            for map_iter, item in enumerate(attention_store.get_average_attention()[f"{location}_cross"], start=1):
                if not torch.is_tensor(item):
                    continue
                res = int(item.shape[1] ** .5)
                cross_maps = item.reshape(-1, res, res, item.shape[-1])
                for head in range(0, min(8, cross_maps.shape[0])):
                    map1 = cross_maps[head, :, :, 1]
                    tag = (f"{location}_res_{res}_head_{head}_mapiter_{map_iter}_avg_{map1.mean().item():.3}"
                           f"_max_{map1.max().item():.3}")
                    self.save_viridis(map1, tag)
                self.save_viridis((cross_maps.sum(0) / 8)[:, :, 1], f"{location}_res_{res}_avgheads_mapiter_{map_iter}")
        self.save_viridis(attention_maps[:, :, 1], "final")

    def _aggregate_and_get_max_attention_per_token(self, attention_store, attention_res=16, smooth_attentions=False,
                                                   sigma=0.5, kernel_size=3, normalize_eot=False):
        return self._loss_host(*self._aggregate_loss_device(attention_store, attention_res, smooth_attentions, sigma,
                                                            kernel_size, normalize_eot))

    @staticmethod
    def group_losses_by_sumprompt(losses):
        """[(token index | None, value)] -> (total, {sub-prompt: value}); sum within a sub-prompt, or the mean
        when config.sub_prompt_avg_within (reference :359-387).  fp32 accumulation like the reference's tensors."""
        groups = {}
        for tok, val in losses:
            sub = None if tok is None else state.config.token_dict[tok]["subprompt"]
            groups.setdefault(sub, []).append(val)
        total = None
        finals = {}
        for sub, vals in groups.items():
            acc = None
            for v in vals:
                v = v.float() if torch.is_tensor(v) else torch.tensor([float(v)])
                if state.config.sub_prompt_avg_within:
                    v = v / len(vals)
                acc = v if acc is None else acc + v
            finals[sub] = acc
            total = acc if total is None else total + acc
        if total is None:
            total = torch.zeros(1)
        return total, finals

    @staticmethod
    def get_centering_loss(center, losses_dict, i):
        res = 16
        part1 = (losses_dict["col"][i] - center[0] * res).abs() / (res - 1.)
        part2 = 4. * (losses_dict["row"][i] - center[1] * res).abs() / (res - 1.)
        return part1 + part2

    @staticmethod
    def _compute_loss(losses_dict, return_losses=False):
        """-> (loss (1,) on the device with grad, losses [(token, value)], unscaled_losses [(token, value)]).
        The per-token values are host-side fp32 scalars (already synchronised by the loss evaluation)."""
        fused = losses_dict["_fused"]
        host, plan = fused["host_terms"], fused["plan"]
        losses, unscaled = [], []
        for t, e in enumerate(plan.entries):
            info = state.config.token_dict[e["index"]]
            losses.append((e["index"], host[t, TERM["token_loss"]:TERM["token_loss"] + 1]))
            unscaled.append((e["index"], host[t, TERM["unscaled"]:TERM["unscaled"] + 1]))
            if info["loss_type"] == helpers.AnnotationType.BOX:
                helpers.log(f"{state.cur_time_step_iter:02d}.{state.sub_iteration:02d} loss for {info['word']}: "
                            f"{host[t, TERM['unscaled']].item()}")
        loss = fused["loss"]
        if "custom_loss" in losses_dict:
            custom = losses_dict["custom_loss"]
            losses.append((None, fused["host_custom"]))
            unscaled.append((None, fused["host_custom"]))
            loss = loss + custom.to(loss.dtype).reshape(1)
        return loss, losses, unscaled

    def meets_threshold(self, i, thresholds, losses):
        """reference :1074-1088 — every sub-prompt's (unscaled) loss must be <= the step's threshold."""
        _, per_sub = GuidedAttention.group_losses_by_sumprompt(losses)
        if (i not in thresholds and i != -1) or len(thresholds) == 0:
            return True
        thresh = list(thresholds.values())[-1] if i == -1 else thresholds[i]
        for val in per_sub.values():
            if bool(val > thresh):  # fp32 tensor vs Python float: compared in fp32, as in the reference
                return False
        return True

    def _update_latent(self, latents, loss, step_size):
        """latents - step_size * dLoss/dlatents (reference :456-470): autograd back through the UNet
        (K1 / K3+K4 backward kernels inside), then the fused axpy + mean|grad| kernel."""
        runner = self._runner
        if runner is not None and loss is runner.loss:
            grad_cond = runner.backward()  # hipGraph replay of the captured backward pass
        else:
            grad_cond = torch.autograd.grad(loss.requires_grad_(True), [latents], retain_graph=True)[0]
        self.unet_calls["bwd"] += 1
        new_latents, absmean = ops.latent_axpy(latents.detach(), grad_cond, float(step_size), True)
        self._deferred_log.append(("gradient size average: ", absmean))
        return new_latents

    def _guidance_forward(self, latents, t, cond, time_projection=None):
        self.unet_calls["fwd_b1_grad"] += 1
        if self.guidance_forward == "truncated" and self._truncate_at is not None:
            out = self.unet(latents, t, encoder_hidden_states=cond, stop_after_up_block=self._truncate_at,
                            time_projection=time_projection).sample
            if hasattr(self._attention_store, "flush"):
                self._attention_store.flush()
            return out
        return self.unet(latents, t, encoder_hidden_states=cond, time_projection=time_projection).sample

    def _guidance_eval(self, latents, t, cond, attention_store, attention_res, smooth_attentions, sigma, kernel_size,
                       normalize_eot):
        """One guidance evaluation = restart the autograd graph at the latents, UNet forward with capture,
        aggregate, loss.  -> (latents leaf the loss depends on, losses_dict).  Eager, or one hipGraph replay."""
        runner = self._runner
        if runner is not None:
            self.unet_calls["fwd_b1_grad"] += 1
            leaf, parts = runner.evaluate(latents, t, attention_store)
            return leaf, self._loss_host(*parts)
        latents = latents.clone().detach().requires_grad_(True)
        self._guidance_forward(latents, t, cond)
        losses_dict = self._aggregate_and_get_max_attention_per_token(
            attention_store=attention_store, attention_res=attention_res, smooth_attentions=smooth_attentions,
            sigma=sigma, kernel_size=kernel_size, normalize_eot=normalize_eot)
        return latents, losses_dict

    def _perform_iterative_refinement_step(self, latents, loss, threshold, text_embeddings, text_input,
                                           attention_store, step_size, t, attention_res=16, smooth_attentions=True,
                                           sigma=0.5, kernel_size=3, max_refinement_steps=5, normalize_eot=False):
        """Update the latents until every sub-prompt meets the step's threshold or the iteration cap is hit
        (reference :475-581), then one more forward + loss whose graph the caller differentiates."""
        self.inside_iterative_refinement = True
        if state.curHyperParams.get("use_optimizer", False):
            raise NotImplementedError("SGD-momentum refinement (use_optimizer) is off by default and not provided")
        iteration = 0
        state.sub_iteration = iteration
        losses = None
        unscaled_losses = None
        cond1 = text_embeddings[1].unsqueeze(0)
        ev = (attention_store, attention_res, smooth_attentions, sigma, kernel_size, normalize_eot)
        runner = self._runner
        custom = getattr(state.config, "custom_loss", None)
        if runner is not None and self.speculative_refinement and not custom and self._loss_plan(
                smooth_attentions, sigma, kernel_size).T > 0:
            return self._refine_run_ahead(latents, t, cond1, ev, step_size, max_refinement_steps)
        while losses is None or not self.meets_threshold(state.cur_time_step_iter, state.config.thresholds,
                                                         unscaled_losses):
            helpers.log(f"subiteration: {iteration}")
            iteration += 1
            state.sub_iteration = iteration
            latents, losses_dict = self._guidance_eval(latents, t, cond1, *ev)  # restarts the graph at the latents
            loss, losses, unscaled_losses = self._compute_loss(losses_dict, return_losses=True)
            if not self._loss_is_zero(losses_dict):  # reference :551 `elif loss != 0`
                latents = self._update_latent(latents, loss, step_size)
            if iteration >= max_refinement_steps:
                helpers.log(f"\t Exceeded max number of iterations ({max_refinement_steps})! ", self.verbose)
                break
        latents, max_attention_per_index = self._guidance_eval(latents, t, cond1, *ev)
        loss, losses, unscaled_losses = self._compute_loss(max_attention_per_index, return_losses=True)
        helpers.log(f"\t Finished with loss of: {max_attention_per_index['_fused']['host_total'].item()} "
                    f"iter: {iteration}", self.verbose)
        state.sub_iteration = 0
        return loss, latents, max_attention_per_index

    def _refine_run_ahead(self, latents, t, cond1, ev, step_size, max_refinement_steps):
        """The loop of _perform_iterative_refinement_step on the hipGraph runner with the host one evaluation BEHIND the GPU:
        eval_k is enqueued, then — before its loss table is read — backward_k, the latent update and eval_k+1.  The GPU runs
        eval -> backward -> update -> eval ... back to back; the device -> host copy of each table (pinned, asynchronous) and
        the host's threshold logic overlap the next pass (round 3: 215 us + 129 us of idle GPU per iteration,
        profiles/r3_gpu_idle_gaps.md).  Same launches, same order, same counters and log lines as the loop above; the one
        speculated fact is `loss != 0` (reference :551): a loss of exactly 0 discards the enqueued update + evaluation,
        takes their counts back and evaluates the unchanged latents again, as the reference would."""
        runner = self._runner
        iteration = 0
        helpers.log(f"subiteration: {iteration}")
        # the latents each evaluation starts from, as tensors of their own: the leaf an evaluation returns is the runner's ONE
        # static buffer, which the evaluation enqueued ahead overwrites
        current = latents.detach().clone()
        pending = self._guidance_eval_enqueue(current, t, ev[0])
        while True:
            iteration += 1
            state.sub_iteration = iteration
            leaf, parts = pending
            n_log = len(self._deferred_log)
            updated = self._update_latent(leaf, runner.loss, step_size)          # enqueues backward_k + the axpy (a new tensor)
            nxt = self._guidance_eval_enqueue(updated, t, ev[0])                 # enqueues eval_k+1 on the updated latents
            losses_dict = self._loss_host(*parts)                                # waits for eval_k's table only
            loss, losses, unscaled_losses = self._compute_loss(losses_dict, return_losses=True)
            if self._loss_is_zero(losses_dict):                                  # reference :551: no update on a zero loss
                del self._deferred_log[n_log:]
                self.unet_calls["bwd"] -= 1
                self.unet_calls["fwd_b1_grad"] -= 1
                self.discarded_speculations += 1
                nxt = self._guidance_eval_enqueue(current, t, ev[0])             # the unchanged latents, evaluated again
            else:
                current = updated
            pending = nxt
            if iteration >= max_refinement_steps:
                helpers.log(f"\t Exceeded max number of iterations ({max_refinement_steps})! ", self.verbose)
                break
            if self.meets_threshold(state.cur_time_step_iter, state.config.thresholds, unscaled_losses):
                break
            helpers.log(f"subiteration: {iteration}")
        latents, parts = pending                                                 # the final evaluation (:566-578)
        max_attention_per_index = self._loss_host(*parts)
        loss, losses, unscaled_losses = self._compute_loss(max_attention_per_index, return_losses=True)
        helpers.log(f"\t Finished with loss of: {max_attention_per_index['_fused']['host_total'].item()} "
                    f"iter: {iteration}", self.verbose)
        state.sub_iteration = 0
        return loss, latents, max_attention_per_index

    @staticmethod
    def _loss_is_zero(losses_dict):
        return losses_dict["_fused"]["host_total"].item() == 0

    discarded_speculations = 0   # evaluations enqueued ahead of a `loss != 0` test that came out the other way (never counted)

    def _guidance_eval_enqueue(self, latents, t, attention_store):
        """Enqueue one captured guidance evaluation; -> (latents leaf, loss parts with the table still on its way)."""
        self.unet_calls["fwd_b1_grad"] += 1
        return self._runner.evaluate(latents, t, attention_store)

    verbose = False

    # ------------------------------------------------------------------ the denoising loop
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None], attention_store: AttentionStore, attention_res: int = 16,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, negative_prompt: Optional[Union[str, List[str]]] = None,
                 num_images_per_prompt: Optional[int] = 1, eta: float = 0.0,
                 generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: int = 1,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None, max_iter_to_alter: Optional[int] = 25,
                 run_standard_sd: bool = False, thresholds: Optional[dict] = {0: 0.05, 10: 0.5, 20: 0.8},
                 scale_factor: int = 20, scale_range: Tuple[float, float] = (1., 0.5), smooth_attentions: bool = True,
                 sigma: float = 0.5, kernel_size: int = 3, sd_2_1: bool = False,
                 renoise_noise: Optional[List[torch.Tensor]] = None):
        """Same keywords as the reference (:747-777).  Extra, optional: `renoise_noise`, a list of host-generated
        noise tensors consumed by the recurse re-noise step instead of the device generator (RNG parity runs);
        `output_type="latent"` returns the final latents without the VAE."""
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not on the guided-attention path")
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds)
        self.prompt = prompt
        batch_size = 1 if isinstance(prompt, str) else (len(prompt) if prompt is not None else prompt_embeds.shape[0])
        if batch_size * num_images_per_prompt != 1:
            raise NotImplementedError("the guidance pass works on one image (the reference indexes prompt_embeds[1])")
        device = self.device
        if device.type != "cuda":
            raise GaError("GuidedAttention runs on the GPU only (HIP kernels); there is no CPU fallback")
        do_cfg = guidance_scale > 1.0
        text_inputs, prompt_embeds = self._encode_prompt(prompt, device, num_images_per_prompt, do_cfg, negative_prompt,
                                                         prompt_embeds=prompt_embeds,
                                                         negative_prompt_embeds=negative_prompt_embeds)
        state.always_save_iter = [0, 1, 2]
        self.scheduler = DDIMScheduler.from_config(self.scheduler.config)
        self.scheduler.set_timesteps(num_inference_steps, device="cpu")
        timesteps = self.scheduler.timesteps
        acp = self.scheduler.alphas_cumprod
        state.sigmas = (((1 - acp) / acp) ** 0.5).numpy()
        state.timesteps = timesteps
        latents = self.prepare_latents(1, self.unet.in_channels, height, width, prompt_embeds.dtype, device, generator,
                                       latents)
        scale_range = np.linspace(scale_range[0], scale_range[1], len(timesteps))
        if max_iter_to_alter is None:
            max_iter_to_alter = len(timesteps) + 1
        recurse_steps = max(state.curHyperParams.get("recurse_steps", 1), 1)
        recurse_until = state.curHyperParams.get("recurse_until", 20)
        if len(thresholds) == 0:
            thresholds = {0: float("inf")}
        renoise_gen = None
        if recurse_steps > 1 and renoise_noise is None:
            seed = generator.initial_seed() if generator is not None else 0
            renoise_gen = torch.Generator(device).manual_seed(seed)  # deterministic recurse (reference :921)
        renoise_noise = list(renoise_noise) if renoise_noise is not None else None
        if hasattr(attention_store, "attention_res"):
            attention_store.attention_res = attention_res
        self.unet_calls = {"fwd_b1_grad": 0, "bwd": 0, "fwd_b2": 0, "loss_evals": 0, "joint_b3": 0}
        self._attention_store = attention_store
        self._deferred_log = []
        self._deferred_losses = []
        self._truncate_at = self._truncation_point(attention_res, height, width)
        cond = prompt_embeds[1:2] if do_cfg else prompt_embeds[0:1]
        guided = bool(getattr(state.config, "token_dict", None)) or bool(getattr(state.config, "custom_loss", None))
        self._runner = None
        self._dump = bool(self.reference_side_effects or getattr(state.config, "diagnostic_level", 0) > 0)
        # paint-with-words changes the attention kernels' arguments from step to step (sigma_t, on / off): eager
        paint = bool((state.curHyperParams or {}).get("paint_with_words_stop", 0))
        if self.use_graphs and do_cfg and guided and not run_standard_sd and not self._dump and not paint:
            from .graphs import GraphRunner
            self._runner = GraphRunner.for_run(self, attention_store, prompt_embeds, latents, attention_res,
                                               smooth_attentions, sigma, kernel_size, sd_2_1)

        for i, t in enumerate(timesteps):
            t_int = int(t)
            a_t, a_prev = self.scheduler.alphas_for(t_int)
            for recurse_step in range(recurse_steps):
                did_we_update = False
                noise_joint = None
                state.cur_time_step_iter = i
                helpers.log(f"iteration {i}", self.verbose)
                may_update = (not state.config.only_update_on_threshold_steps and i < max_iter_to_alter) or \
                             (i in state.config.thresholds) or (i in thresholds)
                if not guided:
                    # nothing to guide (no annotated token): the reference still runs the guidance forward here,
                    # with no consumer; kept only for call-count fidelity unless skip_unused_guidance is set
                    if not self.skip_unused_guidance:
                        self._guidance_forward(latents, t_int, cond)
                elif (self._runner is not None and self._runner.joint and not may_update and not run_standard_sd
                      and not self.skip_unused_guidance):
                    # The guidance evaluation of this step cannot change the latents (no threshold, no per-step
                    # update): its forward and the CFG pair see the SAME latents and run as one batch-3 pass.  Every
                    # evaluation of the reference is still performed; the loss is computed and logged as before.
                    self.unet_calls["fwd_b1_grad"] += 1
                    self.unet_calls["fwd_b2"] += 1
                    self.unet_calls["joint_b3"] += 1  # of the two counters above, how many ran as one batch-3 pass
                    parts, noise_joint = self._runner.joint_forward(latents, t_int, attention_store)
                    # the loss of such a step is only logged: keep the device values and read them back after the
                    # loop (no host sync here, the host runs ahead of the GPU); the log lines go to their place
                    self._deferred_losses.append((len(helpers.lines), i, parts[:4] + (parts[4].clone(),)))
                elif not (self.skip_unused_guidance and (run_standard_sd or not may_update)):
                    with torch.enable_grad():
                        latents, max_attention_per_index = self._guidance_eval(
                            latents, t_int, cond, attention_store, attention_res, smooth_attentions, sigma,
                            kernel_size, sd_2_1)
                        if not run_standard_sd:
                            loss, losses, unscaled_losses = self._compute_loss(losses_dict=max_attention_per_index)
                            if not self.meets_threshold(i, thresholds, unscaled_losses):
                                did_we_update = True
                                loss, latents, max_attention_per_index = self._perform_iterative_refinement_step(
                                    latents=latents, loss=loss, threshold=thresholds[i], text_embeddings=prompt_embeds,
                                    text_input=text_inputs, attention_store=attention_store,
                                    step_size=scale_factor * np.sqrt(scale_range[i]), t=t_int,
                                    attention_res=attention_res, smooth_attentions=smooth_attentions,
                                    max_refinement_steps=10, sigma=sigma, kernel_size=kernel_size,
                                    normalize_eot=sd_2_1)
                            if (not state.config.only_update_on_threshold_steps and i < max_iter_to_alter) or \
                                    (i in state.config.thresholds):
                                # the reference tests the losses from BEFORE the refinement here (:999)
                                if not self.meets_threshold(-1, state.config.thresholds, unscaled_losses):
                                    did_we_update = True
                                    loss, losses, unscaled_losses = self._compute_loss(
                                        losses_dict=max_attention_per_index)
                                    if max_attention_per_index["_fused"]["host_total"].item() != 0:  # :1002
                                        latents = self._update_latent(latents=latents, loss=loss,
                                                                      step_size=scale_factor * np.sqrt(scale_range[i]))
                                helpers.log(f"Iteration {i} | Loss: "
                                            f"{max_attention_per_index['_fused']['host_total'].item():0.4f}", self.verbose)
                latents = latents.detach()
                # CFG pass with the (possibly updated) latents, no autograd
                if noise_joint is not None:
                    noise_pred = noise_joint
                else:
                    model_in = torch.cat([latents] * 2) if do_cfg else latents
                    model_in = self.scheduler.scale_model_input(model_in, t_int)
                    self.unet_calls["fwd_b2"] += 1
                    if self._runner is not None and do_cfg:
                        noise_pred = self._runner.cfg_forward(latents, t_int, attention_store)
                    else:
                        noise_pred = self.unet(model_in, t_int, encoder_hidden_states=prompt_embeds).sample
                if do_cfg:
                    eps_uncond, eps_text = noise_pred.chunk(2)
                    latents, _x0 = ops.cfg_ddim_step(eps_uncond, eps_text, guidance_scale, latents, a_t, a_prev, self._dump)
                else:
                    latents, _x0 = ops.cfg_ddim_step(noise_pred, noise_pred, 1.0, latents, a_t, a_prev, self._dump)
                if self._dump:  # reference :1031-1037 (each of these synchronises with the device)
                    helpers.log_latent_stats(latents, True)
                    if state.config.diagnostic_level > 1:
                        self.save_image(latents, "xt")
                    if (state.config.diagnostic_level > 0 or state.cur_time_step_iter in state.always_save_iter) \
                            and self.vae is not None:
                        self.save_image(_x0, "pred")
                if callback is not None and i % callback_steps == 0:
                    callback(i, t_int, latents)
                if i > recurse_until or not did_we_update:
                    break
                if recurse_step != recurse_steps - 1:
                    # back to the noise level of step t (reference :1047-1053)
                    prev_timestep = t_int - self.scheduler.config.num_train_timesteps // self.scheduler.num_inference_steps
                    if prev_timestep > 0:
                        Bt = a_t / float(acp[prev_timestep])
                        if renoise_noise is not None:
                            noise = renoise_noise.pop(0).to(device=device, dtype=latents.dtype)
                        else:
                            noise = torch.randn(latents.shape, generator=renoise_gen, device=device).to(latents.dtype)
                        latents = ops.latent_axpby(latents, noise, math.sqrt(Bt), math.sqrt(1 - Bt))

        if self._deferred_log:  # device scalars are read once, after the loop, in ONE device->host copy
            vals = torch.stack([v.detach().reshape(()).float() for _, v in self._deferred_log]).cpu()
            for (text, _), v in zip(self._deferred_log, vals):
                helpers.log(text + str(v.item()))
        self._deferred_log = []
        shift = 0
        if self._deferred_losses:  # likewise the packed loss tables of the log-only steps (same plan: same length)
            host_rows = torch.stack([parts[4] for _, _, parts in self._deferred_losses]).cpu()
            self._deferred_losses = [(pos, step, parts[:4] + (host_rows[j],))
                                     for j, (pos, step, parts) in enumerate(self._deferred_losses)]
        for pos, step, parts in self._deferred_losses:  # loss logs of the log-only steps, inserted where they belong
            state.cur_time_step_iter, state.sub_iteration = step, 0
            mark = len(helpers.lines)
            self._compute_loss(losses_dict=self._loss_host(*parts))
            fresh = helpers.lines[mark:]
            del helpers.lines[mark:]
            helpers.lines[pos + shift:pos + shift] = fresh
            shift += len(fresh)
        self._deferred_losses = []
        has_nsfw_concept = False
        if output_type == "latent":
            image = latents
        else:
            image = self.decode_latents(latents)
            if output_type == "pil":
                image = self.numpy_to_pil(image)
        if not return_dict:
            return (image, has_nsfw_concept)
        return PipelineOutput(images=image, nsfw_content_detected=has_nsfw_concept, latents=latents,
                              unet_calls=dict(self.unet_calls))

    def _truncation_point(self, attention_res, height, width):
        """(up-block index, layers) after which no res^2 cross-attention map is produced any more."""
        lat = height // self.vae_scale_factor
        n_down = len(self.unet.down_blocks)
        last = None
        for i, blk in enumerate(self.unet.up_blocks):
            side = lat // (2 ** (n_down - 1 - i))
            if blk.has_cross_attention and side == attention_res:
                last = (i, len(blk.resnets))
        return last


GuidedAttentionPipeline = GuidedAttention  # the name the task framing uses
