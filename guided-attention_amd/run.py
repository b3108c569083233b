"""Entry points with the reference's names (run.py): `setup`, `load_model`, `parseMetaPrompt`,
`overrideConfig`, `run_on_prompt`, `execute`, the custom-loss plugin API (`CustomLossBase`,
`register_custom_loss`, `ToLeftOf`) and a `main()` CLI taking the RunConfig fields as `--flags` (argparse:
pyrallis is not available here).  `--interactive true` starts the Flask front-end of gui.py."""
import argparse
import dataclasses
import sys
from abc import ABC, abstractmethod
from pathlib import Path
from typing import List

import torch

from .config import RunConfig
from .pipeline_guided_attention import GuidedAttention
from .utils import helpers, ptp_utils, shared_state
from .utils.ptp_utils import AttentionStore


def load_model(config: RunConfig, random_init=None):
    """reference run.py:18-29.  Model ids resolve to local folders only (no network); set
    GA_RANDOM_INIT=1 or pass random_init=True to build seeded random weights of the architecture.
    Under torch.distributed.run (WORLD_SIZE > 1) every rank builds the architecture on its own GPU, only rank 0
    reads / initialises the weights, and ONE bucketed RCCL broadcast over xGMI hands them to the other ranks."""
    import os
    from . import parallel
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the guided-attention path runs on HIP kernels only")
    rank, world, local = parallel.init_distributed()
    device = torch.device("cuda", local if "LOCAL_RANK" in os.environ else 0)
    name = "stabilityai/stable-diffusion-2-1-base" if config.sd_2_1 else "CompVis/stable-diffusion-v1-4"
    name = os.environ.get("GA_MODEL_DIR", name)
    if random_init is None:
        random_init = os.environ.get("GA_RANDOM_INIT", "0") == "1"
    revision = "fp16" if config.half_precision else None
    stable = GuidedAttention.from_pretrained(name, revision=revision, random_init=random_init, weights=(rank == 0))
    stable = stable.to(device)
    if world > 1:
        for module in (stable.unet, stable.vae, stable.text_encoder):
            if isinstance(module, torch.nn.Module):
                parallel.broadcast_module_(module)
    # the UNet passes replay as hipGraphs (captured in the first image's warm-up); GA_EAGER=1 launches every kernel
    # from the host instead (same kernels, same results, ~3x slower at batch 1)
    stable.use_graphs = os.environ.get("GA_EAGER", "0") != "1"
    return stable


def run_on_prompt(prompt: List[str], model: GuidedAttention, controller: AttentionStore, seed: torch.Generator,
                  config: RunConfig, **extra):
    if controller is not None:
        ptp_utils.register_attention_control(model, controller)
    outputs = model(prompt=prompt, attention_store=controller, attention_res=config.attention_res,
                    guidance_scale=config.guidance_scale, generator=seed,
                    num_inference_steps=config.n_inference_steps, max_iter_to_alter=config.max_iter_to_alter,
                    run_standard_sd=config.run_standard_sd, thresholds=config.thresholds,
                    scale_factor=config.scale_factor, scale_range=config.scale_range,
                    smooth_attentions=config.smooth_attentions, sigma=config.sigma, kernel_size=config.kernel_size,
                    sd_2_1=config.sd_2_1, **extra)
    return outputs.images[0] if not extra else outputs


def get_indices(tokenized_prompt, tokens):
    n = len(tokens)
    for i in range(0, len(tokenized_prompt) - n):
        if tokenized_prompt[i:i + n] == tokens:
            return list(range(i, i + n))
    return None  # the reference falls through the same way; parseMetaPrompt then raises TypeError


def overrideConfig(config):
    if "meta_prompt" in shared_state.curHyperParams:
        config.meta_prompt = shared_state.curHyperParams["meta_prompt"]
    if "thresholds" in shared_state.curHyperParams:
        config.thresholds = shared_state.curHyperParams["thresholds"]


def parseMetaPrompt(config):
    config.prompt, config.meta_info, config.custom_loss = helpers.parse_prompt(config.meta_prompt)
    shared_state.config = config
    tok = config.stable.tokenizer
    tokenized_prompt = tok(config.prompt)["input_ids"]
    token_dict = {}
    for phrase, kind, geom in config.meta_info:
        tokens = tok(phrase)["input_ids"][1:-1]
        for idx in get_indices(tokenized_prompt, tokens):
            token_dict[idx] = {"word": tok.decode(tokenized_prompt[idx]), "loss_type": kind, "loss": geom,
                               "subprompt": phrase}
    config.token_dict = token_dict


def execute(config, save=True):
    """One image per (seed, hyper-parameter state) (reference run.py:93-135).  The reference runs them serially on one
    device; images of different (seed, state) are independent, so under torch.distributed.run the job list is striped
    over the ranks (job j on rank j % world, one process per GPU, no per-step exchange) and rank 0 gathers the final
    latents and images back into job order.  Single process: exactly the reference's serial loop.
    Returns the path of the last job's image (as the reference does); on rank 0 `shared_state.last_results` holds
    {"latents": [...], "images": [...]} in job order (None on the other ranks)."""
    from . import parallel
    from .utils import vis_utils
    rank, world = parallel.rank_world()
    jobs = [(seed, hp) for seed in config.seeds for hp in shared_state.get_hyperparam_states()]
    images, latents, image_path, paths = [], [], None, {}
    for j, (seed, hp) in enumerate(jobs):
        shared_state.curHyperParams = hp
        overrideConfig(config)
        parseMetaPrompt(config)
        out_dir = config.output_path / helpers.get_inner_folder_name()
        name = helpers.dictToString(shared_state.curHyperParams)
        paths[j] = out_dir / f"{seed}{name}.png"
        if j % world != rank:
            continue
        helpers.log_clear()
        shared_state.cur_seed = seed
        print(f"Seed: {seed}")
        g = torch.Generator(config.stable.device).manual_seed(seed)
        controller = AttentionStore()
        out = run_on_prompt(prompt=config.prompt, model=config.stable, controller=controller, seed=g, config=config,
                            output_type="pil")
        image = out.images[0]
        images.append(image)
        latents.append(out.latents.detach())
        if save:
            out_dir.mkdir(exist_ok=True, parents=True)
            helpers.annotate_image(image)
            try:
                image.save(paths[j])
            except OSError:
                print("bad path. this is often due to exceeding max path length.")
                name = ""
                paths[j] = out_dir / f"{seed}.png"
                image.save(paths[j])
            helpers.log_save(out_dir / f"{seed}{name}.txt")
            helpers.save_latent_stats(out_dir / f"{seed}{name}figure.png")
    if jobs:
        image_path = paths[len(jobs) - 1]
    if world > 1:
        import numpy as np
        from PIL import Image
        dev = config.stable.device
        pix = [torch.from_numpy(np.asarray(im.convert("RGB")).copy()).to(dev) for im in images]
        g_lat, g_pix = parallel.gather_tensors(latents), parallel.gather_tensors(pix)
        if rank == 0:
            latents = parallel.unstripe(g_lat)
            images = [Image.fromarray(t.cpu().numpy()) for t in parallel.unstripe(g_pix)]
    shared_state.last_results = {"latents": latents, "images": images} if rank == 0 else None
    if rank == 0 and save and images:
        joined = vis_utils.get_image_grid(images)  # a grid of the results across all seeds (reference :131-134)
        if not config.interactive:
            helpers.annotate_image(joined)
        joined.save(config.output_path / f"{helpers.get_meta_prompt_clean()}.png")
    return image_path


skipLoading = False


def setup(config, random_init=None):
    shared_state.config = config
    if not skipLoading:
        config.stable = load_model(config, random_init=random_init)


class CustomLossBase(ABC):
    """Plugin API of the reference (run.py:148-176): a Python loss over the (H, W, n_text_tokens) map
    softmax(100 * A[:, :, 1:last]); it is added to the fused loss and differentiated by autograd."""

    @abstractmethod
    def calc_loss(self, cross_attention_maps, text_args: str) -> torch.Tensor:
        pass

    def subprompts_of_interest(self, text_args: str) -> list:
        return []

    def parse_text_args(self, text_args: str):
        import ast
        return ast.literal_eval(text_args)

    def find_indices_for_sub_prompt(self, sub_prompt):
        tok = shared_state.config.stable.tokenizer
        full = tok(shared_state.config.prompt)["input_ids"][1:-1]
        sub = tok(sub_prompt)["input_ids"][1:-1]
        for i in range(len(full) - len(sub) + 1):
            if full[i:i + len(sub)] == sub:
                return list(range(i, i + len(sub)))

    def get_map_for_token(self, cross_attention_maps, token_index: int, pixel_wise_normalization=True):
        image_map = cross_attention_maps[:, :, token_index]
        return image_map / image_map.sum() if pixel_wise_normalization else image_map


class ToLeftOf(CustomLossBase):
    """`[CustomLoss:toLeftOf (a, b)]`: the attention centroid of sub-prompt a must sit at least 20 % of the map width
    to the left of b's (reference run.py:180-225, including its normalisation of BOTH centroids by the token count
    of the left sub-prompt).  Plain PyTorch on the GPU map; differentiated by autograd and added to the fused loss."""

    def calc_loss(self, cross_attention_maps, text_args: str) -> torch.Tensor:
        args = self.parse_text_args(self.quote_items_in_tuple(text_args))
        left = self.find_indices_for_sub_prompt(args[0])
        right = self.find_indices_for_sub_prompt(args[1])
        width = cross_attention_maps.shape[1]
        cols = torch.arange(width, device=cross_attention_maps.device, dtype=cross_attention_maps.dtype) + 0.5

        def centroid(idx):
            total = cross_attention_maps.new_zeros(1)
            for i in idx:
                total = total + (self.get_map_for_token(cross_attention_maps, i, True) * cols[None, :]).sum() / len(left)
            return total

        gap = .2 * width
        loss = (centroid(left) + gap - centroid(right)) / width * 9
        return torch.clamp(loss, min=0)

    def subprompts_of_interest(self, text_args: str) -> list:
        return list(self.parse_text_args(self.quote_items_in_tuple(text_args)))

    def quote_items_in_tuple(self, text_args):
        items = text_args.strip("()").split(",")
        return "(" + ",".join(f"'{item.strip()}'" for item in items) + ")"


def register_custom_loss(name: str, customLoss: CustomLossBase):
    if not hasattr(shared_state.config, "registered_loss_functions"):
        shared_state.config.registered_loss_functions = {}
    shared_state.config.registered_loss_functions[name] = customLoss


def _parse_cli(argv):
    ap = argparse.ArgumentParser(description="guided-attention sampling on MI355X")
    for f in dataclasses.fields(RunConfig):
        flag = f"--{f.name}"
        if f.type is bool or isinstance(f.default, bool):
            ap.add_argument(flag, type=lambda s: s.lower() in ("1", "true", "yes"), default=f.default)
        elif f.name == "seeds":
            ap.add_argument(flag, type=lambda s: [int(x) for x in s.strip("[]").split(",")], default=[42])
        elif f.name == "thresholds":
            ap.add_argument(flag, type=lambda s: {int(k): float(v) for k, v in (kv.split(":") for kv in s.strip("{}").split(","))},
                            default={0: 0.1, 3: 0.8})
        elif f.name == "scale_range":
            ap.add_argument(flag, type=lambda s: tuple(float(x) for x in s.strip("()").split(",")), default=(1.0, 0.5))
        elif f.name == "output_path":
            ap.add_argument(flag, type=Path, default=Path("./outputs"))
        elif f.default is dataclasses.MISSING and f.default_factory is dataclasses.MISSING:
            ap.add_argument(flag, type=str, required=True)
        else:
            ap.add_argument(flag, type=type(f.default), default=f.default)
    return RunConfig(**vars(ap.parse_args(argv)))


def main(argv=None):
    config = _parse_cli(sys.argv[1:] if argv is None else argv)
    setup(config)
    register_custom_loss("toLeftOf", ToLeftOf())  # as the reference's main() does (run.py:240)
    if config.interactive:   # reference run.py:242-244
        from . import gui
        gui.run()
    else:
        execute(config)


if __name__ == "__main__":
    main()
