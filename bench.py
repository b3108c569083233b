#!/usr/bin/env python3
"""bench.py — guided images/sec of the guided-attention hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one guided image: the full 50-step DDIM guided sampling of the SD-1.x UNet at 512^2
(latent 64^2), prompt 'a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]', guidance 7.5, the
reference's shipped hyper-parameters (thresholds {0:1.0}, recurse_steps 3, max 10 refinement
iterations), fp16, timed from after prompt encoding to the final latents (the VAE
decode is outside the metric).  Synthetic inputs: seeded random-init UNet weights (no checkpoint
offline), synthetic prompt embeddings, host-generated initial latents and re-noise tensors.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

META_PROMPT = "a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]"
META_PROMPT_SD21 = "a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55] under a [moon:.35,.05,.35,.35]"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed guided images per GPU")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="default", choices=["default", "every-step"])
    ap.add_argument("--model", default=os.environ.get("GA_BENCH_MODEL", "sd15"), choices=["sd15", "sd21", "tiny"],
                    help="sd21 = BASELINE config 4: SD-2.1 UNet shapes at 768^2 (latent 96^2, maps 24x24), 3 box tokens")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--guidance-forward", default="full", choices=["full", "truncated"])
    ap.add_argument("--skip-unused-guidance", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying hipGraphs")
    ap.add_argument("--no-joint-pass", action="store_true",
                    help="run the loss-only guidance forward and the CFG pair as two passes (B=1, B=2) instead of one B=3 pass")
    return ap.parse_args()


def build_pipeline(args, device, rank, world):
    from guided_attention_amd import parallel
    from guided_attention_amd.pipeline_guided_attention import GuidedAttention
    from guided_attention_amd.text import SyntheticTextEncoder, WordTokenizer
    from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
    cfg = {"sd15": UNetConfig.sd15, "sd21": UNetConfig.sd21}.get(args.model, lambda: UNetConfig.tiny(64, 768))()
    with torch.device(device):
        unet = UNet2DConditionModel(cfg)
    unet = unet.half()
    if rank == 0:
        unet.init_weights_(seed=0)       # seeded random weights of the SD-1.x architecture
    t0 = time.perf_counter()
    n_msgs = parallel.broadcast_module_(unet)  # RCCL over xGMI: the only start-up collective
    torch.cuda.synchronize()
    bcast_s = time.perf_counter() - t0
    pipe = GuidedAttention(unet, None, None, SyntheticTextEncoder(cfg.cross_attention_dim), WordTokenizer())
    pipe.to(device, torch.float16)
    pipe.guidance_forward = args.guidance_forward
    pipe.skip_unused_guidance = args.skip_unused_guidance
    pipe.use_graphs = not args.eager
    pipe.batch_loss_only_guidance = not args.no_joint_pass
    return pipe, cfg, {"messages": n_msgs, "seconds": bcast_s}


def make_run(args, pipe, cfg, device):
    from guided_attention_amd import run
    from guided_attention_amd.config import RunConfig
    from guided_attention_amd.utils import helpers, ptp_utils, shared_state as state
    rc = RunConfig(meta_prompt=META_PROMPT_SD21 if args.model == "sd21" else META_PROMPT,
                   output_path="/tmp/ga_bench_out", half_precision=True, n_inference_steps=args.ddim_steps)
    if args.model == "sd21":
        rc.attention_res = cfg.sample_size // 4   # the 24x24 maps of the 96^2 latent (synthetic embeddings: no EOT slice)
    if args.workload == "every-step":
        rc.only_update_on_threshold_steps = False
        rc.max_iter_to_alter = 25
    rc.stable = pipe
    state.curHyperParams = state.get_hyperparam_states()[0]
    run.overrideConfig(rc)        # thresholds := {0: 1.0}, as the reference does at run time
    run.parseMetaPrompt(rc)
    g = torch.Generator("cpu").manual_seed(1234)
    embeds = torch.randn(2, 77, cfg.cross_attention_dim, generator=g).to(device, torch.float16)
    lat_side = cfg.sample_size

    def one_image(seed):
        gs = torch.Generator("cpu").manual_seed(seed)
        latents = torch.randn(1, 4, lat_side, lat_side, generator=gs)
        noise = [torch.randn(1, 4, lat_side, lat_side, generator=gs) for _ in range(2 * args.ddim_steps)]
        helpers.log_clear()
        state.cur_seed = seed
        controller = ptp_utils.AttentionStore()
        ptp_utils.register_attention_control(pipe, controller)
        return pipe(prompt=None, prompt_embeds=embeds[1:2], negative_prompt_embeds=embeds[0:1],
                    attention_store=controller, attention_res=rc.attention_res, guidance_scale=rc.guidance_scale,
                    num_inference_steps=rc.n_inference_steps, max_iter_to_alter=rc.max_iter_to_alter,
                    thresholds=rc.thresholds, scale_factor=rc.scale_factor, scale_range=rc.scale_range,
                    smooth_attentions=rc.smooth_attentions, sigma=rc.sigma, kernel_size=rc.kernel_size,
                    latents=latents, renoise_noise=noise, output_type="latent")

    return one_image, rc, embeds


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes a container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16


def kernel_work(key):
    """Algorithmic work of ONE call of a hand-written kernel: ("hbm", bytes) or ("mfma", flops).
    capture fwd  = Q + O + K + V (+ P when stored);  capture bwd = Q + dO + dQ + K + V (+ the broadcast dP map)
    self-attn    = 4 N^2 D flops per head forward, 2.5x that for the backward (5 matrix products, recompute not counted)
    GroupNorm    = read x + write y (fwd), read x, dy + write dx (bwd)."""
    kind, B, H, N, Kt, D, flag, dt = key
    esz = 4 if dt == "torch.float32" else 2
    if kind in ("geglu_fwd", "geglu_bwd"):            # B = rows, D = F: x [rows][2F] in, y [rows][F] out (bwd: + dy, dx)
        return "hbm", esz * B * D * (3 if kind == "geglu_fwd" else 5)
    if kind == "bias_residual_add":                  # y + residual in, out
        return "hbm", esz * B * D * 3
    if kind == "add_layer_norm_fwd":                 # (a,) x in; (x_new,) y out
        return "hbm", esz * B * D * (4 if flag else 2)
    if kind == "add_layer_norm_bwd":                 # row, dy (, g_res) in; dx out
        return "hbm", esz * B * D * (4 if flag else 3)
    if kind.startswith("group_norm"):
        elems = B * N * D  # here H = groups, N = pixels, D = channels
        return "hbm", esz * elems * (2 if kind == "group_norm_fwd" else 3)
    C = H * D
    if kind == "attn_capture_fwd":
        return "hbm", esz * (2 * B * N * C + 2 * B * Kt * C + (B * H * N * Kt if flag else 0))
    if kind == "attn_capture_bwd":
        return "hbm", esz * (3 * B * N * C + 2 * B * Kt * C + (N * Kt if flag else 0))
    flops = 4.0 * B * H * N * N * D
    return "mfma", flops * (1.0 if kind == "self_attn_fwd" else 2.5)


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel's largest shape from the committed PMC run (FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc passes, gfx950 correction applied: profiles/r1_pmc_self_attn.json);
    None when no PMC profile of that kernel is committed."""
    try:
        doc = json.loads((ROOT / "profiles" / "r1_pmc_self_attn.json").read_text())
        k = doc["kernels"][kernel]
        return {"hbm_bytes": k["hbm_bytes_corrected"], "algorithmic_bytes": k["algorithmic_bytes"], "shape": doc["shape"],
                "source": "profiles/r1_pmc_self_attn.json"}
    except (OSError, KeyError, ValueError):
        return None


def roofline_entry(census, ops):
    """Dominant hand-written kernel = the ga_* entry point with the largest total time over the timed region.
    achieved = sum of algorithmic work over its calls / sum of their durations; each shape's duration is measured by a
    back-to-back hipGraph replay between two HIP events on the launch stream (the weighted mean equals what
    `rocprofv3 --kernel-trace --stats` reports as the kernel's AverageNs: profiles/)."""
    per_kind = {}
    for key, count in census.items():
        us = ops.replay_launch_us(key)
        bound, work = kernel_work(key)
        d = per_kind.setdefault(key[0], {"bound": bound, "time_us": 0.0, "work": 0.0, "calls": 0, "shapes": []})
        d["time_us"] += us * count
        d["work"] += work * count
        d["calls"] += count
        rate = work / us / 1e3 if bound == "hbm" else work / us / 1e6  # GB/s or TFLOP/s
        d["shapes"].append({"B": key[1], "H": key[2], "N": key[3], "D": key[5], "flag": key[6], "calls": count,
                            "call_us": round(us, 2), "rate": round(rate, 1)})
    if not per_kind:
        return None

    def summary(kind, d):
        if d["bound"] == "hbm":
            achieved, peak, unit = d["work"] / d["time_us"] / 1e3, HBM_PEAK_GBS, "GB/s"
        else:
            achieved, peak, unit = d["work"] / d["time_us"] / 1e6, MFMA_PEAK_TFLOPS, "TFLOP/s"
        return {"bound": d["bound"], "kernel": "ga_" + kind, "achieved": round(achieved, 1), "peak": peak, "unit": unit,
                "frac": round(achieved / peak, 4), "avg_call_us": round(d["time_us"] / d["calls"], 2),
                "calls": d["calls"], "total_ms": round(d["time_us"] / 1e3, 2)}

    kind, d = max(per_kind.items(), key=lambda kv: kv[1]["time_us"])
    out = summary(kind, d)
    out["traffic"] = pmc_traffic("ga_" + kind)
    out["shapes"] = sorted(d["shapes"], key=lambda x: -x["calls"] * x["call_us"])[:6]
    out["other_kernels"] = [summary(k, v) for k, v in sorted(per_kind.items(), key=lambda kv: -kv[1]["time_us"]) if k != kind]
    return out


def cpu_baseline(args, cfg, calls, rc):
    """The oracle (CPU fp32 restatement, oracle/pipeline.py) on the host cores: one guidance evaluation
    (forward with autograd + loss), one backward to the latents and one CFG forward of the SAME UNet
    shape, extrapolated with the GPU run's per-image call counts."""
    import copy
    from oracle import loss as oloss
    from oracle.pipeline import GuidedSampler
    from guided_attention_amd.unet import UNet2DConditionModel
    cores = usable_cores()
    torch.set_num_threads(cores)
    unet = UNet2DConditionModel(cfg).init_weights_(seed=0).float()
    for p in unet.parameters():
        p.requires_grad_(False)
    entries = [{"index": 2, "kind": "BOX", "geom": (.6, .3, .4, .55), "subprompt": "robot"},
               {"index": 5, "kind": "BOX", "geom": (.2, .3, .4, .55), "subprompt": "blue vase"},
               {"index": 6, "kind": "BOX", "geom": (.2, .3, .4, .55), "subprompt": "blue vase"}]
    s = GuidedSampler(unet, oloss.TokenPlan(entries), thresholds={0: 1.0}, steps=args.ddim_steps)
    g = torch.Generator("cpu").manual_seed(1234)
    embeds = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    lat = torch.randn(1, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator("cpu").manual_seed(0))
    t0 = time.perf_counter()
    with torch.enable_grad():
        lat_g, r, _ = s._evaluate(lat, 981, embeds[1:2])
    t1 = time.perf_counter()
    with torch.enable_grad():
        s._update(lat_g, r["loss"], 20.0)
    t2 = time.perf_counter()
    with torch.no_grad():
        unet(torch.cat([lat] * 2), 981, encoder_hidden_states=embeds)
    t3 = time.perf_counter()
    per = {"fwd_b1_grad": t1 - t0, "bwd": t2 - t1, "fwd_b2": t3 - t2}
    sec_per_image = sum(per[k] * calls[k] for k in per)
    return {"value": 1.0 / sec_per_image, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "1x guidance forward+loss (autograd), 1x backward to latents, 1x CFG forward (B=2) of the same "
                      f"{args.model} UNet in fp32 on the host, extrapolated with the GPU run's per-image call counts "
                      f"{ {k: calls[k] for k in per} }",
            "seconds_per_kind": {k: round(v, 3) for k, v in per.items()}, "seconds_per_image": round(sec_per_image, 1)}


def main():
    args = parse()
    from guided_attention_amd import ops, parallel
    rank, world, local = parallel.init_distributed()
    if world > 1:
        # one MIOpen user database / kernel cache per rank: N processes benchmarking the same conv shapes at the same
        # time otherwise queue on the file locks of one shared database
        os.environ.setdefault("MIOPEN_USER_DB_PATH", f"/tmp/ga_miopen_{os.getuid()}_{local}")
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", f"/tmp/ga_miopen_{os.getuid()}_{local}/cache")
        os.makedirs(os.environ["MIOPEN_CUSTOM_CACHE_DIR"], exist_ok=True)
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the guided-attention path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    ops.load()
    pipe, cfg, bcast = build_pipeline(args, device, rank, world)
    one_image, rc, _ = make_run(args, pipe, cfg, device)
    import torch.distributed as dist

    seed_of = lambda j: rank + world * j   # seeds striped by rank (weak scaling: K images per GPU)
    for j in range(args.warmup):
        one_image(1000 + seed_of(j))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.start_census()
    t0 = time.perf_counter()
    calls = None
    finals = []
    for j in range(args.steps):
        out = one_image(seed_of(j))
        finals.append(out.latents)
        calls = out.unet_calls
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    census = ops.stop_census()
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    gathered = parallel.gather_tensors(finals)  # the end-of-run gather of the final latents (32 KB each)
    ok = all(torch.isfinite(f).all().item() for f in finals)
    if rank == 0:
        n_images = args.steps * world
        roof = roofline_entry(census, ops)
        flops_per_fwd = {"sd15": 0.803e12, "sd21": 2.149e12}.get(args.model)
        line = {
            "metric": ("guided images/sec (50-step SD-2.1 768^2)" if args.model == "sd21" else
                       "guided images/sec (50-step SD-1.5 512^2)"), "value": n_images / elapsed, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"W-{args.workload}: {'SD-2.1 UNet 768^2 (latent 96^2)' if args.model == 'sd21' else 'SD-1.x UNet 512^2 (latent 64^2)'}, {args.ddim_steps} DDIM steps, "
                                   f"meta_prompt '{rc.meta_prompt}', guidance 7.5, thresholds {rc.thresholds}, 1 seed per step",
                       "parallelism": f"seed-parallel x{world}", "guidance_forward": args.guidance_forward,
                       "skip_unused_guidance": args.skip_unused_guidance, "model": args.model,
                       "loss_only_steps": ("two passes (B=1 guidance, B=2 CFG)" if args.no_joint_pass or args.eager else
                                           "guidance forward + CFG pair of a step without latent update batched as one "
                                           "B=3 pass (every evaluation performed)"),
                       "launch": "eager" if args.eager else "hipGraph replay of the UNet passes (captured in warm-up)",
                       "weights": "seeded random init (no checkpoint offline)"},
            "unet_calls_per_image": calls, "finite": ok,
            "weight_broadcast": bcast,
        }
        if flops_per_fwd:
            tf = flops_per_fwd * (calls["fwd_b1_grad"] + 2 * calls["fwd_b2"] + calls["bwd"]) / 1e12
            line["end_to_end"] = {"tflop_per_image": round(tf, 1),
                                  "achieved_tflops_per_gpu": round(tf * args.steps / elapsed, 1),
                                  "frac_of_2.5PF_dense_fp16": round(tf * args.steps / elapsed / 2500.0, 4)}
        line["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, cfg, calls, rc)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
