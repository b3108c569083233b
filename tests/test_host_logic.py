"""CPU tests of the product's HOST layer (no kernels run): the reference-named mirrors under
guided_attention_amd/ against the fixtures produced by the reference (tests/golden) and its documented defaults."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_json, load_npz
from guided_attention_amd import run
from guided_attention_amd.config import RunConfig
from guided_attention_amd.pipeline_guided_attention import GuidedAttention, GuidedAttentionPipeline
from guided_attention_amd.scheduler import DDIMScheduler
from guided_attention_amd.text import WordTokenizer
from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
from guided_attention_amd.utils import helpers, ptp_utils, shared_state as state


class _Custom:
    def subprompts_of_interest(self, args):
        return [a.strip() for a in args.strip("()").split(",")]


def _ser(meta_info):
    out = []
    for tok, typ, val in meta_info:
        v = [val.x, val.y, val.width, val.height, val.size] if typ.name == "BOX" else (list(val) if typ.name == "COOR" else None)
        out.append([tok, typ.name, v])
    return out


@pytest.mark.parametrize("case", load_json("g2_parse_prompt.json"), ids=lambda c: c["meta_prompt"][:28])
def test_parse_prompt_matches_reference(case):
    state.config = SimpleNamespace(registered_loss_functions={"toLeftOf": _Custom()})
    prompt, meta, custom = helpers.parse_prompt(case["meta_prompt"])
    assert prompt == case["prompt"]
    assert _ser(meta) == case["meta_info"]
    assert {k: v[1] for k, v in custom.items()} == case["custom_losses"]


def test_parse_prompt_errors_like_the_reference():
    state.config = SimpleNamespace(registered_loss_functions={})
    with pytest.raises(ValueError):
        helpers.parse_prompt("a [cat .2,.3]")          # no colon: str.index raises ValueError
    with pytest.raises(KeyError):
        helpers.parse_prompt("a [CustomLoss:nope (x)]")  # unregistered custom loss


def test_inside_box_and_rect_match_reference():
    cases = load_json("g3_inside_box.json")
    arrs = load_npz("g3_inside_box.npz")
    for c in cases:
        state.curHyperParams = dict(state.hyperParameterOverrides, shrink_factor=c["shrink"])
        rect = helpers.Rect(*c["rect"], 1).of_size(float(c["res"]))
        assert [rect.x, rect.y, rect.width, rect.height] == c["scaled"]
        assert list(helpers.Rect(*c["rect"], 1).center()) == c["center"]
        mask = helpers.inside_mask(rect, c["res"]).numpy().astype(np.uint8)
        assert np.array_equal(mask, arrs[f"mask{c['id']}"]), c


def test_calculate_bounding_box_losses_cpu_tensor():
    state.curHyperParams = dict(state.hyperParameterOverrides)
    g = load_npz("g4_loss.npz")  # recompute inside/outside of one token from the reference's own smoothed map is not
    P = torch.rand(16, 16)       # stored; check the defining identities instead
    P = P / P.sum()
    inside, outside = helpers.calculate_bounding_box_losses(helpers.Rect(.6, .3, .4, .55, 1).of_size(16.0), P)
    mask = helpers.inside_mask(helpers.Rect(.6, .3, .4, .55, 1).of_size(16.0), 16)
    assert mask.sum() == 24
    np.testing.assert_allclose(float(inside), 1 - float(P[mask].sum()), rtol=1e-6)
    np.testing.assert_allclose(float(inside), float(outside), rtol=1e-5)  # both equal 1 - mass inside
    # strict mode (helpers.py:216-264) against the oracle's pixel-loop restatement (itself pinned to g4 strict_*)
    from oracle import loss as oloss
    state.curHyperParams["strict"] = True
    s_in, s_out = helpers.calculate_bounding_box_losses(helpers.Rect(.6, .3, .4, .55, 1).of_size(16.0), P)
    Wn, m2, n_in = oloss.strict_weights((.6, .3, .4, .55), 16, state.curHyperParams["shrink_factor"])
    assert n_in == 24 and np.array_equal(m2, mask.numpy())
    ref_in = (Wn * 2 * np.maximum(1.0 / n_in - P.numpy(), 0))[m2].sum()
    ref_out = (Wn * P.numpy())[~m2].sum()
    np.testing.assert_allclose([float(s_in), float(s_out)], [ref_in, ref_out], rtol=2e-6)
    state.curHyperParams["strict"] = False


def test_meets_threshold_and_grouping_match_reference():
    doc = load_json("g5_meets_threshold.json")
    state.config = SimpleNamespace(token_dict={int(t): {"subprompt": s} for t, s in doc["token_subprompt"].items()},
                                   sub_prompt_avg_within=False)
    pipe = GuidedAttention.__new__(GuidedAttention)
    for row in doc["rows"]:
        thr = {int(k): v for k, v in doc["thr_sets"][row["thresholds"]].items()}
        losses = [(t, torch.tensor([v], dtype=torch.float32)) for t, v in doc["loss_sets"][row["losses"]]]
        assert pipe.meets_threshold(row["i"], thr, losses) == row["result"], row
    total, per = GuidedAttention.group_losses_by_sumprompt([(2, torch.tensor([.5])), (5, torch.tensor([.25])), (6, torch.tensor([.25]))])
    assert float(total) == 1.0 and {k: float(v) for k, v in per.items()} == {"robot": .5, "blue vase": .5}
    state.config.sub_prompt_avg_within = True
    total, per = GuidedAttention.group_losses_by_sumprompt([(2, torch.tensor([.5])), (5, torch.tensor([.25])), (6, torch.tensor([.75]))])
    assert float(total) == 1.0 and float(per["blue vase"]) == 0.5


def test_run_config_and_hyper_parameter_defaults():
    cfg = RunConfig(meta_prompt="x", output_path="/tmp/ga_test_out")
    assert (cfg.seeds, cfg.n_inference_steps, cfg.guidance_scale, cfg.max_iter_to_alter, cfg.attention_res) == ([42], 50, 7.5, 25, 16)
    assert cfg.thresholds == {0: 0.1, 3: 0.8} and cfg.scale_factor == 20 and cfg.scale_range == (1.0, 0.5)
    assert (cfg.smooth_attentions, cfg.sigma, cfg.kernel_size) == (True, 0.5, 3)
    assert cfg.only_update_on_threshold_steps is True and cfg.sub_prompt_avg_within is False and cfg.sd_2_1 is False
    assert state.hyperParameterOverrides == {"strict": False, "inside_loss_scale": .2, "outside_loss_scale": .2,
                                             "shrink_factor": .15, "thresholds": {0: 1.}, "use_optimizer": False,
                                             "recurse_until": 14, "recurse_steps": 3}
    assert state.get_hyperparam_states() == [state.hyperParameterOverrides]
    # overrideConfig: the hyper-parameter thresholds REPLACE the CLI ones (reference run.py:75-79)
    state.curHyperParams = state.get_hyperparam_states()[0]
    run.overrideConfig(cfg)
    assert cfg.thresholds == {0: 1.0}
    cli = run._parse_cli(["--meta_prompt", "a [cat:.1,.2]", "--seeds", "[3,4]", "--thresholds", "{0:0.2,5:0.7}",
                          "--half_precision", "true", "--output_path", "/tmp/ga_test_out"])
    assert cli.seeds == [3, 4] and cli.thresholds == {0: 0.2, 5: 0.7} and cli.half_precision is True


def test_parse_meta_prompt_builds_the_token_dict():
    cfg = RunConfig(meta_prompt="a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]", output_path="/tmp/ga_test_out")
    cfg.stable = SimpleNamespace(tokenizer=WordTokenizer())
    run.parseMetaPrompt(cfg)
    assert cfg.prompt == "a robot and a blue vase"
    assert sorted(cfg.token_dict) == [2, 5, 6]  # robot = 2, blue = 5, vase = 6
    assert cfg.token_dict[5]["subprompt"] == cfg.token_dict[6]["subprompt"] == "blue vase"
    assert cfg.token_dict[2]["loss_type"] == helpers.AnnotationType.BOX and cfg.token_dict[2]["word"] == "robot"
    bad = RunConfig(meta_prompt="a [unicorn horn:.1,.2] b", output_path="/tmp/ga_test_out")
    bad.stable = SimpleNamespace(tokenizer=WordTokenizer())
    bad.meta_prompt = "[zebra:.1,.2]"
    run.parseMetaPrompt(bad)  # single word: found
    bad.meta_prompt = "a [b c:.1,.2] d e"
    run.parseMetaPrompt(bad)
    assert sorted(bad.token_dict) == [2, 3]


def test_attention_store_bookkeeping_without_kernels():
    state.config = SimpleNamespace(save_individual_CA_maps=False)
    for capture, kept in (("reference", 3), ("loss-only", 1)):
        store = ptp_utils.AttentionStore(capture=capture)
        store.num_att_layers = 4
        assert store.wants_probs(True, 256) and (store.wants_probs(False, 1024) == (capture == "reference"))
        assert not store.wants_probs(True, 4096)
        calls = [(torch.zeros(8, 256, 77), True, "down"), (ptp_utils.ProbsNotCaptured((8, 4096, 77), torch.float16, "cpu"), True, "up"),
                 (torch.zeros(8, 1024, 1024) if capture == "reference" else ptp_utils.ProbsNotCaptured((8, 1024, 1024), torch.float16, "cpu"), False, "up"),
                 (torch.zeros(8, 64, 77) if capture == "reference" else ptp_utils.ProbsNotCaptured((8, 64, 77), torch.float16, "cpu"), True, "mid")]
        for probs, is_cross, place in calls:
            store(probs, is_cross, place)
        assert store.cur_step == 1 and store.cur_att_layer == 0
        assert sum(len(v) for v in store.attention_store.values()) == kept
        assert store.step_store == ptp_utils.AttentionStore.get_empty_store()
        store.reset()
        assert store.attention_store == {} and store.cur_step == 0
    with pytest.raises(ValueError):
        ptp_utils.AttentionStore(capture="everything")
    empty = ptp_utils.EmptyControl()
    empty.num_att_layers = 1
    empty(torch.zeros(1, 4, 4), True, "mid")
    assert empty.cur_step == 1 and not empty.wants_probs(True, 256)


def test_register_attention_control_names_and_places():
    unet = UNet2DConditionModel(UNetConfig.tiny(32, 48))
    model = SimpleNamespace(unet=unet)
    store = ptp_utils.AttentionStore()
    ptp_utils.register_attention_control(model, store)
    assert store.num_att_layers == 32 == len(unet.attn_processors)
    places = {n: p.place_in_unet for n, p in unet.attn_processors.items()}
    assert places["down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor"] == "down"
    assert places["mid_block.attentions.0.transformer_blocks.0.attn2.processor"] == "mid"
    assert places["up_blocks.3.attentions.2.transformer_blocks.0.attn2.processor"] == "up"
    with pytest.raises(ValueError):
        unet.set_attn_processor({"only.one.processor": None})


def test_unet_inventory_and_ddim_schedule():
    with torch.device("meta"):
        unet = UNet2DConditionModel(UNetConfig.sd15())
    assert sum(p.numel() for p in unet.parameters()) == 859_520_964  # the SD-1.x UNet
    names = dict(unet.named_parameters())
    for key in ("conv_in.weight", "time_embedding.linear_1.weight", "down_blocks.0.resnets.0.time_emb_proj.bias",
                "down_blocks.1.attentions.1.transformer_blocks.0.attn2.to_k.weight", "mid_block.attentions.0.proj_in.weight",
                "up_blocks.1.attentions.2.transformer_blocks.0.ff.net.0.proj.weight", "up_blocks.2.upsamplers.0.conv.bias",
                "down_blocks.2.downsamplers.0.conv.weight", "conv_norm_out.weight", "conv_out.bias"):
        assert key in names, key
    assert names["down_blocks.1.attentions.1.transformer_blocks.0.attn2.to_k.weight"].shape == (640, 768)
    assert names["up_blocks.1.resnets.2.conv1.weight"].shape == (1280, 1920, 3, 3)
    s = DDIMScheduler()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(981, 0, -20))  # reference utils/shared_state.py:8
    a_t, a_prev = s.alphas_for(1)
    assert a_prev == float(s.alphas_cumprod[0]) and 0 < s.alphas_for(981)[0] < 0.01
    x, eps = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    out = s.step(eps, 501, x)
    a, ap = s.alphas_for(501)
    x0 = (x - (1 - a) ** .5 * eps) / a ** .5
    torch.testing.assert_close(out.prev_sample, ap ** .5 * x0 + (1 - ap) ** .5 * eps)
    with pytest.raises(NotImplementedError):
        s.step(eps, 501, x, eta=0.5)


def test_pipeline_surface_and_loud_failures():
    assert GuidedAttentionPipeline is GuidedAttention
    import inspect
    params = inspect.signature(GuidedAttention.__call__).parameters
    for kw in ("prompt", "attention_store", "attention_res", "height", "width", "num_inference_steps", "guidance_scale",
               "negative_prompt", "num_images_per_prompt", "eta", "generator", "latents", "prompt_embeds",
               "negative_prompt_embeds", "output_type", "return_dict", "callback", "callback_steps", "cross_attention_kwargs",
               "max_iter_to_alter", "run_standard_sd", "thresholds", "scale_factor", "scale_range", "smooth_attentions",
               "sigma", "kernel_size", "sd_2_1"):
        assert kw in params, kw
    assert params["thresholds"].default == {0: 0.05, 10: 0.5, 20: 0.8} and params["max_iter_to_alter"].default == 25
    with pytest.raises(FileNotFoundError):
        GuidedAttention.from_pretrained("CompVis/stable-diffusion-v1-4")  # no network, no silent random weights
    pipe = GuidedAttention.from_pretrained("x", random_init=True, unet_config=UNetConfig.tiny(32, 48))
    with pytest.raises(ValueError):
        pipe.check_inputs("p", 100, 512, 1)
    with pytest.raises(ValueError):
        pipe.check_inputs(None, 512, 512, 1)
    lat = pipe.prepare_latents(1, 4, 256, 256, torch.float32, "cpu", torch.Generator().manual_seed(3))
    assert lat.shape == (1, 4, 32, 32)
    with pytest.raises(ValueError):
        pipe.prepare_latents(1, 4, 256, 256, torch.float32, "cpu", None, torch.zeros(1, 4, 8, 8))
    # the default attention processor is the HIP one: a CPU forward must raise, not fall back
    from guided_attention_amd._lib import GaError
    state.curHyperParams = dict(state.hyperParameterOverrides)
    with pytest.raises(GaError):
        pipe.unet(torch.zeros(1, 4, 32, 32), 981, encoder_hidden_states=torch.zeros(1, 77, 48))


def test_time_projection_buffer_equals_the_in_forward_computation():
    """The cached timestep-only buffer (sinusoid -> time MLP -> SiLU -> per-block projections, block-major) must be
    what the UNet computes inside its forward, for any batch, and must follow weight changes."""
    import torch
    from guided_attention_amd.unet import ResnetBlock2D, UNet2DConditionModel, UNetConfig
    unet = UNet2DConditionModel(UNetConfig.tiny(16, 32)).init_weights_(seed=3).float()
    for p in unet.parameters():
        p.requires_grad_(False)
    from oracle.attention import OracleStore
    from oracle.pipeline import install_processors
    install_processors(unet, OracleStore())  # the product's attention refuses the CPU; the checker's processors run here
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 4, 16, 16)).astype(np.float32))
    ctx = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 77, 32)).astype(np.float32))
    ref = unet(x, 481, encoder_hidden_states=ctx).sample                       # CPU: computes the projection in place
    flat = unet.time_projection(481, 2)
    blocks = [m for m in unet.modules() if isinstance(m, ResnetBlock2D)]
    assert flat.numel() == 2 * sum(b.time_emb_proj.out_features for b in blocks)
    out = unet(x, torch.tensor(0), encoder_hidden_states=ctx, time_projection=flat).sample  # timestep is not read
    assert (out - ref).abs().max() < 1e-4 * ref.abs().max()   # a 1-row GEMM vs a 2-row GEMM: fp32 rounding only
    assert unet.time_projection(481, 2) is flat                                  # cached per (timestep, batch)
    assert unet.time_projection(481, 1).numel() * 2 == flat.numel()
    with pytest.raises(ValueError):
        unet(x, 481, encoder_hidden_states=ctx, time_projection=unet.time_projection(481, 1))
    with torch.no_grad():  # a weight changes -> new buffer (a block with > 1 channel per group: with one channel per
        # group the norm that follows removes any per-channel constant, and the time term with it)
        next(b for b in blocks if b.time_emb_proj.out_features >= 128).time_emb_proj.weight.mul_(-1.5)
    flat2 = unet.time_projection(481, 2)
    assert flat2 is not flat and not torch.equal(flat2, flat)
    ref2 = unet(x, 481, encoder_hidden_states=ctx).sample
    out2 = unet(x, 481, encoder_hidden_states=ctx, time_projection=flat2).sample
    assert (out2 - ref2).abs().max() < 1e-4 * ref2.abs().max() and (ref2 - ref).abs().max() > 1e-3 * ref.abs().max()


# ---- loss-plan cache key: content, not identity (round-1 advisor finding)
def _plan_pipe(meta_prompt="a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]"):
    pipe = GuidedAttention(UNet2DConditionModel(UNetConfig.tiny(32, 48)), None, None, None, WordTokenizer())
    cfg = RunConfig(meta_prompt=meta_prompt, output_path="/tmp/ga_test_out")
    cfg.stable = pipe
    state.curHyperParams = dict(state.hyperParameterOverrides)
    run.parseMetaPrompt(cfg)
    return pipe, cfg


def test_loss_plan_is_keyed_on_content_not_identity():
    pipe, cfg = _plan_pipe()
    p1 = pipe._loss_plan(True, 0.5, 3)
    run.parseMetaPrompt(cfg)                     # run.execute builds a FRESH but equal token_dict per (seed, state)
    assert pipe._loss_plan(True, 0.5, 3) is p1   # -> same plan, hence the same hipGraphs
    state.curHyperParams["recurse_steps"] = 1    # a hyper-parameter the plan does not read
    assert pipe._loss_plan(True, 0.5, 3) is p1
    cfg.token_dict[2]["loss"].x = 0.1            # in-place edit of a Rect: the plan bakes the geometry in by value
    p2 = pipe._loss_plan(True, 0.5, 3)
    assert p2 is not p1 and p2.tokens[0].geom[0] == 0.1
    state.curHyperParams["shrink_factor"] = 0.0  # a hyper-parameter the plan reads
    p3 = pipe._loss_plan(True, 0.5, 3)
    assert p3 is not p2 and p3.params.shrink == 0.0
    assert pipe._loss_plan(False, 0.5, 3) is not p3


# ---- text K/V cache: entries a captured hipGraph reads are pinned against eviction
def test_kv_cache_pins_survive_eviction():
    attn = torch.nn.Module()
    attn.to_k, attn.to_v = torch.nn.Linear(6, 4, bias=False), torch.nn.Linear(6, 4, bias=False)
    unet = torch.nn.Module()
    unet.a = attn
    static = torch.randn(2, 5, 6)                        # the runner's static prompt buffer
    k0, v0 = ptp_utils.cached_context_projections(attn, static)
    k1, _ = ptp_utils.cached_context_projections(attn, static[1:2])
    assert ptp_utils.pin_context_projections(unet, {static.untyped_storage().data_ptr()}, +1) == 2
    others = [torch.randn(1, 5, 6) for _ in range(3 * ptp_utils.KV_CACHE_ENTRIES)]
    for ctx in others:                                   # eager images with other prompts
        ptp_utils.cached_context_projections(attn, ctx)
    cache = attn.__dict__["_kv_cache"]
    assert len(cache) == 2 + ptp_utils.KV_CACHE_ENTRIES  # the pinned pair + a bounded FIFO of the rest
    k0b, v0b = ptp_utils.cached_context_projections(attn, static)
    assert k0b is k0 and v0b is v0                       # same tensors: the pointers the graph captured stay valid
    with torch.no_grad():
        static.copy_(torch.randn(2, 5, 6))               # new prompt written into the static buffer
    ptp_utils.refresh_context_projections(unet)
    assert torch.allclose(k0, attn.to_k(static)) and torch.allclose(k1, attn.to_k(static[1:2]))  # refreshed in place
    ptp_utils.pin_context_projections(unet, {static.untyped_storage().data_ptr()}, -1)
    for ctx in others:
        ptp_utils.cached_context_projections(attn, ctx + 1)
    assert len(attn.__dict__["_kv_cache"]) == ptp_utils.KV_CACHE_ENTRIES


def test_transposed_weight_cache_outlives_the_view_it_was_asked_with():
    """fused_linear._transposed: the cached W^T of a 1x1 convolution's weight must survive the (out, in) VIEW it was
    requested with (a fresh view per call) and die with the parameter: tied to the view, every guidance backward
    transposed the 27 projection / shortcut weights again (round 3)."""
    import gc
    import torch
    from guided_attention_amd import fused_linear
    conv = torch.nn.Conv2d(8, 16, 1)
    t1 = fused_linear._transposed(fused_linear.conv1x1_weight(conv))
    gc.collect()
    t2 = fused_linear._transposed(fused_linear.conv1x1_weight(conv))
    assert t1 is t2 and torch.equal(t1, conv.weight.reshape(16, 8).t())
    with torch.no_grad():
        conv.weight.add_(1.0)                       # a new version: re-built
    t3 = fused_linear._transposed(fused_linear.conv1x1_weight(conv))
    assert t3 is not t1 and torch.equal(t3, conv.weight.reshape(16, 8).t())
    lin = torch.nn.Linear(8, 16)
    assert fused_linear._transposed(lin.weight) is fused_linear._transposed(lin.weight)


def test_no_gc_scope_restores_the_collector():
    """ops.no_gc (every hipGraph capture of the package runs inside it): automatic collection off inside, back to what it was
    after, also when the body raises."""
    import gc
    from guided_attention_amd import ops
    assert gc.isenabled()
    with ops.no_gc():
        assert not gc.isenabled()
    assert gc.isenabled()
    try:
        with ops.no_gc():
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert gc.isenabled()
    gc.disable()
    try:
        with ops.no_gc():
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()


def test_folded_layernorm_weights_are_evicted_only_when_dead_or_replaced_and_runners_keep_theirs():
    """fused_linear._folded (round-4 ADVICE): the per-norm cache used to `clear()` once it held more than 8 entries, freeing
    folded weights that captured hipGraphs read by raw pointer.  Now (a) only entries whose weight is gone, or that a newer
    version of the same weight replaced, are dropped, and (b) whatever a capture was handed stays alive through
    `ops.keepalive_scope`, whatever the cache does afterwards."""
    import gc
    from guided_attention_amd import fused_linear, ops
    norm = torch.nn.LayerNorm(8)
    lins = [torch.nn.Linear(8, 16) for _ in range(12)]
    with ops.keepalive_scope() as keep:                 # what GraphRunner._capture wraps its captures in
        first = fused_linear._folded(lins[0].weight, lins[0].bias, norm)
    assert all(any(t is k for k in keep.tensors) for t in first)
    for lin in lins[1:]:                                # more than 8 LIVE weights: nothing may be dropped
        fused_linear._folded(lin.weight, lin.bias, norm)
    cache = norm.__dict__["_ga_folded"]
    assert len(cache) == 12 and fused_linear._folded(lins[0].weight, lins[0].bias, norm)[0] is first[0]
    wg = first[0]
    ref = (lins[0].weight.float() * norm.weight.float()[None, :])
    assert torch.allclose(wg.float(), ref) and torch.allclose(first[1], ref.sum(1))
    # dead weights go at the next insertion
    del lins[4:], lin                                  # (the loop variable held the last one)
    gc.collect()
    extra = torch.nn.Linear(8, 16)
    fused_linear._folded(extra.weight, extra.bias, norm)
    assert len(cache) == 5
    # a new version of a live weight replaces its older entry at the next eviction; the capture's copy is untouched
    with torch.no_grad():
        lins[0].weight.mul_(2.0)
    second = fused_linear._folded(lins[0].weight, lins[0].bias, norm)
    assert second[0] is not first[0] and torch.allclose(second[0].float(), 2 * ref)
    more = [torch.nn.Linear(8, 16) for _ in range(6)]
    for lin in more:
        fused_linear._folded(lin.weight, lin.bias, norm)
    assert sum(1 for k in cache if k[0] == lins[0].weight.data_ptr()) == 1
    assert torch.allclose(keep.tensors[0].float(), ref)     # the runner's reference still holds the tensor it captured against
    assert ops._KEEP is None


def test_from_pretrained_loads_a_local_diffusers_folder(tmp_path):
    """`GuidedAttention.from_pretrained(<local diffusers folder>)` (reference: run.py:18-29 loads
    `CompVis/stable-diffusion-v1-4` that way; there is no network here, so the folder is written by the test): UNet weights by
    their diffusers names from `unet/diffusion_pytorch_model.safetensors`, the VAE's decoder-side tensors from
    `vae/diffusion_pytorch_model.safetensors` (encoder tensors ignored), `revision="fp16"` -> half precision, a mismatching
    checkpoint refused, a missing folder refused unless random_init."""
    from safetensors.torch import save_file
    from guided_attention_amd.pipeline_guided_attention import GuidedAttention
    from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
    cfg = UNetConfig.tiny(32, 48)
    src = UNet2DConditionModel(cfg).init_weights_(seed=77)
    sd = {k: v.detach().clone().contiguous() for k, v in src.state_dict().items()}
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight" in sd and "conv_in.weight" in sd   # diffusers names
    folder = tmp_path / "stable-diffusion-tiny"
    (folder / "unet").mkdir(parents=True)
    (folder / "vae").mkdir()
    save_file(sd, str(folder / "unet" / "diffusion_pytorch_model.safetensors"))
    pqc_w, pqc_b = torch.randn(4, 4, 1, 1), torch.randn(4)
    save_file({"post_quant_conv.weight": pqc_w, "post_quant_conv.bias": pqc_b, "encoder.conv_in.weight": torch.randn(8, 3, 3, 3),
               "quant_conv.weight": torch.randn(8, 8, 1, 1)}, str(folder / "vae" / "diffusion_pytorch_model.safetensors"))
    pipe = GuidedAttention.from_pretrained(str(folder), unet_config=cfg)
    got = pipe.unet.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert torch.equal(pipe.vae.post_quant_conv.weight, pqc_w) and torch.equal(pipe.vae.post_quant_conv.bias, pqc_b)
    assert pipe.unet.dtype == torch.float32 and pipe.tokenizer is not None and pipe.text_encoder is not None
    half = GuidedAttention.from_pretrained(str(folder), revision="fp16", unet_config=cfg)
    assert half.unet.dtype == torch.float16
    assert torch.equal(half.unet.conv_in.weight, sd["conv_in.weight"].half())
    bare = GuidedAttention.from_pretrained(str(folder), unet_config=cfg, weights=False)     # ranks != 0 before the broadcast
    assert set(bare.unet.state_dict()) == set(sd)
    # a checkpoint of another architecture is refused, by name
    bad = dict(sd)
    bad.pop("conv_in.weight")
    bad["conv_in.kernel"] = sd["conv_in.weight"]
    save_file(bad, str(folder / "unet" / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(RuntimeError, match="checkpoint mismatch"):
        GuidedAttention.from_pretrained(str(folder), unet_config=cfg)
    with pytest.raises(FileNotFoundError):
        GuidedAttention.from_pretrained("CompVis/stable-diffusion-v1-4")
    GuidedAttention.from_pretrained("CompVis/stable-diffusion-v1-4", random_init=True, unet_config=cfg)
