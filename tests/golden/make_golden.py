#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own functions.

Runs ONLY in the build container (it needs /root/reference, which never travels to the
GPU box).  It imports the reference's hot-path modules, drives them on explicit seeded
inputs and writes inputs + expected outputs as small .npz / .json DATA files.  No
reference source text is copied: only numbers and strings produced by running it.

How the reference is made importable here:
  * utils/gaussian_smoothing.py, utils/helpers.py, utils/shared_state.py, config.py import
    as-is.
  * utils/ptp_utils.py and pipeline_guided_attention.py import third-party packages that
    are not installed (cv2, IPython, diffusers 0.12.1, pyrallis) purely for names that the
    hot-path functions never call; empty placeholder modules satisfy those import
    statements.  None of the arithmetic that is pinned below goes through a placeholder.
  * the reference pins tensors with `.cuda()`; on this CPU-only box that call is made the
    identity for the duration of this script.

Usage:  python tests/golden/make_golden.py        (rewrites tests/golden/*.npz, *.json)
"""
import json
import logging
import math
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT))
import hashrand  # noqa: E402  (inputs rebuilt bit-for-bit by the tests instead of being stored)


# ----------------------------------------------------------------------------- import recipe
class _Placeholder(types.ModuleType):
    """Module whose every attribute is an empty class (satisfies `from x import Y`)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {})
        setattr(self, name, cls)
        return cls


def import_reference():
    if not REF.exists():
        raise SystemExit("reference checkout not present; fixtures can only be regenerated in the build container")
    sys.path.insert(0, str(REF))
    names = [
        "cv2", "IPython", "IPython.display", "pyrallis",
        "diffusers", "diffusers.configuration_utils", "diffusers.models",
        "diffusers.models.unet_2d_condition", "diffusers.models.cross_attention",
        "diffusers.schedulers", "diffusers.utils", "diffusers.pipelines",
        "diffusers.pipelines.pipeline_utils", "diffusers.pipelines.stable_diffusion",
        "diffusers.pipelines.stable_diffusion.safety_checker",
    ]
    for n in names:
        sys.modules[n] = _Placeholder(n)
        if "." in n:  # `from pkg import sub` must resolve to the sub-module placeholder
            parent, leaf = n.rsplit(".", 1)
            setattr(sys.modules[parent], leaf, sys.modules[n])
    sys.modules["diffusers.utils"].logging = types.SimpleNamespace(get_logger=logging.getLogger)
    sys.modules["pyrallis"].wrap = lambda *a, **k: (lambda f: f)
    import transformers
    if not hasattr(transformers, "CLIPFeatureExtractor"):
        transformers.CLIPFeatureExtractor = type("CLIPFeatureExtractor", (), {})
    # neutralise device pinning (CPU-only box)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    cwd = os.getcwd()
    os.chdir("/tmp")  # RunConfig.__post_init__ creates ./outputs
    try:
        import config as ref_config
        import utils.shared_state as state
        import utils.helpers as helpers
        import utils.gaussian_smoothing as gs
        import utils.ptp_utils as ptp
        import pipeline_guided_attention as pga
    finally:
        os.chdir(cwd)
    return ref_config, state, helpers, gs, ptp, pga


ref_config, state, helpers, gs, ptp, pga = import_reference()

BASE_PROMPT = "a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]"


class WordTokenizer:
    """Whitespace tokenizer with CLIP's framing (BOS=49406, EOS=49407); for the prompts
    used here every word is one CLIP token, so indices match the real tokenizer."""

    def __init__(self):
        self.vocab = {}
        self.model_max_length = 77

    def _id(self, w):
        return self.vocab.setdefault(w, 1000 + len(self.vocab))

    def __call__(self, text, **kw):
        return {"input_ids": [49406] + [self._id(w) for w in text.split()] + [49407]}

    def decode(self, tid):
        for w, i in self.vocab.items():
            if i == tid:
                return w
        return {49406: "<|startoftext|>", 49407: "<|endoftext|>"}.get(tid, "?")


class Harness(pga.GuidedAttention):
    def __init__(self):
        self.tokenizer = WordTokenizer()
        self.prompt = None

    def save_viridis(self, tensor1, tag):  # PNG side effect of the reference: dropped
        pass


def fresh_config(meta_prompt, **over):
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        cfg = ref_config.RunConfig(meta_prompt=meta_prompt)
    finally:
        os.chdir(cwd)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def setup_prompt(h, meta_prompt, hyper=None, **cfg_over):
    """Mirror of run.py:parseMetaPrompt driven with the word tokenizer."""
    cfg = fresh_config(meta_prompt, **cfg_over)
    state.config = cfg
    state.curHyperParams = dict(state.hyperParameterOverrides)
    if hyper:
        state.curHyperParams.update(hyper)
    state.cur_time_step_iter = 0
    state.sub_iteration = 0
    cfg.prompt, cfg.meta_info, cfg.custom_loss = helpers.parse_prompt(cfg.meta_prompt)
    tokenized = h.tokenizer(cfg.prompt)["input_ids"]
    token_dict = {}
    for item in cfg.meta_info:
        toks = h.tokenizer(item[0])["input_ids"][1:-1]
        n = len(toks)
        idx = None
        for i in range(0, len(tokenized) - n):
            if tokenized[i:i + n] == toks:
                idx = list(range(i, i + n))
                break
        for i in idx:
            token_dict[i] = {"word": h.tokenizer.decode(tokenized[i]), "loss_type": item[1],
                             "loss": item[2], "subprompt": item[0]}
    cfg.token_dict = token_dict
    if not cfg.custom_loss:
        del cfg.custom_loss  # the reference's hasattr() probe must see "no custom loss"
    h.prompt = cfg.prompt
    return cfg


def ser_meta(meta_info):
    out = []
    for tok, typ, val in meta_info:
        if typ == helpers.AnnotationType.BOX:
            v = [val.x, val.y, val.width, val.height, val.size]
        elif typ == helpers.AnnotationType.COOR:
            v = list(val)
        else:
            v = None
        out.append([tok, typ.name, v])
    return out


def f32(t):
    if isinstance(t, (int, float)):
        return np.float32(t)
    return t.detach().to(torch.float32).cpu().numpy()


# ----------------------------------------------------------------------------- G1
def g1_gaussian():
    out = {}
    for k, s in [(3, 0.5), (3, 1.0), (5, 1.0), (5, 0.75)]:
        m = gs.GaussianSmoothing(channels=1, kernel_size=k, sigma=s, dim=2)
        out[f"k{k}_s{s}"] = f32(m.weight[0, 0])
    # forward on a reflect-padded random image (the call pattern of pipeline:252-254)
    g = torch.Generator().manual_seed(11)
    img = torch.rand(16, 16, generator=g)
    m = gs.GaussianSmoothing(channels=1, kernel_size=3, sigma=0.5, dim=2)
    inp = torch.nn.functional.pad(img[None, None], (1, 1, 1, 1), mode="reflect")
    out["fwd_in"] = f32(img)
    out["fwd_out"] = f32(m(inp)[0, 0])
    np.savez(OUT / "g1_gaussian.npz", **out)


# ----------------------------------------------------------------------------- G2
class _DummyCustom:
    def subprompts_of_interest(self, args):
        return [a.strip() for a in args.strip("()").split(",")]


def g2_parse():
    prompts = [
        BASE_PROMPT,
        "a [rat:.2,.4] and a [fox:.6,.5]",
        "a [robot:.6,.3,.4,.55] and a [vase:0,.3,.4,.55] and the [moon:.35,.05,.35,.35]",
        "a photo of a [cat:0.1,0.2,0.3,0.4] sitting on grass",
        "[dog: .5 , .5] running",
        "plain prompt with no annotations",
        "a [cat:.2,.5] and a [vase:.7,.5] [CustomLoss:toLeftOf (cat, vase)]",
        "  leading spaces and a [red ball:.1,.1,.2,.2]",
    ]
    cases = []
    for p in prompts:
        cfg = fresh_config(p)
        cfg.registered_loss_functions = {"toLeftOf": _DummyCustom()}
        state.config = cfg
        prompt, meta, custom = helpers.parse_prompt(p)
        cases.append({"meta_prompt": p, "prompt": prompt, "meta_info": ser_meta(meta),
                      "custom_losses": {k: v[1] for k, v in custom.items()}})
    (OUT / "g2_parse_prompt.json").write_text(json.dumps(cases, indent=1))


# ----------------------------------------------------------------------------- G3
def g3_masks():
    rects = [(.6, .3, .4, .55), (.2, .3, .4, .55), (0., 0., .5, .5), (.25, .25, .5, .5),
             (.35, .05, .35, .35), (0., .3, .4, .55), (.7, .7, .5, .5), (.4, .4, .01, .01)]
    cases = []
    arrs = {}
    n = 0
    for res in (16, 24, 32):
        for shrink in (0.0, 0.15, 0.0625):
            state.curHyperParams = dict(state.hyperParameterOverrides, shrink_factor=shrink)
            for r in rects:
                rect = helpers.Rect(r[0], r[1], r[2], r[3], 1).of_size(float(res))
                mask = np.zeros((res, res), np.uint8)
                for ii in range(res):
                    for jj in range(res):
                        mask[ii, jj] = 1 if helpers.inside_box(jj, ii, rect) else 0
                arrs[f"mask{n}"] = mask
                cases.append({"id": n, "res": res, "shrink": shrink, "rect": list(r),
                              "scaled": [rect.x, rect.y, rect.width, rect.height],
                              "count": int(mask.sum()), "center": list(helpers.Rect(*r, 1).center())})
                n += 1
    np.savez_compressed(OUT / "g3_inside_box.npz", **arrs)
    (OUT / "g3_inside_box.json").write_text(json.dumps(cases))


# ----------------------------------------------------------------------------- G4
def make_maps(kind, seed, res=16, ntok=77):
    g = torch.Generator().manual_seed(seed)
    if kind == "flat":
        A = torch.softmax(torch.randn(res, res, ntok, generator=g), -1)
    elif kind == "sharp":
        A = torch.softmax(3.0 * torch.randn(res, res, ntok, generator=g), -1)
    elif kind == "onehot":
        logits = torch.randn(res, res, ntok, generator=g)
        hot = torch.randint(0, ntok, (res, res), generator=g)
        logits.scatter_(-1, hot[..., None], 14.0)
        A = torch.softmax(logits, -1)
    elif kind == "bos":  # realistic: BOS takes most of the mass, text tokens ~1e-2, pads ~1e-3
        logits = torch.randn(res, res, ntok, generator=g)
        logits[..., 0] += 6.0
        logits[..., 1:8] += 2.0
        # a blob of attention for token 2 and token 6 so the centroid terms are not degenerate
        yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
        logits[..., 2] += 2.5 * torch.exp(-((yy - 4.0) ** 2 + (xx - 11.0) ** 2) / 8.0)
        logits[..., 6] += 2.5 * torch.exp(-((yy - 10.0) ** 2 + (xx - 3.0) ** 2) / 8.0)
        A = torch.softmax(logits, -1)
    else:
        raise ValueError(kind)
    return A.to(torch.float32)


def run_loss_case(h, A0, smooth, sigma, ksize, normalize_eot):
    A_leaf = A0.clone().requires_grad_(True)
    maps = A_leaf * 1.0  # the reference multiplies a view of the maps in place: needs a non-leaf
    d = h._compute_max_attention_per_index(maps, smooth_attentions=smooth, sigma=sigma,
                                           kernel_size=ksize, normalize_eot=normalize_eot)
    loss, losses, unscaled = pga.GuidedAttention._compute_loss(d)
    total, per_sub = pga.GuidedAttention.group_losses_by_sumprompt(losses)
    _, per_sub_unscaled = pga.GuidedAttention.group_losses_by_sumprompt(unscaled)
    grad = torch.autograd.grad(loss.requires_grad_(True), [A_leaf], allow_unused=True)[0]
    if grad is None:
        grad = torch.zeros_like(A_leaf)
    T = len(d["max_loss"])

    def lst(x):
        return np.array([float(v) for v in x], np.float32).reshape(T)

    return {
        "A": f32(A0), "max": lst(d["max_loss"]), "col": lst(d["col"]), "row": lst(d["row"]),
        "inside": lst(d["inside_loss"]), "outside": lst(d["outside_loss"]),
        "loss": np.float32(float(loss)), "tok": np.array([k for k, _ in losses], np.int32),
        "losses": np.array([float(v) for _, v in losses], np.float32),
        "unscaled": np.array([float(v) for _, v in unscaled], np.float32),
        "sub_unscaled": np.array([float(v) for v in per_sub_unscaled.values()], np.float32),
        "dA": f32(grad),
    }


def g4_loss():
    h = Harness()
    specs = [
        # name, meta prompt, map kind, seed, smooth, avg_within, sd21(normalize_eot), hyper overrides
        ("base_flat", BASE_PROMPT, "flat", 1, True, False, False, None),
        ("base_sharp", BASE_PROMPT, "sharp", 2, True, False, False, None),
        ("base_onehot", BASE_PROMPT, "onehot", 3, True, False, False, None),
        ("base_bos", BASE_PROMPT, "bos", 4, True, False, False, None),
        ("base_bos_nosmooth", BASE_PROMPT, "bos", 4, False, False, False, None),
        ("base_bos_avg", BASE_PROMPT, "bos", 5, True, True, False, None),
        ("base_bos_eot", BASE_PROMPT, "bos", 6, True, False, True, None),
        ("coor_bos", "a [rat:.2,.4] and a [fox:.6,.5]", "bos", 7, True, False, False, None),
        ("mixed_sharp", "a [robot:.6,.3,.4,.55] and a [vase:.2,.45] on the [moon:.35,.05,.35,.35]", "sharp", 8,
         True, False, False, None),
        ("hyper_bos", BASE_PROMPT, "bos", 9, True, False, False,
         {"inside_loss_scale": .5, "outside_loss_scale": .1, "shrink_factor": 0.0, "bb_center_weight": 0.0}),
        ("hyper2_flat", BASE_PROMPT, "flat", 10, True, True, False,
         {"inside_loss_scale": 1.0, "outside_loss_scale": .3, "shrink_factor": 0.05, "bb_center_weight": 0.2}),
        # kernel_size 5 is not runnable in the reference (pad is hard-coded to 1: IndexError at
        # pipeline_guided_attention.py:267), so only sigma varies here
        ("s1_bos", BASE_PROMPT, "bos", 12, True, False, False, {"_sigma": 1.0, "_ksize": 3}),
        # strict bounding-box mode (helpers.py:216-264: weight table + hinge terms)
        ("strict_bos", BASE_PROMPT, "bos", 13, True, False, False, {"strict": True}),
        ("strict_sharp_noshrink", BASE_PROMPT, "sharp", 14, True, False, False, {"strict": True, "shrink_factor": 0.0}),
        ("strict_mixed_flat", "a [robot:.6,.3,.4,.55] and a [vase:.2,.45] on the [moon:.35,.05,.35,.35]", "flat", 15,
         False, True, False, {"strict": True, "inside_loss_scale": .7, "outside_loss_scale": .4}),
    ]
    meta = []
    arrs = {}
    for name, mp, kind, seed, smooth, avg, eot, hyper in specs:
        hyper = dict(hyper or {})
        sigma = hyper.pop("_sigma", 0.5)
        ksize = hyper.pop("_ksize", 3)
        cfg = setup_prompt(h, mp, hyper or None, sub_prompt_avg_within=avg)
        A0 = make_maps(kind, seed)
        r = run_loss_case(h, A0, smooth, sigma, ksize, eot)
        for k, v in r.items():
            arrs[f"{name}.{k}"] = v
        td = {}
        for k, v in cfg.token_dict.items():
            val = v["loss"]
            if v["loss_type"] == helpers.AnnotationType.BOX:
                val = [val.x, val.y, val.width, val.height]
            else:
                val = list(val)
            td[str(k)] = {"word": v["word"], "loss_type": v["loss_type"].name, "loss": val,
                          "subprompt": v["subprompt"]}
        meta.append({"name": name, "meta_prompt": mp, "prompt": cfg.prompt, "smooth": smooth, "sigma": sigma,
                     "kernel_size": ksize, "sub_prompt_avg_within": avg, "normalize_eot": eot,
                     "n_prompt_tokens": len(h.tokenizer(cfg.prompt)["input_ids"]),
                     "hyper": dict(state.curHyperParams), "token_dict": td})
    np.savez_compressed(OUT / "g4_loss.npz", **arrs)
    (OUT / "g4_loss.json").write_text(json.dumps(meta, indent=1, default=str))


# ----------------------------------------------------------------------------- G5
def g5_threshold():
    h = Harness()
    setup_prompt(h, BASE_PROMPT)
    rows = []
    loss_sets = {
        "low": [(2, torch.tensor([0.05])), (5, torch.tensor([0.02])), (6, torch.tensor([0.03]))],
        "mid": [(2, torch.tensor([0.30])), (5, torch.tensor([0.20])), (6, torch.tensor([0.15]))],
        "high": [(2, torch.tensor([0.90])), (5, torch.tensor([0.70])), (6, torch.tensor([0.60]))],
        "edge": [(2, torch.tensor([0.25])), (5, torch.tensor([0.125])), (6, torch.tensor([0.125]))],
    }
    thr_sets = {"default": {0: 1.0}, "two": {0: 0.1, 3: 0.8}, "three": {0: 0.05, 10: 0.5, 20: 0.8},
                "quarter": {0: 0.25}, "empty": {}}
    for ln, losses in loss_sets.items():
        for tn, thr in thr_sets.items():
            for i in (-1, 0, 1, 3, 10, 20):
                try:
                    res = bool(h.meets_threshold(i, thr, losses))
                except Exception as e:  # e.g. i == -1 with an empty dict never indexes: record anyway
                    res = f"raises:{type(e).__name__}"
                rows.append({"losses": ln, "thresholds": tn, "i": i, "result": res})
    doc = {"loss_sets": {k: [[t, float(v)] for t, v in ls] for k, ls in loss_sets.items()},
           "thr_sets": {k: {str(a): b for a, b in v.items()} for k, v in thr_sets.items()},
           "token_subprompt": {"2": "robot", "5": "blue vase", "6": "blue vase"}, "rows": rows}
    (OUT / "g5_meets_threshold.json").write_text(json.dumps(doc))


# ----------------------------------------------------------------------------- G6
class DuckAttention(torch.nn.Module):
    """Duck-typed stand-in for the attention module the processor is handed (only its
    linear layers and the head split/merge; the softmax(QK^T)V arithmetic under test is the
    reference processor's own)."""

    def __init__(self, C, heads, ctx_dim, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.heads = heads
        self.scale = (C // heads) ** -0.5
        self.upcast_attention = False
        self.upcast_softmax = False
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(ctx_dim, C, bias=False)
        self.to_v = torch.nn.Linear(ctx_dim, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / math.sqrt(p.shape[-1])))
        self.captured = {}

        def keep_q(mod, inp, outp):  # returns None: the output is not replaced
            outp.retain_grad()
            self.captured["q"] = outp

        self.to_q.register_forward_hook(keep_q)

    def prepare_attention_mask(self, mask, n):
        return mask

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        return t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, n, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, n, d).permute(0, 2, 1, 3).reshape(b, n, d * self.heads)


def g6_processor():
    h = Harness()
    setup_prompt(h, BASE_PROMPT)
    specs = [
        # name, C, heads, N, ctx_len (None = self attention), ctx_dim, batch, place
        ("cross_d16", 32, 2, 256, 77, 48, 1, "up"),
        ("cross_d40", 80, 2, 256, 77, 48, 1, "down"),
        ("cross_d40_b2", 80, 2, 64, 77, 48, 2, "mid"),
        ("cross_big", 16, 2, 1056, 77, 24, 1, "down"),  # N > 32^2: not stored
        ("self_d16", 32, 2, 64, None, 32, 1, "mid"),
        ("self_d40", 80, 2, 256, None, 80, 1, "up"),
        # paint-with-words (ptp_utils.py:113-138): additive box mask scaled by the global score maximum and log(1+sigma_t)
        ("cross_pww_d16", 32, 2, 256, 77, 48, 1, "up"),
        ("cross_pww_d40_b2", 80, 2, 64, 77, 48, 2, "mid"),
    ]
    arrs = {}
    meta = []
    sys.path.insert(0, str(OUT.parent.parent))
    from oracle.pipeline import alphas_cumprod, ddim_timesteps
    acp = alphas_cumprod().numpy().astype(np.float64)
    for si, (name, C, heads, N, ctx_len, ctx_dim, B, place) in enumerate(specs):
        seed = 600 + 10 * si
        pww = None
        state.curHyperParams = dict(state.hyperParameterOverrides)
        if "pww" in name:
            pww = {"stop": 3, "weight": 0.7, "iter": 1}
            state.curHyperParams.update(paint_with_words_stop=pww["stop"], paint_with_words_weight=pww["weight"])
            state.cur_time_step_iter = pww["iter"]
            state.timesteps = ddim_timesteps(50)
            state.sigmas = ((1 - acp) / acp) ** 0.5
            pww["log1p_sigma"] = float(np.log(1 + state.get_sigma()))
        attn = DuckAttention(C, heads, ctx_dim, seed=0)
        with torch.no_grad():  # weights from the integer-hash generator: rebuilt by the tests
            for pi, (pn, p) in enumerate(attn.named_parameters()):
                w = hashrand.normalish(tuple(p.shape), seed + 1 + pi) * np.float32(1.5 / math.sqrt(p.shape[-1]))
                p.copy_(torch.from_numpy(w))
        store = ptp.AttentionStore()
        store.num_att_layers = 1
        proc = ptp.AttendExciteCrossAttnProcessor(attnstore=store, place_in_unet=place)
        x = torch.from_numpy(hashrand.normalish((B, N, C), seed)).requires_grad_(True)
        ctx = None if ctx_len is None else torch.from_numpy(hashrand.normalish((B, ctx_len, ctx_dim), seed + 7))
        out = proc(attn, x, encoder_hidden_states=ctx)
        key = f"{place}_{'cross' if ctx_len is not None else 'self'}"
        stored = store.attention_store[key]
        R1 = torch.from_numpy(hashrand.normalish(tuple(out.shape), seed + 8))
        scal = (out * R1).sum()
        if stored:
            P = stored[0]
            R2 = torch.from_numpy(hashrand.normalish(tuple(P.shape), seed + 9))
            scal = scal + (P * R2).sum()
            arrs[f"{name}.P"] = f32(P)
        scal.backward()
        arrs[f"{name}.out"] = f32(out)
        arrs[f"{name}.dx"] = f32(x.grad)
        arrs[f"{name}.dq"] = f32(attn.captured["q"].grad)
        meta.append({"name": name, "C": C, "heads": heads, "N": N, "ctx_len": ctx_len, "ctx_dim": ctx_dim, "pww": pww,
                     "batch": B, "place": place, "scale": attn.scale, "stored": bool(stored), "seed": seed,
                     "param_order": [pn for pn, _ in attn.named_parameters()],
                     "store_keys": {k: len(v) for k, v in store.attention_store.items()},
                     "cur_step": store.cur_step, "cur_att_layer": store.cur_att_layer})
    np.savez_compressed(OUT / "g6_processor.npz", **arrs)
    (OUT / "g6_processor.json").write_text(json.dumps(meta, indent=1))


# ----------------------------------------------------------------------------- G7
def g7_aggregate():
    h = Harness()
    setup_prompt(h, BASE_PROMPT)
    arrs = {}
    meta = []
    for name, B, heads in (("b1", 1, 8), ("b2", 2, 4)):
        store = ptp.AttentionStore()
        layout = [("down", True, 1024), ("down", True, 256), ("down", True, 256), ("down", False, 256),
                  ("mid", True, 64), ("up", True, 256), ("up", True, 256), ("up", True, 256), ("up", True, 1024),
                  ("up", False, 64), ("down", True, 4096)]
        store.num_att_layers = len(layout)
        for li, (place, is_cross, N) in enumerate(layout):
            K = 77 if is_cross else N
            # aggregation is linear: positive hash-uniform inputs are as good as real softmax rows
            P = torch.from_numpy(hashrand.uniform((B * heads, N, K), 700 + 20 * B + li))
            store(P, is_cross, place)
        for res, is_cross, where in ((16, True, ("up", "down", "mid")), (16, True, ("up",)), (8, True, ("up", "down", "mid")),
                                     (32, True, ("down", "up")), (16, False, ("up", "down", "mid"))):
            A = ptp.aggregate_attention(store, res, where, is_cross, 0)
            tag = f"{name}.agg_r{res}_{'c' if is_cross else 's'}_{'-'.join(where)}"
            arrs[tag] = f32(A)
        meta.append({"name": name, "batch": B, "heads": heads, "seed_base": 700 + 20 * B,
                     "layout": [[p, c, n] for p, c, n in layout],
                     "store_keys": {k: len(v) for k, v in store.attention_store.items()}})
    np.savez_compressed(OUT / "g7_aggregate.npz", **arrs)
    (OUT / "g7_aggregate.json").write_text(json.dumps(meta, indent=1))


# ----------------------------------------------------------------------------- G8
def g8_update():
    h = Harness()
    setup_prompt(h, BASE_PROMPT)
    g = torch.Generator().manual_seed(88)
    lat = torch.randn(1, 4, 8, 8, generator=g).requires_grad_(True)
    w = torch.randn(1, 4, 8, 8, generator=g)
    loss = (torch.sin(lat) * w).sum().reshape(1) * 0.01
    step = 20 * math.sqrt(0.75)
    new = pga.GuidedAttention._update_latent(lat, loss, step)
    np.savez(OUT / "g8_update_latent.npz", latents=f32(lat), w=f32(w), step=np.float64(step), out=f32(new),
             grad=f32(torch.autograd.grad(loss, [lat])[0]))

# ----------------------------------------------------------------------------- G9 loop-level trace
def hash_init_(module, seed):
    """Deterministic, platform-independent weights (tests rebuild them bit-for-bit)."""
    with torch.no_grad():
        for pi, (name, p) in enumerate(module.named_parameters()):
            if name.endswith("bias"):
                p.zero_()
            elif p.dim() == 1:
                p.fill_(1.0)
            else:
                fan_in = p[0].numel()
                u = hashrand.uniform(tuple(p.shape), seed + pi) * np.float32(2.0) - np.float32(1.0)
                p.copy_(torch.from_numpy(u * np.float32(math.sqrt(3.0 / fan_in))))
    return module


class _DiffusersFacade:
    """The build's UNet blocks behind the diffusers-0.12.1 call signatures that the REFERENCE's own UNet forward
    (pipeline_guided_attention.py:583-743) uses, so that this forward — time embedding, conv_in, the down / mid / up
    sequence with its skip-tuple slicing and `upsample_size` rule, norm / act / conv_out — is what runs, not the
    build's `UNet2DConditionModel.forward`.  The reference installs it with `unet.__dict__['forward'] = self.forward`
    (:854); calling the facade dispatches to that entry exactly as `nn.Module.__call__` would.

    What is adapted (block-internal conventions of diffusers, restated; absent from /root/reference):
      * blocks receive the raw time embedding `temb`; the build's ResnetBlock2D takes SiLU(temb) (the SiLU sits
        inside diffusers' ResnetBlock2D);
      * up blocks receive `res_hidden_states_tuple` and consume it from its END;
      * `conv_norm_out` and `conv_act` are separate callables (the build fuses the SiLU into the norm layer)."""

    class _Down:
        def __init__(self, blk):
            self.blk, self.has_cross_attention = blk, blk.has_cross_attention

        def __call__(self, hidden_states, temb, encoder_hidden_states=None, attention_mask=None,
                     cross_attention_kwargs=None):
            x, outs = self.blk(hidden_states, F.silu(temb), encoder_hidden_states)
            return x, tuple(outs)

    class _Up:
        def __init__(self, blk):
            self.blk, self.has_cross_attention, self.resnets = blk, blk.has_cross_attention, blk.resnets

        def __call__(self, hidden_states, temb, res_hidden_states_tuple, encoder_hidden_states=None,
                     cross_attention_kwargs=None, upsample_size=None, attention_mask=None):
            skips = list(res_hidden_states_tuple)
            out = self.blk(hidden_states, skips, F.silu(temb), encoder_hidden_states, upsample_size)
            assert not skips, "an up block must consume exactly len(resnets) skip tensors"
            return out

    def __init__(self, real):
        from guided_attention_amd.unet import timestep_embedding
        self.real = real
        cfg = real.config
        self.config = types.SimpleNamespace(sample_size=cfg.sample_size, cross_attention_dim=cfg.cross_attention_dim,
                                            block_out_channels=list(cfg.block_out_channels),
                                            center_input_sample=cfg.center_input_sample, class_embed_type=None)
        self.in_channels = real.in_channels
        self.num_upsamplers = real.num_upsamplers
        self.dtype = real.dtype
        self.class_embedding = None
        self.time_proj = lambda timesteps: timestep_embedding(timesteps, cfg.block_out_channels[0])
        self.time_embedding = real.time_embedding
        self.conv_in = real.conv_in
        self.down_blocks = [self._Down(b) for b in real.down_blocks]
        self.mid_block = lambda sample, emb, encoder_hidden_states=None, attention_mask=None, \
            cross_attention_kwargs=None: real.mid_block(sample, F.silu(emb), encoder_hidden_states)
        self.up_blocks = [self._Up(b) for b in real.up_blocks]
        n = real.conv_norm_out
        self.conv_norm_out = lambda x: F.group_norm(x, n.num_groups, n.weight, n.bias, n.eps)
        self.conv_act = F.silu
        self.conv_out = real.conv_out
        self.calls = []

    @property
    def attn_processors(self):
        return self.real.attn_processors

    def set_attn_processor(self, procs):
        self.real.set_attn_processor(procs)

    def zero_grad(self):
        pass

    def __call__(self, sample, t, encoder_hidden_states=None, cross_attention_kwargs=None):
        self.calls.append((int(sample.shape[0]), bool(torch.is_grad_enabled() and sample.requires_grad)))
        out = self.__dict__["forward"](sample, t, encoder_hidden_states, return_dict=False)   # installed at :854
        return types.SimpleNamespace(sample=out[0])


def g9_loop():
    sys.path.insert(0, str(OUT.parent.parent))
    from guided_attention_amd.scheduler import DDIMScheduler
    from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
    import contextlib

    pga.DDIMScheduler = DDIMScheduler  # "the build's DDIM" (diffusers is not installed)
    real_generator, real_randn = torch.Generator, torch.randn
    torch.Generator = lambda device=None: real_generator("cpu")            # reference hard-codes 'cuda' (:921)
    torch.randn = lambda *a, **k: real_randn(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})

    class LoopHarness(Harness):
        vae_scale_factor = 8
        _execution_device = torch.device("cpu")

        def __init__(self, unet, embeds):
            super().__init__()
            self.unet = unet
            self.scheduler = DDIMScheduler()
            self.embeds = embeds
            tok_outer = self.tokenizer

            class Tok:  # padded-tensor front of the word tokenizer
                model_max_length = 77

                def __call__(self_, text, padding=None, max_length=None, truncation=None, return_tensors=None):
                    if return_tensors is None:
                        return tok_outer(text)
                    texts = [text] if isinstance(text, str) else text
                    return types.SimpleNamespace(input_ids=torch.zeros(len(texts), 77, dtype=torch.long) +
                                                 (0 if texts[0] else 1))

                def decode(self_, tid):
                    return tok_outer.decode(tid)

                def batch_decode(self_, ids):
                    return []

            self.tokenizer = Tok()

            class Enc:
                config = types.SimpleNamespace()
                dtype = torch.float32

                def __call__(self_, ids, attention_mask=None):  # ids all-ones marks the empty (negative) prompt
                    return (embeds[0:1] if int(ids[0, 0]) == 1 else embeds[1:2],)

            self.text_encoder = Enc()

        def check_inputs(self, *a, **k):
            pass

        def prepare_latents(self, bs, ch, height, width, dtype, device, generator, latents=None):
            return latents

        def prepare_extra_step_kwargs(self, generator, eta):
            return {}

        def progress_bar(self, total=None):
            return contextlib.nullcontext(types.SimpleNamespace(update=lambda: None))

        def save_image(self, latent, tag):
            pass

        def decode_latents(self, latents):
            self.final_latents = latents.detach().clone()
            return np.zeros((1, 8, 8, 3), np.float32)

        def numpy_to_pil(self, image):
            return [image]

    cases = [
        # name, steps, call thresholds, only_update_on_threshold_steps, max_iter_to_alter, hyper, scale_factor
        ("default_like", 6, {0: 1.0}, True, 25, {"recurse_steps": 3, "recurse_until": 14}, 20),
        ("every_step", 6, {0: 9.0, 2: 1.7}, False, 4, {"recurse_steps": 2, "recurse_until": 1}, 20),
        ("no_recurse_thr2", 5, {0: 2.5, 1: 0.5}, True, 25, {"recurse_steps": 1}, 10),
    ]
    arrs, meta = {}, []
    for ci, (name, steps, thr, only_thr, max_alter, hyper, sf) in enumerate(cases):
        cfgu = UNetConfig.tiny(sample_size=32, cross_attention_dim=48)
        real = hash_init_(UNet2DConditionModel(cfgu), 9000 + 1000 * ci).float()
        for p in real.parameters():
            p.requires_grad_(False)
        shim = _DiffusersFacade(real)   # the loop runs the reference's OWN UNet forward (:583-743) on the build's blocks
        embeds = torch.from_numpy(hashrand.normalish((2, 77, 48), 9100 + ci))
        h = LoopHarness(shim, embeds)
        cfg = setup_prompt(h, BASE_PROMPT, hyper, only_update_on_threshold_steps=only_thr)
        cfg.thresholds = dict(thr)  # run.py:overrideConfig: config.thresholds := hyper-param thresholds
        state.curHyperParams["thresholds"] = dict(thr)
        state.cur_seed = 7
        helpers.log_clear()
        controller = ptp.AttentionStore()
        ptp.register_attention_control(h, controller)
        lat0 = torch.from_numpy(hashrand.normalish((1, 4, 32, 32), 9200 + ci))
        gen = real_generator("cpu").manual_seed(1234 + ci)
        out = h(prompt=cfg.prompt, attention_store=controller, attention_res=16, guidance_scale=7.5, generator=gen,
                num_inference_steps=steps, max_iter_to_alter=max_alter, run_standard_sd=False, thresholds=cfg.thresholds,
                scale_factor=sf, scale_range=(1.0, 0.5), smooth_attentions=True, sigma=0.5, kernel_size=3, sd_2_1=False,
                latents=lat0.clone(), return_dict=False)
        log = "".join(helpers.lines)
        n_b1 = sum(1 for b, g in shim.calls if b == 1)
        n_b2 = sum(1 for b, g in shim.calls if b == 2)
        n_bwd = log.count("gradient size average")
        losses = [float(l.split("Loss:")[1]) for l in log.splitlines() if l.startswith("Iteration") and "Loss:" in l]
        import re
        fin = [float(re.search(r"tensor\(\[([^\]]+)\]", l).group(1)) for l in log.splitlines()
               if "Finished with loss of" in l]
        arrs[f"{name}.final_latents"] = f32(h.final_latents)
        arrs[f"{name}.iter_losses"] = np.array(losses, np.float32)
        arrs[f"{name}.refine_final_losses"] = np.array(fin, np.float32)
        meta.append({"name": name, "steps": steps, "thresholds": {str(k): v for k, v in thr.items()},
                     "only_update_on_threshold_steps": only_thr, "max_iter_to_alter": max_alter, "hyper": hyper,
                     "scale_factor": sf, "unet_seed": 9000 + 1000 * ci, "embed_seed": 9100 + ci,
                     "latent_seed": 9200 + ci, "renoise_seed": 1234 + ci, "fwd_b1": n_b1, "fwd_b2": n_b2, "bwd": n_bwd,
                     "subiterations": log.count("subiteration:"), "call_sequence": "".join(str(b) for b, g in shim.calls),
                     "final_abs_mean": float(h.final_latents.abs().mean())})
        print(name, meta[-1]["fwd_b1"], meta[-1]["bwd"], meta[-1]["fwd_b2"], meta[-1]["final_abs_mean"], losses[:4], fin)
    torch.Generator, torch.randn = real_generator, real_randn
    np.savez_compressed(OUT / "g9_loop.npz", **arrs)
    (OUT / "g9_loop.json").write_text(json.dumps(meta, indent=1))

# ----------------------------------------------------------------------------- G10 custom loss plugin
def g10_custom_loss():
    import run as ref_run  # pyrallis is a placeholder; ToLeftOf / CustomLossBase are plain classes
    h = Harness()
    mp = "a [cat:.2,.5] and a [vase:.7,.5] [CustomLoss:toLeftOf (cat, vase)]"
    cfg = fresh_config(mp)
    cfg.registered_loss_functions = {"toLeftOf": ref_run.ToLeftOf()}
    cfg.stable = types.SimpleNamespace(tokenizer=h.tokenizer)
    state.config = cfg
    state.curHyperParams = dict(state.hyperParameterOverrides)
    cfg.prompt, cfg.meta_info, cfg.custom_loss = helpers.parse_prompt(mp)
    arrs = {}
    for name, kind, seed in (("bos", "bos", 31), ("sharp", "sharp", 32)):
        A0 = make_maps(kind, seed)
        A_leaf = A0.clone().requires_grad_(True)
        text = torch.softmax(A_leaf[:, :, 1:-1] * 100, dim=-1)
        fn, args = cfg.custom_loss["toLeftOf"]
        v = fn.calc_loss(text, args)
        (gA,) = torch.autograd.grad(v.sum(), [A_leaf], allow_unused=True)
        arrs[f"{name}.A"] = f32(A0)
        arrs[f"{name}.loss"] = f32(v)
        arrs[f"{name}.dA"] = f32(gA if gA is not None else torch.zeros_like(A0))
    np.savez_compressed(OUT / "g10_custom_loss.npz", **arrs)
    (OUT / "g10_custom_loss.json").write_text(json.dumps({"meta_prompt": mp, "prompt": cfg.prompt, "args": cfg.custom_loss["toLeftOf"][1],
                                                         "meta_info": ser_meta(cfg.meta_info)}))


# ----------------------------------------------------------------------------- G11 the reference's own UNet forward
def g11_unet_forward():
    """`GuidedAttention.forward` (pipeline_guided_attention.py:583-743) of the reference, driven on the build's blocks
    through `_DiffusersFacade`: pins the top-level wiring of guided_attention_amd/unet.py:forward (time path, skip
    order and slicing, `forward_upsample_size` / `upsample_size`, `center_input_sample`, the output head)."""
    sys.path.insert(0, str(OUT.parent.parent))
    from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
    from oracle.attention import OracleStore
    from oracle.pipeline import install_processors
    cases = [
        # name, config, latent (H, W), batch, timestep, seed
        ("tiny_32", UNetConfig.tiny(32, 48), (32, 32), 1, 981, 11000),
        ("tiny_36x28_b2", UNetConfig.tiny(32, 48), (36, 28), 2, 441, 11100),     # not multiples of 8: upsample_size path
        ("tiny_centered", UNetConfig(sample_size=16, block_out_channels=(32, 64, 128, 128), attention_head_dim=2,
                                     cross_attention_dim=48, center_input_sample=True), (16, 16), 1, 1, 11200),
        ("sd21_like_24", UNetConfig(sample_size=24, block_out_channels=(32, 64, 128, 128), attention_head_dim=(1, 2, 4, 4),
                                    cross_attention_dim=40, use_linear_projection=True), (24, 24), 1, 721.5, 11300),
    ]
    arrs, meta = {}, []
    for name, cfg, (H, W), B, t, seed in cases:
        real = hash_init_(UNet2DConditionModel(cfg), seed).float()
        with torch.no_grad():   # non-zero biases so that every bias path of the wiring is visible
            for pi, (pn, p) in enumerate(real.named_parameters()):
                if pn.endswith("bias"):
                    p.copy_(torch.from_numpy(hashrand.normalish(tuple(p.shape), seed + 5000 + pi) * np.float32(0.05)))
        for p in real.parameters():
            p.requires_grad_(False)
        install_processors(real, OracleStore())
        fac = _DiffusersFacade(real)
        h = Harness()
        h.unet = fac
        x = torch.from_numpy(hashrand.normalish((B, 4, H, W), seed + 1))
        ctx = torch.from_numpy(hashrand.normalish((B, 77, cfg.cross_attention_dim), seed + 2))
        with torch.no_grad():
            (y,) = pga.GuidedAttention.forward(h, x, t, ctx, return_dict=False)
            y_t = pga.GuidedAttention.forward(h, x, torch.tensor(t), ctx, return_dict=False)[0]   # 0-d tensor timestep
        assert torch.equal(y, y_t)
        arrs[f"{name}.out"] = f32(y)
        meta.append({"name": name, "seed": seed, "shape": [B, 4, H, W], "timestep": t,
                     "config": {"sample_size": cfg.sample_size, "block_out_channels": list(cfg.block_out_channels),
                                "attention_head_dim": cfg.attention_head_dim if isinstance(cfg.attention_head_dim, int)
                                else list(cfg.attention_head_dim), "cross_attention_dim": cfg.cross_attention_dim,
                                "use_linear_projection": cfg.use_linear_projection,
                                "center_input_sample": cfg.center_input_sample},
                     "out_abs_mean": float(y.abs().mean())})
        print("g11", name, tuple(y.shape), meta[-1]["out_abs_mean"])
    np.savez_compressed(OUT / "g11_unet_forward.npz", **arrs)
    (OUT / "g11_unet_forward.json").write_text(json.dumps(meta, indent=1))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)  # deterministic reductions
    g1_gaussian()
    g2_parse()
    g3_masks()
    g4_loss()
    g5_threshold()
    g6_processor()
    g7_aggregate()
    g8_update()
    g9_loop()
    g10_custom_loss()
    g11_unet_forward()
    for p in sorted(OUT.glob("g*")):
        print(f"{p.name:32s} {p.stat().st_size:9d} B")


if __name__ == "__main__":
    main()
