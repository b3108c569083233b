"""Oracle (TEST INFRASTRUCTURE ONLY): the Gaussian-smoothed bounding-box loss over the
aggregated cross-attention maps, restated on CPU.

Two independent forms are given on purpose:
  * `loss_torch`   — differentiable fp32/fp64 PyTorch-CPU form (vectorised masks/index grids
                     instead of the reference's Python pixel loops); autograd provides dA.
  * `loss_and_grad_numpy` — closed-form numpy forward AND hand-derived backward, i.e. the
                     same algebra the HIP kernels `ga_smooth_loss_fwd/bwd` implement.
Both are pinned to tests/golden/g4_loss.npz (the reference's own output + autograd gradient).

Follows (paths relative to the reference checkout):
  pipeline_guided_attention.py:201-296   _compute_max_attention_per_index
  pipeline_guided_attention.py:359-451   group_losses_by_sumprompt / get_centering_loss / _compute_loss
  pipeline_guided_attention.py:1074-1088 meets_threshold
  utils/gaussian_smoothing.py:21-71      GaussianSmoothing
  utils/helpers.py:15-30,164-173,215-277 Rect / inside_box / calculate_bounding_box_losses
"""
import math
from collections import OrderedDict

import numpy as np
import torch

# utils/shared_state.py:21 — the hyper-parameters that feed the loss
DEFAULT_HYPER = {"strict": False, "inside_loss_scale": .2, "outside_loss_scale": .2, "shrink_factor": .15,
                 "thresholds": {0: 1.}, "use_optimizer": False, "recurse_until": 14, "recurse_steps": 3}


def gaussian_weights(kernel_size=3, sigma=0.5):
    """utils/gaussian_smoothing.py:21-47.  Note the reference's exponent is -((x-mu)/(2 sigma))^2,
    not the textbook -(x-mu)^2/(2 sigma^2); the normalisation makes the 1/(sigma sqrt(2 pi)) factor moot
    but it is kept so that fp32 rounding follows the reference."""
    ax = torch.arange(kernel_size, dtype=torch.float32)
    mean = (kernel_size - 1) / 2
    g1 = 1 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-((ax - mean) / (2 * sigma)) ** 2)
    k2 = g1[:, None] * g1[None, :]
    return (k2 / k2.sum()).numpy()


def scaled_rect(rect, res):
    """helpers.py:28-30 Rect.of_size: every field times float(res/size) with size = 1 (float64)."""
    ratio = float(res / 1)
    return tuple(float(v) * ratio for v in rect)


def inside_mask(rect, res, shrink):
    """helpers.py:164-173 inside_box evaluated at every pixel centre (jj+.5, ii+.5), float64,
    closed interval on both sides, box shrunk by shrink*width / shrink*height per side."""
    x, y, w, h = scaled_rect(rect, res)
    ox = shrink * w
    oy = shrink * h
    c = np.arange(res, dtype=np.float64) + 0.5
    in_x = (c >= (x + ox)) & (c <= (x + w - ox))
    in_y = (c >= (y + oy)) & (c <= (y + h - oy))
    return (in_y[:, None] & in_x[None, :])


def rect_center(rect):
    """helpers.py:26-27"""
    x, y, w, h = rect
    return (x + w / 2.0, y + h / 2.0)


class TokenPlan:
    """What run.py:81-91 leaves in config.token_dict, as plain data.
    entries: list of dict(index=int (position in the 77-token prompt), kind='BOX'|'COOR',
             geom=(x,y,w,h)|(x,y) in fractions of the image, subprompt=str)."""

    def __init__(self, entries, hyper=None, sub_prompt_avg_within=False):
        self.entries = list(entries)
        self.hyper = dict(DEFAULT_HYPER)
        if hyper:
            self.hyper.update(hyper)
        self.avg_within = bool(sub_prompt_avg_within)

    @classmethod
    def from_golden(cls, meta):
        ents = []
        for k, v in meta["token_dict"].items():
            ents.append({"index": int(k), "kind": v["loss_type"], "geom": tuple(v["loss"]),
                         "subprompt": v["subprompt"]})
        hyper = {k: v for k, v in meta["hyper"].items() if k != "thresholds"}
        return cls(ents, hyper, meta["sub_prompt_avg_within"])

    def token_weights(self):
        """group_losses_by_sumprompt (pipeline:359-387): sum within a sub-prompt, or mean when
        sub_prompt_avg_within; sum across sub-prompts  ==  a per-token weight of 1 or 1/count."""
        counts = OrderedDict()
        for e in self.entries:
            counts[e["subprompt"]] = counts.get(e["subprompt"], 0) + 1
        return [1.0 / counts[e["subprompt"]] if self.avg_within else 1.0 for e in self.entries]


def text_slice(ntok, normalize_eot, n_prompt_tokens):
    """pipeline:209-217: tokens [1, last) with last = -1 (i.e. ntok-1) or, for SD-2.1, the index
    of the first end-of-text token = len(tokenizer(prompt).input_ids) - 1."""
    last = (n_prompt_tokens - 1) if normalize_eot else (ntok - 1)
    return 1, last


def loss_torch(A, plan, smooth=True, sigma=0.5, kernel_size=3, normalize_eot=False, n_prompt_tokens=None,
               dtype=torch.float32):
    """A: (res,res,ntok) tensor (may require grad).  Returns dict with per-token lists and `loss`.
    Generalises the reference's literal 16 / 15. to res / (res-1) (identical at res=16)."""
    res = A.shape[0]
    ntok = A.shape[-1]
    first, last = text_slice(ntok, normalize_eot, n_prompt_tokens)
    A = A.to(dtype)
    S = torch.softmax(A[:, :, first:last] * 100, dim=-1)                      # pipeline:217-219
    hp = plan.hyper
    G = torch.from_numpy(gaussian_weights(kernel_size, sigma)).to(dtype)
    pad = kernel_size // 2  # the reference hard-codes 1 (== 3 // 2); other sizes are an extrapolation
    jj = (torch.arange(res, dtype=dtype) + 0.5)[None, :]
    ii = (torch.arange(res, dtype=dtype) + 0.5)[:, None]
    out = {"max": [], "col": [], "row": [], "inside": [], "outside": [], "token_loss": [], "unscaled": []}
    total = torch.zeros((), dtype=dtype)
    cw = hp.get("bb_center_weight", .05)
    for e, w_tok in zip(plan.entries, plan.token_weights()):
        M = S[:, :, e["index"] - 1]                                           # pipeline:228,249
        if smooth:                                                             # pipeline:251-254
            inp = torch.nn.functional.pad(M[None, None], (pad, pad, pad, pad), mode="reflect")
            M = torch.nn.functional.conv2d(inp, G[None, None])[0, 0]
        out["max"].append(M.max())                                             # pipeline:255
        Pn = M / M.sum()                                                       # pipeline:263
        col = (jj * Pn).sum()                                                  # pipeline:264-268
        row = (ii * Pn).sum()
        out["col"].append(col)
        out["row"].append(row)
        if e["kind"] == "BOX":
            mask = torch.from_numpy(inside_mask(e["geom"], res, hp["shrink_factor"]))
            if hp.get("strict", False):                                        # helpers.py:216-264
                Wn, _, n_in = strict_weights(e["geom"], res, hp["shrink_factor"])
                Wt = torch.from_numpy(Wn).to(dtype)
                at_most = 1.0 / n_in
                inside = (Wt * 2.0 * torch.clamp(at_most - Pn, min=0))[mask].sum()
                outside = (Wt * torch.clamp(Pn, min=0))[~mask].sum()
            else:
                inside = 1.0 - Pn[mask].sum()                                  # helpers.py:265-277
                outside = Pn[~mask].sum()
            center = rect_center(e["geom"])
        else:
            inside = torch.zeros((), dtype=dtype)
            outside = torch.zeros((), dtype=dtype)
            center = e["geom"]
        # pipeline:391-395 get_centering_loss (max(0, .) of a non-negative value is the identity)
        centering = (col - center[0] * res).abs() / (res - 1.) + 4. * (row - center[1] * res).abs() / (res - 1.)
        if e["kind"] == "BOX":                                                 # pipeline:415-438
            unscaled = inside + outside
            item = hp["inside_loss_scale"] * inside + hp["outside_loss_scale"] * outside * 3
            if cw > 0:
                item = item + cw * centering
        else:                                                                  # pipeline:409-414
            item = centering
            unscaled = centering
        out["inside"].append(inside)
        out["outside"].append(outside)
        out["token_loss"].append(item)
        out["unscaled"].append(unscaled)
        total = total + w_tok * item
    out["loss"] = total
    return out


def strict_weights(rect, res, shrink):
    """helpers.py:216-246: the per-pixel weight table of calculate_bounding_box_losses, normalised separately over the
    inside and the outside pixels.  Inside: np.interp of the normalised distance from the box centre over
    xp [0, .333, .666, 1], fp [3, 2.5, 1, .2]; outside: 1.  Geometry in float64, table and sums in fp32 (the
    reference writes into a `tr.ones(16,16)` fp32 tensor and accumulates the sums in pixel order)."""
    x, y, w, h = scaled_rect(rect, res)
    mask = inside_mask(rect, res, shrink)
    cx, cy = x + w / 2.0, y + h / 2.0
    W = np.ones((res, res), np.float32)
    for ii in range(res):
        for jj in range(res):
            if mask[ii, jj]:
                d = math.sqrt(math.pow(2 * (cx - (jj + .5)) / w, 2) + math.pow(2 * (cy - (ii + .5)) / h, 2)) / math.sqrt(2)
                W[ii, jj] = np.float32(np.interp(d, [0, .333, .666, 1.0], [3, 2.5, 1, .2]))
    s_in = np.float32(0)
    s_out = np.float32(0)
    for ii in range(res):
        for jj in range(res):
            if mask[ii, jj]:
                s_in = np.float32(s_in + W[ii, jj])
            else:
                s_out = np.float32(s_out + W[ii, jj])
    Wn = np.where(mask, W / (s_in if s_in != 0 else np.float32(1)), W / (s_out if s_out != 0 else np.float32(1)))
    return Wn.astype(np.float32), mask, int(mask.sum())


def loss_reference_loops(A, plan, smooth=True, sigma=0.5, kernel_size=3, normalize_eot=False, n_prompt_tokens=None):
    """The loss evaluated the way the reference executes it: Python double loops over the res x res pixels doing
    scalar fp32 tensor operations, each an autograd node (pipeline:248-281, helpers.py:215-277), including the
    `strict` branch (helpers.py:250-264).  Slow on purpose — it is the timing leg "reference-style loop loss" of
    bench.py's cpu_baseline and the second, independent form the strict mode is checked with.
    A: (res, res, ntok) fp32 NON-LEAF tensor (the reference scales a view of it in place)."""
    res, ntok = A.shape[0], A.shape[-1]
    first, last = text_slice(ntok, normalize_eot, n_prompt_tokens)
    hp = plan.hyper
    text = A[:, :, first:last]
    text *= 100          # in place on the view, as the reference does (pipeline:217-218)
    S = torch.nn.functional.softmax(text, dim=-1)
    G = torch.from_numpy(gaussian_weights(kernel_size, sigma))
    pad = kernel_size // 2
    cw = hp.get("bb_center_weight", .05)
    out = {"max": [], "col": [], "row": [], "inside": [], "outside": [], "token_loss": [], "unscaled": []}
    groups = OrderedDict()
    for e in plan.entries:
        image = S[:, :, e["index"] - 1]
        if smooth:
            inp = torch.nn.functional.pad(image[None, None], (pad, pad, pad, pad), mode="reflect")
            image = torch.nn.functional.conv2d(inp, G[None, None])[0, 0]
        out["max"].append(image.max())
        Pn = image / image.sum()
        col = torch.zeros(1)
        row = torch.zeros(1)
        for ii in range(res):
            for jj in range(res):
                col = col + (jj + .5) * Pn[ii][jj]
                row = row + (ii + .5) * Pn[ii][jj]
        if e["kind"] == "BOX":
            Wn, mask, n_in = strict_weights(e["geom"], res, hp["shrink_factor"])
            at_most = 1.0 / n_in           # ZeroDivisionError when no pixel centre is inside, as in the reference
            zero = torch.zeros(1)
            if hp.get("strict", False):
                inside = torch.zeros(1)
                outside = torch.zeros(1)
                for ii in range(res):
                    for jj in range(res):
                        if mask[ii, jj]:
                            inside = inside + float(Wn[ii, jj]) * (2. * max(zero, at_most - Pn[ii, jj]))
                        else:
                            outside = outside + float(Wn[ii, jj]) * max(zero, Pn[ii, jj] - zero)
            else:
                s_in = torch.zeros(1)
                s_out = torch.zeros(1)
                for ii in range(res):
                    for jj in range(res):
                        if mask[ii, jj]:
                            s_in = s_in + Pn[ii, jj]
                        else:
                            s_out = s_out + Pn[ii, jj]
                inside, outside = 1. - s_in, s_out
            center = rect_center(e["geom"])
        else:
            inside = outside = torch.zeros(1)
            center = e["geom"]
        centering = (col - center[0] * res).abs() / (res - 1.) + 4. * (row - center[1] * res).abs() / (res - 1.)
        if e["kind"] == "BOX":
            unscaled = inside + outside
            item = hp["inside_loss_scale"] * inside + hp["outside_loss_scale"] * outside * 3
            if cw > 0:
                item = item + cw * centering
        else:
            item = unscaled = centering
        for k, v in (("col", col), ("row", row), ("inside", inside), ("outside", outside), ("token_loss", item),
                     ("unscaled", unscaled)):
            out[k].append(v)
        groups.setdefault(e["subprompt"], []).append(item)
    # group_losses_by_sumprompt (pipeline:359-387): per sub-prompt 0 + v/cnt (or v) in order, then 0 + the group totals
    total = torch.zeros(1)
    for vals in groups.values():
        sub = torch.zeros(1)
        for v in vals:
            sub = sub + (v / len(vals) if plan.avg_within else v)
        total = total + sub
    out["loss"] = total
    return out


def subprompt_sums(plan, values):
    """pipeline:359-387 for a list of per-token values (e.g. the unscaled losses)."""
    sums = OrderedDict()
    for e, w, v in zip(plan.entries, plan.token_weights(), values):  # fp32 accumulation, as the reference's tensors
        sums[e["subprompt"]] = np.float32(sums.get(e["subprompt"], np.float32(0.0)) + np.float32(w) * np.float32(float(v)))
    return sums


def meets_threshold(i, thresholds, sub_sums):
    """pipeline:1074-1088.  `sub_sums`: mapping sub-prompt -> unscaled loss sum."""
    if (i not in thresholds and i != -1) or len(thresholds) == 0:
        return True
    thr = list(thresholds.values())[-1] if i == -1 else thresholds[i]
    for v in sub_sums.values():
        # the reference compares an fp32 tensor with a Python float: the float is rounded to fp32 first
        if np.float32(v) > np.float32(thr):
            return False
    return True


# ------------------------------------------------------------------ closed form (numpy), fwd + bwd
def _reflect_index(i, n):
    if i < 0:
        return -i
    if i >= n:
        return 2 * (n - 1) - i
    return i


def smooth_matrix(res, kernel_size, sigma):
    """1-D operator (res x res) of reflect-pad + 1-D correlation; the 2-D smoothing is Rm @ M @ Rm.T
    because the normalised 2-D kernel is the outer product of the normalised 1-D kernel."""
    k2 = gaussian_weights(kernel_size, sigma).astype(np.float64)
    k1 = k2.sum(1)  # separable: rows sum to the 1-D normalised kernel
    pad = kernel_size // 2
    Rm = np.zeros((res, res), np.float64)
    for i in range(res):
        for t in range(kernel_size):
            Rm[i, _reflect_index(i + t - pad, res)] += k1[t]
    return Rm


def loss_and_grad_numpy(A, plan, smooth=True, sigma=0.5, kernel_size=3, normalize_eot=False,
                        n_prompt_tokens=None):
    """Closed-form forward and backward in float64.  Returns (terms dict, dLoss/dA (res,res,ntok))."""
    A = np.asarray(A, np.float64)
    res, _, ntok = A.shape
    first, last = text_slice(ntok, normalize_eot, n_prompt_tokens)
    z = 100.0 * A[:, :, first:last]
    z = z - z.max(-1, keepdims=True)
    S = np.exp(z)
    S /= S.sum(-1, keepdims=True)
    hp = plan.hyper
    Rm = smooth_matrix(res, kernel_size, sigma) if smooth else np.eye(res)
    jj = (np.arange(res) + 0.5)[None, :]
    ii = (np.arange(res) + 0.5)[:, None]
    cw = hp.get("bb_center_weight", .05)
    dS = np.zeros_like(S)
    terms = {"max": [], "col": [], "row": [], "inside": [], "outside": [], "token_loss": [], "unscaled": []}
    total = 0.0
    for e, w_tok in zip(plan.entries, plan.token_weights()):
        k = e["index"] - 1
        M = Rm @ S[:, :, k] @ Rm.T
        s = M.sum()
        Pn = M / s
        col = (jj * Pn).sum()
        row = (ii * Pn).sum()
        Wn = at_most = None
        if e["kind"] == "BOX":
            mask = inside_mask(e["geom"], res, hp["shrink_factor"])
            if hp.get("strict", False):
                Wn, _, n_in = strict_weights(e["geom"], res, hp["shrink_factor"])
                Wn = Wn.astype(np.float64)
                at_most = float(np.float32(1.0 / n_in))
                inside = (Wn * 2.0 * np.maximum(at_most - Pn, 0.0))[mask].sum()
                outside = (Wn * np.maximum(Pn, 0.0))[~mask].sum()
            else:
                inside = 1.0 - Pn[mask].sum()
                outside = Pn[~mask].sum()
            cx, cy = rect_center(e["geom"])
            w_in, w_out, w_c = hp["inside_loss_scale"], 3.0 * hp["outside_loss_scale"], (cw if cw > 0 else 0.0)
        else:
            mask = np.zeros((res, res), bool)
            inside = outside = 0.0
            cx, cy = e["geom"]
            w_in = w_out = 0.0
            w_c = 1.0
        dc = col - cx * res
        dr = row - cy * res
        centering = abs(dc) / (res - 1.) + 4. * abs(dr) / (res - 1.)
        if e["kind"] == "BOX":
            item = w_in * inside + w_out * outside + w_c * centering
            unscaled = inside + outside
        else:
            item = unscaled = centering
        # d item / d Pn
        g = w_c * (np.sign(dc) / (res - 1.) * jj + 4. * np.sign(dr) / (res - 1.) * ii) * np.ones((res, res))
        if e["kind"] == "BOX":
            if Wn is not None:  # hinge terms: gradient only where the hinge is open
                g = g + np.where(mask, np.where(at_most - Pn > 0, -2.0 * w_in * Wn, 0.0), np.where(Pn > 0, w_out * Wn, 0.0))
            else:
                g = g + np.where(mask, -w_in, w_out)
        gM = (g - (g * Pn).sum()) / s                  # through Pn = M / sum(M)
        dS[:, :, k] += w_tok * (Rm.T @ gM @ Rm)        # adjoint of reflect-pad + correlation
        for name, v in (("max", M.max()), ("col", col), ("row", row), ("inside", inside), ("outside", outside),
                        ("token_loss", item), ("unscaled", unscaled)):
            terms[name].append(v)
        total += w_tok * item
    terms["loss"] = total
    dz = S * (dS - (dS * S).sum(-1, keepdims=True))    # softmax backward
    dA = np.zeros_like(A)
    dA[:, :, first:last] = 100.0 * dz
    return terms, dA
