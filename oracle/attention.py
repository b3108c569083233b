"""Oracle (TEST INFRASTRUCTURE ONLY): the attention-store capture path on CPU.

Follows (paths relative to the reference checkout):
  utils/ptp_utils.py:59-93    AttendExciteCrossAttnProcessor.__call__
  utils/ptp_utils.py:97-146   get_attention_scores, including the paint-with-words branch :113-138 (off by
                              default: `paint_with_words_stop` = 0)
  utils/ptp_utils.py:178-270  AttentionControl / AttentionStore
  utils/ptp_utils.py:273-289  aggregate_attention
"""
import numpy as np
import torch


def head_split(t, heads):
    """diffusers 0.12.1 CrossAttention.head_to_batch_dim: (B, N, h*d) -> (B*h, N, d)."""
    b, n, c = t.shape
    return t.reshape(b, n, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, n, c // heads)


def head_merge(t, heads):
    """diffusers 0.12.1 CrossAttention.batch_to_head_dim: (B*h, N, d) -> (B, N, h*d)."""
    bh, n, d = t.shape
    return t.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3).reshape(bh // heads, n, d * heads)


def attention_probs(q, k, scale, pww=None):
    """ptp_utils.py:97-146 with the default flags: softmax_j(scale * q k^T), computed in the input dtype.
    pww = (mask (N, 77), log(1 + sigma_t)): the paint-with-words branch, scores + mask * .4 * scores.max() * log(...)
    with the maximum over the WHOLE tensor and inside the autograd graph (:134)."""
    scores = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype), q, k.transpose(-1, -2),
                           beta=0, alpha=scale)
    if pww is not None:
        mask, log1p_sigma = pww
        scores = scores + mask.to(scores.dtype).unsqueeze(0).repeat((q.shape[0], 1, 1)) * .4 * scores.max() * log1p_sigma
    return scores.softmax(dim=-1).to(q.dtype)


def paint_with_words_mask(token_boxes, n_pixels, shrink, weight):
    """ptp_utils.py:113-131: (N, 77) mask with `weight` inside each BOX token's rectangle (scaled to the layer's
    hw = int(N ** .5), shrunk by shrink_factor, pixel centres) in that token's column.  token_boxes: {index: (x,y,w,h)}."""
    from . import loss as oloss
    hw = int(n_pixels ** .5)
    mask = torch.zeros((hw, hw, 77))
    for idx, rect in token_boxes.items():
        mask[:, :, idx][torch.from_numpy(oloss.inside_mask(tuple(rect), hw, shrink))] = weight
    return mask.reshape(hw * hw, 77)


class OracleStore:
    """ptp_utils.py:178-270 (AttentionControl + AttentionStore, save_global_store=False)."""

    def __init__(self, max_pixels=32 ** 2):
        self.num_att_layers = -1
        self.max_pixels = max_pixels
        self.reset()

    @staticmethod
    def empty():
        return {f"{p}_{k}": [] for k in ("cross", "self") for p in ("down", "mid", "up")}

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0
        self.step_store = self.empty()
        self.attention_store = {}

    def __call__(self, probs, is_cross, place):
        if probs.shape[1] <= self.max_pixels:                       # :228
            self.step_store[f"{place}_{'cross' if is_cross else 'self'}"].append(probs)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:               # :197-201
            self.cur_att_layer = 0
            self.cur_step += 1
            self.attention_store = self.step_store                 # :232-243
            self.step_store = self.empty()


def aggregate(store_dict, res, from_where, is_cross):
    """ptp_utils.py:273-289 with select = 0: mean over every head-map (all batch entries) whose
    pixel count is res^2, in `from_where` order."""
    maps = []
    for loc in from_where:
        for item in store_dict[f"{loc}_{'cross' if is_cross else 'self'}"]:
            if item.shape[1] == res * res:
                maps.append(item.reshape(-1, res, res, item.shape[-1]))
    out = torch.cat(maps, dim=0)
    return out.sum(0) / out.shape[0]


class OracleAttnProcessor:
    """Plain-PyTorch processor with the reference's protocol
    proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None) -> hidden_states."""

    def __init__(self, store, place, pww=None):
        """pww: None, or a callable (n_pixels) -> (mask (N, 77), log(1 + sigma_t)) | None for the current step."""
        self.store = store
        self.place = place
        self.pww = pww

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None):
        is_cross = encoder_hidden_states is not None
        ctx = encoder_hidden_states if is_cross else hidden_states
        q = head_split(attn.to_q(hidden_states), attn.heads)
        k = head_split(attn.to_k(ctx), attn.heads)
        v = head_split(attn.to_v(ctx), attn.heads)
        pww = self.pww(q.shape[1]) if (self.pww is not None and k.shape[1] == 77) else None
        probs = attention_probs(q, k, attn.scale, pww)
        self.store(probs, is_cross, self.place)
        out = head_merge(torch.bmm(probs, v), attn.heads)
        out = attn.to_out[0](out)
        return attn.to_out[1](out)


# ------------------------------------------------------------------ closed form (numpy float64)
def capture_fwd_numpy(Q, K, V, scale):
    """Q (BH,N,d), K,V (BH,Kt,d) -> P (BH,N,Kt), O (BH,N,d)."""
    Q, K, V = (np.asarray(x, np.float64) for x in (Q, K, V))
    S = scale * Q @ K.transpose(0, 2, 1)
    S -= S.max(-1, keepdims=True)
    P = np.exp(S)
    P /= P.sum(-1, keepdims=True)
    return P, P @ V


def capture_bwd_numpy(Q, K, V, scale, dO, dP_direct=None):
    """Backward of (P, O) w.r.t. Q, K, V given dO and an optional direct gradient on P
    (the loss reaches the latents through P of the res^2 cross-attention layers)."""
    Q, K, V, dO = (np.asarray(x, np.float64) for x in (Q, K, V, dO))
    P, _ = capture_fwd_numpy(Q, K, V, scale)
    dP = dO @ V.transpose(0, 2, 1)
    if dP_direct is not None:
        dP = dP + np.asarray(dP_direct, np.float64)
    dS = P * (dP - (dP * P).sum(-1, keepdims=True))
    dQ = scale * dS @ K
    dK = scale * dS.transpose(0, 2, 1) @ Q
    dV = P.transpose(0, 2, 1) @ dO
    return dQ, dK, dV


def full_bwd_numpy(Q, K, V, scale, dO):
    """All three gradients of O = softmax(scale Q K^T) V (the algebra of ga_self_attn_bwd)."""
    return capture_bwd_numpy(Q, K, V, scale, dO, None)
