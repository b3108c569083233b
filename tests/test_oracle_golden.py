"""The oracle against the fixtures produced by the reference's own functions
(tests/golden/make_golden.py).  CPU only."""
import math

import numpy as np
import pytest
import torch

import hashrand
from conftest import load_json, load_npz
from oracle import attention as oattn
from oracle import loss as oloss


# ---------------------------------------------------------------- G1 Gaussian weights / smoothing
@pytest.mark.parametrize("k,s", [(3, 0.5), (3, 1.0), (5, 1.0), (5, 0.75)])
def test_gaussian_weights(k, s):
    g = load_npz("g1_gaussian.npz")
    np.testing.assert_allclose(oloss.gaussian_weights(k, s), g[f"k{k}_s{s}"], rtol=0, atol=1e-7)


def test_gaussian_known_values():
    w = oloss.gaussian_weights(3, 0.5)  # centre .331911, edge .122103, corner .044919
    assert abs(w[1, 1] - 0.331911) < 1e-6 and abs(w[0, 1] - 0.122103) < 1e-6 and abs(w[0, 0] - 0.044919) < 1e-6


def test_smoothing_forward_matches_reference():
    g = load_npz("g1_gaussian.npz")
    Rm = oloss.smooth_matrix(16, 3, 0.5)
    out = Rm @ g["fwd_in"].astype(np.float64) @ Rm.T
    np.testing.assert_allclose(out, g["fwd_out"], rtol=0, atol=2e-7)


# ---------------------------------------------------------------- G3 inside-box masks
def test_inside_masks():
    cases = load_json("g3_inside_box.json")
    arrs = load_npz("g3_inside_box.npz")
    assert len(cases) == 72
    for c in cases:
        m = oloss.inside_mask(tuple(c["rect"]), c["res"], c["shrink"])
        assert np.array_equal(m.astype(np.uint8), arrs[f"mask{c['id']}"]), c
        assert int(m.sum()) == c["count"]
        np.testing.assert_allclose(oloss.scaled_rect(c["rect"], c["res"]), c["scaled"], rtol=0, atol=0)
        np.testing.assert_allclose(oloss.rect_center(c["rect"]), c["center"], rtol=0, atol=0)


def test_baseline_prompt_pixel_counts():
    # robot 24 px (rows 6-11, cols 11-14), blue vase 30 px (rows 6-11, cols 4-8)
    m = oloss.inside_mask((.6, .3, .4, .55), 16, .15)
    assert m.sum() == 24 and m[6:12, 11:15].all()
    m = oloss.inside_mask((.2, .3, .4, .55), 16, .15)
    assert m.sum() == 30 and m[6:12, 4:9].all()


# ---------------------------------------------------------------- G4 loss + gradient
G4_META = load_json("g4_loss.json")


@pytest.mark.parametrize("meta", G4_META, ids=[m["name"] for m in G4_META])
def test_loss_torch_form(meta):
    g = load_npz("g4_loss.npz")
    n = meta["name"]
    plan = oloss.TokenPlan.from_golden(meta)
    A = torch.from_numpy(g[f"{n}.A"]).requires_grad_(True)
    r = oloss.loss_torch(A, plan, smooth=meta["smooth"], sigma=meta["sigma"], kernel_size=meta["kernel_size"],
                         normalize_eot=meta["normalize_eot"], n_prompt_tokens=meta["n_prompt_tokens"])
    tol = dict(rtol=2e-5, atol=2e-6)
    for key in ("max", "col", "row", "inside", "outside"):
        np.testing.assert_allclose([float(v) for v in r[key]], g[f"{n}.{key}"], err_msg=key, **tol)
    np.testing.assert_allclose([float(v) for v in r["token_loss"]], g[f"{n}.losses"], **tol)
    np.testing.assert_allclose([float(v) for v in r["unscaled"]], g[f"{n}.unscaled"], **tol)
    np.testing.assert_allclose(float(r["loss"]), g[f"{n}.loss"], **tol)
    assert [e["index"] for e in plan.entries] == list(g[f"{n}.tok"])
    sums = oloss.subprompt_sums(plan, r["unscaled"])
    np.testing.assert_allclose(list(sums.values()), g[f"{n}.sub_unscaled"], **tol)
    (dA,) = torch.autograd.grad(r["loss"], [A])
    ref = g[f"{n}.dA"]
    scale = np.abs(ref).max()
    np.testing.assert_allclose(dA.numpy(), ref, rtol=0, atol=2e-5 * scale + 1e-9)


@pytest.mark.parametrize("meta", G4_META, ids=[m["name"] for m in G4_META])
def test_loss_closed_form(meta):
    g = load_npz("g4_loss.npz")
    n = meta["name"]
    plan = oloss.TokenPlan.from_golden(meta)
    terms, dA = oloss.loss_and_grad_numpy(g[f"{n}.A"], plan, smooth=meta["smooth"], sigma=meta["sigma"],
                                          kernel_size=meta["kernel_size"], normalize_eot=meta["normalize_eot"],
                                          n_prompt_tokens=meta["n_prompt_tokens"])
    tol = dict(rtol=2e-5, atol=2e-6)
    for key in ("max", "col", "row", "inside", "outside"):
        np.testing.assert_allclose(terms[key], g[f"{n}.{key}"], err_msg=key, **tol)
    np.testing.assert_allclose(terms["token_loss"], g[f"{n}.losses"], **tol)
    np.testing.assert_allclose(terms["loss"], g[f"{n}.loss"], **tol)
    ref = g[f"{n}.dA"]
    np.testing.assert_allclose(dA, ref, rtol=0, atol=2e-5 * np.abs(ref).max() + 1e-9)


# ---------------------------------------------------------------- G5 meets_threshold
def test_meets_threshold_table():
    doc = load_json("g5_meets_threshold.json")
    entries = [{"index": int(t), "kind": "BOX", "geom": (0, 0, 1, 1), "subprompt": s}
               for t, s in doc["token_subprompt"].items()]
    plan = oloss.TokenPlan(entries)
    for row in doc["rows"]:
        thr = {int(k): v for k, v in doc["thr_sets"][row["thresholds"]].items()}
        vals = [v for _, v in doc["loss_sets"][row["losses"]]]
        sums = oloss.subprompt_sums(plan, vals)
        assert oloss.meets_threshold(row["i"], thr, sums) == row["result"], row


# ---------------------------------------------------------------- G6 processor
class _Attn(torch.nn.Module):
    def __init__(self, meta):
        super().__init__()
        C, ctx = meta["C"], meta["ctx_dim"]
        self.heads, self.scale = meta["heads"], meta["scale"]
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(ctx, C, bias=False)
        self.to_v = torch.nn.Linear(ctx, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])
        with torch.no_grad():
            for pi, (pn, p) in enumerate(self.named_parameters()):
                assert pn == meta["param_order"][pi]
                w = hashrand.normalish(tuple(p.shape), meta["seed"] + 1 + pi) * np.float32(1.5 / math.sqrt(p.shape[-1]))
                p.copy_(torch.from_numpy(w))


def g6_inputs(meta):
    B, N, C = meta["batch"], meta["N"], meta["C"]
    x = torch.from_numpy(hashrand.normalish((B, N, C), meta["seed"]))
    ctx = None
    if meta["ctx_len"] is not None:
        ctx = torch.from_numpy(hashrand.normalish((B, meta["ctx_len"], meta["ctx_dim"]), meta["seed"] + 7))
    return x, ctx


G6_META = load_json("g6_processor.json")


@pytest.mark.parametrize("meta", G6_META, ids=[m["name"] for m in G6_META])
def test_processor(meta):
    g = load_npz("g6_processor.npz")
    n = meta["name"]
    attn = _Attn(meta)
    x, ctx = g6_inputs(meta)
    x.requires_grad_(True)
    store = oattn.OracleStore()
    store.num_att_layers = 1
    pww = None
    if meta.get("pww"):   # paint-with-words: BASE prompt boxes (tokens 2, 5, 6), shrink .15, the fixture's weight / sigma
        boxes = {2: (.6, .3, .4, .55), 5: (.2, .3, .4, .55), 6: (.2, .3, .4, .55)}
        pww = lambda n_pix: (oattn.paint_with_words_mask(boxes, n_pix, .15, meta["pww"]["weight"]),  # noqa: E731
                             meta["pww"]["log1p_sigma"])
    proc = oattn.OracleAttnProcessor(store, meta["place"], pww)
    qs = {}
    attn.to_q.register_forward_hook(lambda m, i, o: (o.retain_grad(), qs.__setitem__("q", o)) and None)
    out = proc(attn, x, encoder_hidden_states=ctx)
    np.testing.assert_allclose(out.detach().numpy(), g[f"{n}.out"], rtol=1e-5, atol=2e-5)  # fp32 GEMM order
    assert {k: len(v) for k, v in store.attention_store.items()} == meta["store_keys"]
    assert store.cur_step == meta["cur_step"] and store.cur_att_layer == meta["cur_att_layer"]
    key = f"{meta['place']}_{'cross' if ctx is not None else 'self'}"
    R1 = torch.from_numpy(hashrand.normalish(tuple(out.shape), meta["seed"] + 8))
    scal = (out * R1).sum()
    if meta["stored"]:
        P = store.attention_store[key][0]
        np.testing.assert_allclose(P.detach().numpy(), g[f"{n}.P"], rtol=1e-5, atol=1e-7)
        R2 = torch.from_numpy(hashrand.normalish(tuple(P.shape), meta["seed"] + 9))
        scal = scal + (P * R2).sum()
    scal.backward()
    for name, got in (("dx", x.grad), ("dq", qs["q"].grad)):
        ref = g[f"{n}.{name}"]
        np.testing.assert_allclose(got.numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("meta", [m for m in G6_META if m["stored"] and m["ctx_len"] and not m.get("pww")],
                         ids=lambda m: m["name"])
def test_capture_closed_form(meta):
    """numpy closed-form fwd/bwd (the algebra of ga_attn_capture_fwd/bwd) against the reference."""
    g = load_npz("g6_processor.npz")
    n = meta["name"]
    attn = _Attn(meta)
    x, ctx = g6_inputs(meta)
    h = meta["heads"]
    with torch.no_grad():
        q = oattn.head_split(attn.to_q(x), h).numpy()
        k = oattn.head_split(attn.to_k(ctx), h).numpy()
        v = oattn.head_split(attn.to_v(ctx), h).numpy()
    P, O = oattn.capture_fwd_numpy(q, k, v, meta["scale"])
    np.testing.assert_allclose(P, g[f"{n}.P"], rtol=1e-5, atol=1e-7)
    out_shape = g[f"{n}.out"].shape
    R1 = hashrand.normalish(out_shape, meta["seed"] + 8)
    R2 = hashrand.normalish(P.shape, meta["seed"] + 9)
    with torch.no_grad():  # dO = (R1 @ W_out) split into heads
        dOm = torch.from_numpy(R1) @ attn.to_out[0].weight
        dO = oattn.head_split(dOm, h).numpy()
    dQ, _, _ = oattn.capture_bwd_numpy(q, k, v, meta["scale"], dO, R2)
    dq = oattn.head_merge(torch.from_numpy(dQ), h).numpy()
    ref = g[f"{n}.dq"]
    np.testing.assert_allclose(dq, ref, rtol=0, atol=2e-5 * np.abs(ref).max())


# ---------------------------------------------------------------- G7 aggregate_attention
def g7_store(meta):
    store = oattn.OracleStore()
    store.num_att_layers = len(meta["layout"])
    for li, (place, is_cross, N) in enumerate(meta["layout"]):
        K = 77 if is_cross else N
        P = torch.from_numpy(hashrand.uniform((meta["batch"] * meta["heads"], N, K), meta["seed_base"] + li))
        store(P, is_cross, place)
    return store


@pytest.mark.parametrize("meta", load_json("g7_aggregate.json"), ids=lambda m: m["name"])
def test_aggregate(meta):
    g = load_npz("g7_aggregate.npz")
    store = g7_store(meta)
    assert {k: len(v) for k, v in store.attention_store.items()} == meta["store_keys"]
    for key in g.files:
        name, tag = key.split(".")
        if name != meta["name"]:
            continue
        _, r, cs, where = tag.split("_")
        A = oattn.aggregate(store.attention_store, int(r[1:]), where.split("-"), cs == "c")
        np.testing.assert_allclose(A.numpy(), g[key], rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------- G8 latent update
def test_update_latent():
    g = load_npz("g8_update_latent.npz")
    lat = torch.from_numpy(g["latents"]).requires_grad_(True)
    loss = (torch.sin(lat) * torch.from_numpy(g["w"])).sum().reshape(1) * 0.01
    (grad,) = torch.autograd.grad(loss, [lat])
    np.testing.assert_allclose(grad.numpy(), g["grad"], rtol=1e-6, atol=1e-9)
    new = lat.detach() - float(g["step"]) * grad  # pipeline_guided_attention.py:466-469
    np.testing.assert_allclose(new.numpy(), g["out"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["base_bos", "base_bos_eot", "coor_bos", "mixed_sharp", "hyper2_flat", "strict_bos",
                                  "strict_mixed_flat"])
def test_loss_reference_loop_form_is_bit_exact(name):
    """The pixel-loop form (what bench.py times as 'reference-style loop loss', and the second checker of the strict
    mode) executes the reference's own fp32 operation sequence: values and autograd gradient equal the fixture's."""
    g = load_npz("g4_loss.npz")
    meta = next(m for m in G4_META if m["name"] == name)
    plan = oloss.TokenPlan.from_golden(meta)
    A = torch.from_numpy(g[f"{name}.A"]).requires_grad_(True)
    r = oloss.loss_reference_loops(A * 1.0, plan, smooth=meta["smooth"], sigma=meta["sigma"],
                                   kernel_size=meta["kernel_size"], normalize_eot=meta["normalize_eot"],
                                   n_prompt_tokens=meta["n_prompt_tokens"])
    assert float(r["loss"]) == float(g[f"{name}.loss"])
    np.testing.assert_array_equal(np.array([float(v) for v in r["inside"]], np.float32), g[f"{name}.inside"])
    np.testing.assert_array_equal(np.array([float(v) for v in r["outside"]], np.float32), g[f"{name}.outside"])
    (dA,) = torch.autograd.grad(r["loss"], [A])
    np.testing.assert_array_equal(dA.numpy(), g[f"{name}.dA"])
