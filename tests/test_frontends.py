"""§8 f4 — front-ends and diagnostics (off the hot path): the Flask contract of gui.py, the notebook helpers, image
annotation, latent statistics; on the GPU the opt-in PNG / log side effects of the sampling loop and
`vis_utils.show_cross_attention`."""
import json
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_gui_rejects_a_second_generation_while_one_runs(tmp_path, monkeypatch):
    """The captured hipGraphs replay on static buffers: overlapping POSTs must not run concurrently (409 busy)."""
    import threading
    pytest.importorskip("flask")
    from guided_attention_amd import gui
    from guided_attention_amd.config import RunConfig
    from guided_attention_amd.utils import shared_state as state
    state.config = RunConfig(meta_prompt="x", output_path=str(tmp_path))
    started, release, calls = threading.Event(), threading.Event(), []
    png = tmp_path / "out.png"
    png.write_bytes(b"\x89PNG")

    def slow_execute(config):
        calls.append(config.meta_prompt)
        started.set()
        assert release.wait(30)
        return png

    monkeypatch.setattr(gui, "_execute", slow_execute)
    first = {}
    t = threading.Thread(target=lambda: first.update(r=gui.app.test_client().post(
        "/execute_function", data=json.dumps({"variable1": "a [cat:.1,.1,.5,.5]"}), content_type="application/json")))
    t.start()
    assert started.wait(30)
    second = gui.app.test_client().post("/execute_function", data=json.dumps({"variable1": "a [dog:.1,.1,.5,.5]"}),
                                        content_type="application/json")
    assert second.status_code == 409 and "busy" in second.get_json()["error"]
    release.set()
    t.join(30)
    assert first["r"].status_code == 200 and calls == ["a [cat:.1,.1,.5,.5]"]
    third = gui.app.test_client().post("/execute_function", data=json.dumps({"variable1": "a [dog:.1,.1,.5,.5]"}),
                                       content_type="application/json")
    assert third.status_code == 200 and len(calls) == 2


def test_gui_contract(tmp_path, monkeypatch):
    """POST /execute_function {variable1} -> config.meta_prompt set, ONE random seed, execute() called, the PNG copied
    to static/output.png, {"result": path} returned (reference gui.py:25-38); GET / serves the page."""
    from PIL import Image
    from guided_attention_amd import gui
    from guided_attention_amd.config import RunConfig
    from guided_attention_amd.utils import shared_state as state
    state.config = RunConfig(meta_prompt="old", output_path=tmp_path)
    seen = {}

    def fake_execute(config):
        seen["meta_prompt"], seen["seeds"] = config.meta_prompt, list(config.seeds)
        path = tmp_path / "img.png"
        Image.new("RGB", (8, 8), (1, 2, 3)).save(path)
        return path

    monkeypatch.setattr(gui, "_execute", fake_execute)
    client = gui.app.test_client()
    page = client.get("/")
    assert page.status_code == 200 and b"execute_function" in page.data and page.headers["Cache-Control"] == "no-store"
    mp = "a [cat:.2,.5] and a [vase:.7,.5]"
    r = client.post("/execute_function", data=json.dumps({"variable1": mp}), content_type="application/json")
    assert r.status_code == 200 and r.get_json() == {"result": str(tmp_path / "img.png")}
    assert seen["meta_prompt"] == mp and len(seen["seeds"]) == 1 and 0 <= seen["seeds"][0] < 4294967294
    assert (Path(gui.HERE) / "static" / "output.png").exists()
    assert client.post("/post", data={"a": "b"}).data.startswith(b"recived: ")


def test_view_images_and_text_under_image():
    from guided_attention_amd.utils import ptp_utils
    img = np.full((50, 40, 3), 7, np.uint8)
    cap = ptp_utils.text_under_image(img, "robot")
    assert cap.shape == (60, 40, 3) and (cap[:50] == 7).all() and (cap[50:] != 255).any()   # text was drawn in the strip
    grid = ptp_utils.view_images(np.stack([cap, cap, cap]), display_image=False)
    assert grid.size == (3 * 40 + 2 * int(60 * 0.02), 60)
    grid2 = ptp_utils.view_images([cap, cap, cap], num_rows=2, display_image=False)   # padded with one empty panel
    assert grid2.size[1] == 2 * 60 + int(60 * 0.02)


def test_annotate_image_and_latent_stats(tmp_path):
    from PIL import Image
    from guided_attention_amd.config import RunConfig
    from guided_attention_amd.utils import helpers, shared_state as state
    cfg = RunConfig(meta_prompt="a [robot:.5,.25,.25,.5] and a [fox:.25,.75]", output_path=tmp_path, annotate=True)
    state.config = SimpleNamespace(registered_loss_functions={})
    cfg.prompt, cfg.meta_info, cfg.custom_loss = helpers.parse_prompt(cfg.meta_prompt)
    state.config = cfg
    im = Image.new("RGB", (64, 64), (255, 255, 255))
    helpers.annotate_image(im)
    a = np.asarray(im)
    assert (a[16:48, 32] != 255).any() and (a[16, 32:48] != 255).any()      # box edges at x = .5, y = .25
    assert (a[48, 10:22] != 255).any()                                       # the cross of the COOR token at (.25, .75)
    cfg.annotate = False
    im2 = Image.new("RGB", (64, 64), (255, 255, 255))
    helpers.annotate_image(im2)
    assert (np.asarray(im2) == 255).all()
    # latent statistics (reference helpers.py:313-349)
    helpers.means, helpers.stds, helpers.percentile99 = {}, {}, {}
    lat = torch.arange(4 * 4 * 4, dtype=torch.float32).reshape(1, 4, 4, 4) / 10
    helpers.log_latent_stats(lat, True)
    helpers.log_latent_stats(lat * 2, True)
    assert sorted(helpers.means) == ["ch0", "ch1", "ch2", "ch3"] and len(helpers.means["ch2"]) == 2
    np.testing.assert_allclose(helpers.percentile99["ch0"][0], np.quantile(lat[0, 0].abs().numpy(), .99), rtol=1e-6)
    np.testing.assert_allclose(helpers.stds["ch1"][1], (lat[0, 1] * 2).abs().std().item(), rtol=1e-6)
    cfg.diagnostic_level = 1
    helpers.save_latent_stats(tmp_path / "fig.png")
    assert (tmp_path / "fig.png").exists() and helpers.means == {}


@pytest.mark.gpu
def test_reference_side_effects_and_show_cross_attention(tmp_path):
    """Opt-in reproduction of the reference's PNG / log side effects (pipeline_guided_attention.py:243-246, 1031-1037):
    per-token map PNGs for every loss evaluation, predicted-x0 PNGs for steps 0-2, latent statistics, the map sums in the
    log — and the SAME latents as the default (silent, graph-replayed) run.  Then show_cross_attention on the store."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from guided_attention_amd import run
    from guided_attention_amd.config import RunConfig
    from guided_attention_amd.pipeline_guided_attention import GuidedAttention
    from guided_attention_amd.unet import UNetConfig
    from guided_attention_amd.utils import helpers, ptp_utils, shared_state as state, vis_utils
    pipe = GuidedAttention.from_pretrained("random", random_init=True, unet_config=UNetConfig.tiny(32, 48), seed=5)
    pipe.to("cuda", torch.float32)
    pipe.use_graphs = True
    cfg = RunConfig(meta_prompt="a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]", seeds=[11], n_inference_steps=4,
                    output_path=tmp_path / "quiet")
    cfg.stable = pipe
    state.config = cfg
    state.hyperParameterIterations = [{}]
    run.execute(cfg)
    quiet = state.last_results["latents"][0].clone()
    assert not list((tmp_path / "quiet").rglob("_attnmap_*"))           # default: no diagnostic files at all
    cfg.output_path = tmp_path / "loud"
    pipe.reference_side_effects = True
    helpers.means, helpers.stds, helpers.percentile99 = {}, {}, {}
    run.execute(cfg, save=False)
    pipe.reference_side_effects = False
    loud = state.last_results["latents"][0]
    assert (quiet - loud).abs().max().item() < 2e-4 * quiet.abs().max().item()
    folder = tmp_path / "loud" / helpers.get_inner_folder_name() / "11"
    maps = sorted(p.name for p in folder.glob("_attnmap_*"))
    assert any(n.startswith("_attnmap_robot_") for n in maps) and any(n.startswith("_attnmap_vase_") for n in maps)
    assert any("_subiter_01" in n for n in maps)                         # refinement sub-iterations dump their own maps
    preds = sorted(p.name for p in folder.glob("*_pred.png"))
    assert len(preds) == 3 and "cur_time_step_iter_00_" in preds[0] and "cur_time_step_iter_02_" in preds[2]
    assert len(helpers.means["ch0"]) >= 4                                # one entry per DDIM step (plus recurse passes)
    # show_cross_attention: one captioned heat-map panel per token to alter
    from PIL import Image
    ctrl = ptp_utils.AttentionStore()
    ptp_utils.register_attention_control(pipe, ctrl)
    with torch.no_grad():
        emb = torch.randn(1, 77, 48, device="cuda")
        pipe.unet(torch.randn(1, 4, 32, 32, device="cuda"), 981, encoder_hidden_states=emb)
    grid = vis_utils.show_cross_attention("a robot and a blue vase", ctrl, pipe.tokenizer, [2, 6], 16, ("up", "down", "mid"),
                                          orig_image=Image.new("RGB", (64, 64), (90, 120, 200)), display_image=False)
    assert grid.size == (2 * 256 + int(307 * 0.02), 307)                 # 2 panels of 256 px + caption strips
