"""The N-rank plumbing without a GPU:
  * `python bench.py --gpus 2 --launch-check` — the command form the driver uses, run directly: the parent spawns the
    two ranks itself (gloo on this CPU-only box), they rendezvous on 127.0.0.1 and rank 0 prints the line;
  * `bench.rank_envs` builds what torch.distributed.run would have put in the environment;
  * `run.execute` stripes its (seed, hyper-parameter state) jobs over the ranks and rank 0 gets the results back in
    job order (world size 2, gloo; the generation itself is a stand-in — the pipeline has no CPU path);
  * the weight broadcast bumps parameter versions (weight-derived caches are keyed on them)."""
import json
import os
import subprocess
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def test_bench_gpus_2_self_launches_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE line: rank 0's; the other rank's stdout is dropped
    assert lines[0] == {"launch_check": True, "world": 2, "backend": "gloo", "sum": 3.0, "self_launched": True}


def test_bench_gpus_8_launch_path_rendezvouses_eight_ranks():
    """The form the driver's scaling run takes (`python bench.py --gpus 8 ...`), exercised without a node: eight gloo
    ranks rendezvous on 127.0.0.1, all-reduce rank + 1 and rank 0 prints the one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines == [{"launch_check": True, "world": 8, "backend": "gloo", "sum": 36.0, "self_launched": True}]


def test_launcher_deadline_terminates_hung_ranks(tmp_path, monkeypatch):
    """Ranks that never finish (stuck in a collective) are terminated at the deadline; the launcher returns non-zero and
    leaves no child behind."""
    sys.path.insert(0, str(ROOT))
    import bench
    script = tmp_path / "hang.py"
    script.write_text("import time, os\nopen(os.environ['GA_PIDFILE'] + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setenv("GA_PIDFILE", str(tmp_path / "pid"))
    rc = bench.launch_ranks(2, [], deadline_s=3.0)
    assert rc == 124
    for r in range(2):
        pid = int((tmp_path / f"pid{r}").read_text())
        try:
            os.kill(pid, 0)
            alive = True
        except OSError:
            alive = False
        assert not alive, f"rank {r} (pid {pid}) survived the launcher"


def test_bench_under_an_external_launcher_does_not_spawn():
    """With RANK in the environment (torch.distributed.run) bench.py is a rank, not a launcher."""
    sys.path.insert(0, str(ROOT))
    import bench
    port = bench.free_port()
    envs = bench.rank_envs(1, port, base={k: v for k, v in os.environ.items()})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--launch-check"], env=envs[0],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["world"] == 1


def test_rank_envs():
    sys.path.insert(0, str(ROOT))
    import bench
    envs = bench.rank_envs(4, 12345, base={"PATH": "/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "12345" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin" for e in envs)
    args = bench.parse(["--gpus", "8", "--steps", "3", "--warmup", "1"])
    assert (args.gpus, args.steps, args.warmup, args.model) == (8, 3, 1, "sd15")


def test_failed_rank_fails_the_launch(tmp_path):
    """A rank that dies must not leave the launcher waiting on the others: worst return code is reported."""
    sys.path.insert(0, str(ROOT))
    import bench
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    saved = bench.__file__
    bench.__file__ = str(script)
    try:
        assert bench.launch_ranks(2, []) == 7
    finally:
        bench.__file__ = saved


# ----------------------------------------------------------------------------------------- run.execute striping
def _fake_generation(seed_value, hp):
    from PIL import Image
    lat = torch.full((1, 4, 8, 8), float(seed_value) + 0.25 * hp)
    img = Image.fromarray(np.full((16, 16, 3), (seed_value * 7 + hp) % 251, np.uint8))
    return SimpleNamespace(images=[img], latents=lat)


def _execute_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from guided_attention_amd import parallel, run
    from guided_attention_amd.config import RunConfig
    from guided_attention_amd.text import WordTokenizer
    from guided_attention_amd.utils import shared_state as state
    if world > 1:
        parallel.init_distributed("gloo")
    cfg = RunConfig(meta_prompt="a [robot:.6,.3,.4,.55] and a [blue vase:.2,.3,.4,.55]", seeds=[3, 1, 4, 1, 5],
                    output_path=Path(out_dir))
    cfg.stable = SimpleNamespace(device=torch.device("cpu"), tokenizer=WordTokenizer())
    state.hyperParameterIterations = [{}, {"inside_loss_scale": .3}]   # two states per seed -> 10 jobs
    seen = []

    def fake_run_on_prompt(prompt, model, controller, seed, config, **extra):
        hp = 0 if state.curHyperParams["inside_loss_scale"] == .2 else 1
        seen.append((state.cur_seed, hp))
        assert seed.initial_seed() == state.cur_seed and extra == {"output_type": "pil"}
        return _fake_generation(state.cur_seed, hp)

    run.run_on_prompt = fake_run_on_prompt
    try:
        last = run.execute(cfg)
    finally:
        state.hyperParameterIterations = [{}]
    jobs = [(s, h) for s in cfg.seeds for h in (0, 1)]
    assert seen == jobs[rank::world]                                 # this rank generated exactly its stripe
    folder = Path(out_dir) / "a _robot__6,_3,_4,_55_ and a _blue vase__2,_3,_4,_55_"
    assert last.parent == folder and last.name.startswith("5_")      # path of the LAST job, on every rank
    if rank == 0:
        res = state.last_results
        assert [float(t[0, 0, 0, 0]) for t in res["latents"]] == [s + 0.25 * h for s, h in jobs]   # job order
        assert [int(np.asarray(im)[0, 0, 0]) for im in res["images"]] == [(s * 7 + h) % 251 for s, h in jobs]
        assert (Path(out_dir) / "a _robot__6,_3,_4,_55_ and a _blue vase__2,_3,_4,_55_.png").exists()  # the grid
    else:
        assert state.last_results is None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_execute_stripes_jobs_over_two_ranks(tmp_path):
    port = 29900 + os.getpid() % 90
    mp.spawn(_execute_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    folder = tmp_path / "a _robot__6,_3,_4,_55_ and a _blue vase__2,_3,_4,_55_"
    pngs = sorted(p.name for p in folder.glob("*.png"))
    assert len(pngs) == 8, pngs      # 5 seeds x 2 states, seed 1 listed twice -> 8 distinct files, written by both ranks


def test_execute_single_process_is_the_serial_loop(tmp_path):
    _execute_worker(0, 1, 0, str(tmp_path))


# ----------------------------------------------------------------------------------------- broadcast bumps versions
def _bcast_worker(rank, world, port):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from guided_attention_amd import parallel
    parallel.init_distributed("gloo")
    lin = torch.nn.Linear(8, 8)
    for p in lin.parameters():
        p.requires_grad_(False)
    before = [p._version for p in lin.parameters()]
    parallel.broadcast_module_(lin)
    after = [p._version for p in lin.parameters()]
    if rank != 0:
        assert all(a > b for a, b in zip(after, before)), (before, after)
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_bumps_parameter_versions():
    mp.spawn(_bcast_worker, args=(2, 29800 + os.getpid() % 90), nprocs=2, join=True)
