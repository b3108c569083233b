"""The N>1 path on CPU: world_size-2 gloo processes exercise the seed striping, the bucketed weight
broadcast and the final gather of guided_attention_amd.parallel (no GPU, no kernels: a stand-in
per-seed worker makes each 'image' a deterministic function of (weights, seed))."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from guided_attention_amd import parallel
    from guided_attention_amd.unet import UNet2DConditionModel, UNetConfig
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # ranks start with DIFFERENT weights; only rank 0's must survive
    unet = UNet2DConditionModel(UNetConfig.tiny(32, 48))
    if rank == 0:
        unet.init_weights_(seed=5)
    unet.half()
    unet.conv_in.float()  # two dtypes -> two bucket groups
    n_msgs = parallel.broadcast_module_(unet, bucket_bytes=1 << 20)
    assert n_msgs >= 3
    ref = UNet2DConditionModel(UNetConfig.tiny(32, 48)).init_weights_(seed=5).half()
    ref.conv_in.float()
    for (n1, p1), (n2, p2) in zip(unet.named_parameters(), ref.named_parameters()):
        assert torch.equal(p1, p2), n1

    def generate(seed):  # depends on the broadcast weights and on the seed only
        g = torch.Generator().manual_seed(seed)
        return torch.randn(1, 4, 8, 8, generator=g) * unet.conv_out.weight.float().abs().mean()

    seeds = [3, 1, 4, 1, 5, 9, 2]
    assert parallel.shard_seeds(seeds, rank, world) == seeds[rank::world]
    out = parallel.execute_seeds(generate, seeds, module=unet)
    if rank == 0:
        assert len(out) == len(seeds)
        for s, t in zip(seeds, out):
            assert torch.equal(t, generate(s))
        torch.save(torch.stack(out), tmp)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = 29600 + os.getpid() % 300
    out = tmp_path / "gathered.pt"
    mp.spawn(_worker, args=(2, port, str(out)), nprocs=2, join=True)
    assert torch.load(out).shape == (7, 1, 4, 8, 8)


def test_single_process_degenerates():
    sys.path.insert(0, str(ROOT))
    from guided_attention_amd import parallel
    assert parallel.shard_seeds([1, 2, 3], 0, 1) == [1, 2, 3]
    assert parallel.unstripe([[0, 2, 4], [1, 3]]) == [0, 1, 2, 3, 4]
    out = parallel.execute_seeds(lambda s: torch.full((2,), float(s)), [7, 8])
    assert [t[0].item() for t in out] == [7.0, 8.0]
