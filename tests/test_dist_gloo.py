"""N>1 path on CPU: two gloo ranks shard clips, sample them with the host-side sampler logic, gather."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from diff_sal_amd import dist as dsd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy_net(x, t, img, a):
    return torch.sigmoid(0.8 * x + 0.002 * t.float().view(-1, 1, 1, 1) + img[0].mean(dim=(1, 2), keepdim=False)[:, None])


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from diff_sal_amd.sampling import DiffusionSampler

    r, lr, w = dsd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(7)
    n = 5  # uneven split: rank 0 gets 3 clips, rank 1 gets 2
    x = torch.randn(n, 1, 6, 8, generator=g)
    feats = [torch.randn(n, 4, 2, 6, 8, generator=g)]
    model = type("M", (), {"decoder_net": staticmethod(_toy_net)})()
    s = DiffusionSampler(model, timesteps=6, sample_type="dpmsolver", skip_type="time_uniform")
    full = dsd.sample_sharded(s, x, feats, None, batch=2, gather=True)
    local = dsd.sample_sharded(s, x, feats, None, batch=2, gather=False)
    ref = s.sample_dpm_solver(x, feats, None)  # unsharded
    dsd.barrier()
    t = dsd.max_over_ranks(1.0 + rank)
    ret[rank] = (torch.allclose(full, ref, atol=1e-6), tuple(local.shape), t, list(dsd.shard_range(n, rank, world)))
    torch.distributed.destroy_process_group()


def test_two_rank_clip_sharding_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0][0] and ret[1][0]                      # gathered result == unsharded result, in clip order
    assert ret[0][1][0] == 3 and ret[1][1][0] == 2      # balanced contiguous shards
    assert ret[0][2] == 2.0 and ret[1][2] == 2.0        # MAX over ranks
    assert ret[0][3] + ret[1][3] == [0, 1, 2, 3, 4]     # disjoint cover


@pytest.mark.parametrize("n,world", [(0, 2), (1, 4), (8, 8), (64, 8), (7, 3)])
def test_shard_range_covers_everything_once(n, world):
    seen = []
    for r in range(world):
        seen += list(dsd.shard_range(n, r, world))
    assert seen == list(range(n))
    sizes = [len(dsd.shard_range(n, r, world)) for r in range(world)]
    assert max(sizes) - min(sizes) <= 1
