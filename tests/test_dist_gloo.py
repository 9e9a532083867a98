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


# ---- training exchange (K16 / BASELINE configs[3]): flat gradient buckets + asynchronous all-reduce ----
class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.unused = torch.nn.Linear(5, 5)  # never called: its gradient stays zero, its bucket is launched by finish()
        self.body = torch.nn.Sequential(torch.nn.Linear(6, 50), torch.nn.ReLU(), torch.nn.Linear(50, 70), torch.nn.ReLU(),
                                        torch.nn.Linear(70, 3))

    def forward(self, x):
        return self.body(x)


def _toy_trainable():
    torch.manual_seed(3)
    return _Toy()


def _train_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from diff_sal_amd.train_step import FlatParams, GradReducer

    dsd.init_from_env("gloo")
    m = _toy_trainable()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatParams(m, bucket_bytes=1024)  # several buckets
    red = GradReducer(flat)
    same_values = all(torch.equal(before[k], v) for k, v in m.state_dict().items())
    views = all(p.data_ptr() == flat.flat_p.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
    ok_steps = []
    for it in range(2):  # two steps: arm()/finish() are reusable and zero_grad() really clears
        g = torch.Generator().manual_seed(100 * it + rank)
        x = torch.randn(4, 6, generator=g)
        flat.zero_grad()
        red.arm()
        m(x).square().sum().backward()
        red.finish()
        views = views and all(p.grad is None or p.grad.data_ptr() == flat.flat_g.data_ptr() + 4 * o
                              for p, o in zip(flat.params, flat.offsets))
        # expected: sum over ranks of the single-process gradients
        exp = torch.zeros_like(flat.flat_g)
        for r in range(world):
            m2 = _toy_trainable()
            f2 = FlatParams(m2, bucket_bytes=1024)
            xr = torch.randn(4, 6, generator=torch.Generator().manual_seed(100 * it + r))
            f2.zero_grad()
            m2(xr).square().sum().backward()
            f2.gather(range(len(f2.params)))
            exp += f2.flat_g
        ok_steps.append(bool(torch.allclose(flat.flat_g, exp, rtol=1e-6, atol=1e-6)))
    ret[rank] = dict(same_values=same_values, views=views, ok=ok_steps, nb=len(flat.buckets), order=list(red.launch_order),
                     cover=sum(len(b) for b in flat.buckets) == flat.numel,
                     first_is_last_layer=flat.params[0] is list(m.parameters())[-1])
    torch.distributed.destroy_process_group()


def test_two_rank_bucketed_gradient_allreduce():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_train_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r]["same_values"] and ret[r]["views"] and ret[r]["cover"] and ret[r]["first_is_last_layer"]
        assert ret[r]["ok"] == [True, True]
        assert ret[r]["nb"] >= 3
        assert sorted(ret[r]["order"]) == list(range(ret[r]["nb"]))   # every bucket exchanged exactly once
    assert ret[0]["order"] == ret[1]["order"]                          # same collective order on every rank


# ---- wider worlds, uneven buckets, rank-dependent graphs, both exchange modes ----
class _Branchy(torch.nn.Module):
    """Uneven parameter sizes (buckets of 1 .. 2 tensors at 512 B), and a branch that only some ranks execute."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(7, 300)        # 2100 + 300 parameters: a bucket of its own
        self.side = torch.nn.Linear(300, 300)   # used on even ranks only
        self.b = torch.nn.Linear(300, 5)
        self.tiny = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(3)) for _ in range(6)])

    def forward(self, x, use_side):
        h = torch.tanh(self.a(x))
        if use_side:
            h = h + torch.tanh(self.side(h))
        y = self.b(h)
        return y + sum(t.sum() for t in self.tiny)


def _branchy_worker(rank, world, port, mode, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from diff_sal_amd.train_step import FlatParams, GradReducer

    dsd.init_from_env("gloo")
    torch.manual_seed(11)
    m = _Branchy()
    flat = FlatParams(m, bucket_bytes=512)
    red = GradReducer(flat, exchange=mode)
    oks, orders = [], []
    for it in range(2):
        flat.zero_grad()
        red.arm()
        x = torch.randn(3, 7, generator=torch.Generator().manual_seed(50 * it + rank))
        m(x, use_side=(rank % 2 == 0)).square().sum().backward()       # `side` gets no gradient on odd ranks
        red.finish()
        exp = torch.zeros_like(flat.flat_g)
        for r in range(world):
            torch.manual_seed(11)
            m2 = _Branchy()
            f2 = FlatParams(m2, bucket_bytes=512)
            xr = torch.randn(3, 7, generator=torch.Generator().manual_seed(50 * it + r))
            f2.zero_grad()
            m2(xr, use_side=(r % 2 == 0)).square().sum().backward()
            f2.gather(range(len(f2.params)))
            exp += f2.flat_g
        oks.append(bool(torch.allclose(flat.flat_g, exp, rtol=1e-5, atol=1e-5)))
        orders.append(list(red.launch_order))
    ret[rank] = dict(ok=oks, orders=orders, nb=len(flat.buckets), sizes=[len(b) for b in flat.buckets],
                     members=[len(mm) for mm in flat.bucket_members], kinds=list(red.collectives))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(4, "allreduce"), (4, "reduce_scatter"), (8, "allreduce")])
def test_many_rank_exchange_with_rank_dependent_graph(world, mode):
    """4 and 8 gloo ranks; buckets of unequal membership; a layer that receives no gradient on the odd ranks (its bucket closes
    in finish() there and during backward on the even ranks).  Every rank must issue the collectives in the same order --
    bucket order -- or the exchange pairs different buffers; the summed gradient equals the sum of the per-rank gradients."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_branchy_worker, args=(world, port, mode, ret), nprocs=world, join=True)
    nb = ret[0]["nb"]
    assert nb >= 4 and len(set(ret[0]["members"])) >= 2                 # uneven membership
    for r in range(world):
        assert ret[r]["ok"] == [True, True], (r, ret[r])
        assert ret[r]["orders"] == [list(range(nb))] * 2                 # strictly bucket order, on every rank, every step
        assert len(ret[r]["kinds"]) == nb
    if mode == "reduce_scatter":
        # buckets whose length divides by the world take the reduce-scatter + all-gather path, the others the all-reduce
        kinds = ret[0]["kinds"]
        for k, n in zip(kinds, ret[0]["sizes"]):
            assert k == ("reduce_scatter+all_gather" if n % world == 0 else "allreduce")
        assert "reduce_scatter+all_gather" in kinds


# ---- parameters that never receive a gradient must not serialise the exchange behind backward ----
def test_unused_parameters_are_learnt_in_the_first_step_and_no_longer_hold_buckets_open():
    from diff_sal_amd.train_step import FlatParams, GradReducer

    m = _toy_trainable()
    flat = FlatParams(m, bucket_bytes=1024)
    red = GradReducer(flat)
    nb = len(flat.buckets)
    launched_before_finish = []
    for it in range(3):
        flat.zero_grad()
        red.arm()
        m(torch.randn(4, 6, generator=torch.Generator().manual_seed(it))).square().sum().backward()
        launched_before_finish.append(len(red.launch_order))
        red.finish()
        assert red.launch_order == list(range(nb))                   # bucket order, every bucket once
    unused = {i for i, p in enumerate(flat.params) if any(p is q for q in m.unused.parameters())}
    assert red._unused == unused and len(unused) == 2
    # step 1: the bucket holding `unused` (and every later one) waits for finish(); afterwards nothing does
    assert launched_before_finish[0] < nb
    assert launched_before_finish[1] == nb and launched_before_finish[2] == nb


def test_a_gradient_for_a_parameter_learnt_as_unused_is_picked_up_and_the_set_relearnt():
    """R/model.py:15 is DDP(find_unused_parameters=True): the set of gradient-less parameters may change.  The prediction
    learnt in step 1 is verified every step; a parameter it got wrong is still exchanged and leaves the set."""
    from diff_sal_amd.train_step import FlatParams, GradReducer

    torch.manual_seed(11)
    m = _Branchy()
    flat = FlatParams(m, bucket_bytes=512)
    red = GradReducer(flat)
    x = torch.randn(3, 7)

    def one(use_side):
        flat.zero_grad()
        red.arm()
        m(x, use_side=use_side).square().sum().backward()
        red.finish()

    one(False)                                   # `side` learnt as unused
    assert len(red._unused) == 2
    one(True)                                    # its bucket went out before its gradient arrived: repaired by finish()
    assert len(red.relearned) == 2 and red._unused == set()
    idx = [i for i, p in enumerate(flat.params) if p is m.side.weight][0]
    o = flat.offsets[idx]
    ref = torch.autograd.grad(m(x, use_side=True).square().sum(), m.side.weight)[0]
    assert torch.allclose(flat.flat_g[o:o + ref.numel()].view(ref.shape), ref, rtol=1e-6, atol=1e-7)
    assert m.side.weight.grad.data_ptr() == flat.flat_g.data_ptr() + 4 * o
    one(True)                                    # now an ordinary member of its bucket
    assert red.relearned == []
    assert torch.allclose(flat.flat_g[o:o + ref.numel()].view(ref.shape), ref, rtol=1e-6, atol=1e-7)
    red.reset_unused()
    one(False)
    assert len(red._unused) == 2                 # re-learnt from scratch
    red.static_unused = False
    red.reset_unused()
    one(False)
    assert red._unused is None                   # prediction off: nothing learnt


def _late_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from diff_sal_amd.train_step import FlatParams, GradReducer

    dsd.init_from_env("gloo")
    torch.manual_seed(11)
    m = _Branchy()
    flat = FlatParams(m, bucket_bytes=512)
    red = GradReducer(flat)
    # step 0: no rank uses `side` (learnt as unused everywhere); step 1: ONLY rank 1 uses it; step 2: both; step 3: nobody
    plan = [(False, False), (False, True), (True, True), (False, False)]
    oks, relearned, unused = [], [], []
    for it, uses in enumerate(plan):
        flat.zero_grad()
        red.arm()
        x = torch.randn(3, 7, generator=torch.Generator().manual_seed(50 * it + rank))
        m(x, use_side=uses[rank]).square().sum().backward()
        red.finish()
        exp = torch.zeros_like(flat.flat_g)
        for r in range(world):
            torch.manual_seed(11)
            m2 = _Branchy()
            f2 = FlatParams(m2, bucket_bytes=512)
            xr = torch.randn(3, 7, generator=torch.Generator().manual_seed(50 * it + r))
            f2.zero_grad()
            m2(xr, use_side=uses[r]).square().sum().backward()
            f2.gather(range(len(f2.params)))
            exp += f2.flat_g
        oks.append(bool(torch.allclose(flat.flat_g, exp, rtol=1e-5, atol=1e-5)))
        relearned.append(len(red.relearned))
        unused.append(len(red._unused))
    ret[rank] = dict(ok=oks, relearned=relearned, unused=unused)
    torch.distributed.destroy_process_group()


def test_two_ranks_a_gradient_that_appears_later_on_one_rank_only_reaches_every_replica():
    """The advisor's case: a parameter learnt as gradient-less gets a gradient later, on ONE rank.  Every rank must end the step
    with the same, complete sum (no local raise, no hang, no replica divergence)."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_late_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r]["ok"] == [True] * 4, ret[r]
        assert ret[r]["relearned"] == [0, 2, 0, 0]        # found on both ranks together, in the step it happened
        assert ret[r]["unused"] == [2, 0, 0, 0]


def test_train_step_goes_through_the_optimizer_face_and_keeps_scheduler_keys():
    """MultiStepLR must see optimizer.step() before scheduler.step() (no warning), and its `initial_lr` survives a checkpoint."""
    import warnings

    from diff_sal_amd.train_step import FlatAdam

    class _Owner:
        def __init__(self):
            self.n = 0

        def optimizer_step(self):
            self.n += 1

    own = _Owner()
    p = torch.nn.Parameter(torch.zeros(3))
    opt = FlatAdam(own, [p], 1e-4, (0.9, 0.999), 1e-8, 0.0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        opt.step()
        sched.step()
    assert own.n == 1 and abs(opt.param_groups[0]["lr"] - 1e-5) < 1e-12 and opt.param_groups[0]["initial_lr"] == 1e-4


def test_batchnorm_buffers_become_views_of_one_flat_tensor_for_the_broadcast():
    """DDP's broadcast_buffers (R/model.py:15) as ONE collective: the floating-point buffers are views of a flat tensor (no gather,
    no per-buffer copy back); in-place running-statistics updates go through the views; a buffer replaced behind the step's back
    (module.to(...), re-registration) is noticed and the flat tensor rebuilt."""
    import torch.distributed as dist

    from diff_sal_amd.train_step import DiffusionTrainStep

    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
        created = True
    try:
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1), torch.nn.BatchNorm2d(2))
        ts = DiffusionTrainStep(m, exchange_single_rank=True)
        assert ts.broadcast_buffers
        m[1].running_mean.fill_(0.25)
        ts._sync_buffers()
        flat = ts._flat_buffers
        floats = [b for b in m.buffers() if b.dtype == torch.float32]
        assert flat is not None and len(floats) == 4
        lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
        assert all(lo <= b.data_ptr() < hi for b in floats)                         # views, not copies
        assert torch.all(m[1].running_mean == 0.25)                                  # values survived the move
        m[3].running_var.mul_(3.0)                                                   # a kernel's in-place update ...
        assert float(flat.sum()) == float(sum(b.sum() for b in floats))              # ... is seen by the flat tensor
        m[1].running_mean = torch.full((4,), 0.5)                                    # replaced behind the step's back
        ts._sync_buffers()
        assert ts._flat_buffers is not flat                                          # rebuilt
        assert m[1].running_mean.data_ptr() >= ts._flat_buffers.data_ptr() and torch.all(m[1].running_mean == 0.5)
    finally:
        if created:
            dist.destroy_process_group()
