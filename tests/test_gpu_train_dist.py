"""Two training ranks sharing the one GPU of the test box (gloo carries the CUDA buffers): the whole N>1 training path --
bucket gather kernel, hook-driven asynchronous all-reduce, 1/world folded into norm + Adam, buffer broadcast -- against
a single-process computation of the same mean gradient.  (RCCL refuses two ranks on one device; the collective
semantics are the same.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank_tag):
    from oracle import salunet_oracle as orc
    from tests._cases import CASES
    from tests.test_gpu_salunet import build

    cfg = CASES["tiny_av"][0]
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    net = build(cfg, sd)
    net.dropout_p = 0.0
    x, feats, audio = orc.synth_inputs(cfg, 2, True, tag=f"dist{rank_tag}")
    sal = torch.sigmoid(orc.synth_tensor(f"dist{rank_tag}.sal", (2, 1, *cfg.img_size)))
    dq = orc.synth_tensor(f"dist{rank_tag}.dq", tuple(sal.shape))
    dev = "cuda"
    cond = {"feat_list": [f.to(dev) for f in feats], "audio_feat": audio.to(dev)}
    return net, sal.to(dev), dq.to(dev), x.to(dev), cond


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from diff_sal_amd.train_step import DiffusionTrainStep

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, sal, dq, noise, cond = _setup(rank)
    ts = DiffusionTrainStep(net, lr=1e-4, grad_clip=1.0, bucket_mb=0.25)
    assert ts.world == 2 and len(ts.flat.buckets) >= 3
    losses = []
    for it in range(2):
        losses.append(ts.step(sal, cond, t0=200 + 300 * it, noise=noise, dequant_noise=dq).item())
    ret[rank] = dict(p=ts.flat.flat_p.cpu(), losses=losses, order=list(ts.reducer.launch_order),
                     rm=net.invpt_decoder.mt_proj[1].running_mean.cpu())
    dist.destroy_process_group()


def test_two_rank_training_matches_mean_gradient_reference():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert torch.equal(ret[0]["p"], ret[1]["p"])                       # replicas stay bit-identical
    assert sorted(ret[0]["order"]) == list(range(len(ret[0]["order"]))) and ret[0]["order"] == ret[1]["order"]

    # single process: each rank's gradient separately (per-rank BatchNorm statistics, like the reference's DDP), the mean
    # of the two through the same norm + Adam kernels
    from diff_sal_amd import ops
    from diff_sal_amd.train_step import DiffusionTrainStep

    nets = [_setup(r) for r in range(world)]
    steps = [DiffusionTrainStep(n[0], lr=1e-4, grad_clip=1.0, bucket_mb=0.25) for n in nets]
    for it in range(2):
        g = None
        for (net, sal, dq, noise, cond), ts in zip(nets, steps):
            x0, x_t, t, _ = ts.prepare_data(sal, t0=200 + 300 * it, noise=noise, dequant_noise=dq)
            ts.loss_and_backward(x0, x_t, t, cond)
            g = ts.flat.flat_g.clone() if g is None else g + ts.flat.flat_g
        for ts in steps:                                                   # both replicas take the same mean-gradient step
            ts.flat.flat_g.copy_(g)
            ts.step_count += 1
            norm = ops.grad_norm(ts.flat.flat_g, 1.0 / world)
            ops.adam_step(ts.flat.flat_p, ts.flat.flat_g, ts.flat.exp_avg, ts.flat.exp_avg_sq, step=ts.step_count, lr=1e-4,
                          gscale=1.0 / world, norm=norm, max_norm=1.0)
            # rank 0's running statistics are broadcast before every forward (DDP broadcast_buffers)
        for b0, b1 in zip(steps[0].model.buffers(), steps[1].model.buffers()):
            b1.copy_(b0)
    ref = steps[0].flat.flat_p.cpu()
    d = (ret[0]["p"] - ref).abs().max().item()
    moved = (ref - _flat_init()).abs().max().item()
    print("max |param diff| %.3e, largest parameter displacement %.3e" % (d, moved))
    assert moved > 1e-4                                                  # the steps did something
    assert d <= 2e-7                                                     # same kernels, same order: equal up to gloo's summation


def _flat_init():
    from diff_sal_amd.train_step import FlatParams

    net = _setup(0)[0]
    return FlatParams(net).flat_p.cpu()
