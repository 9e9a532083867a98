"""The diffusion training step of the denoiser (BASELINE configs[3]).

What one step does, and the reference lines it mirrors:

    prepare_data   x0 = sal_map + 0.01 randn;  t0 = ONE integer in [0, T) for the whole rank-batch;
                   x_t = sqrt(a_hat[t0]) x0 + sqrt(1 - a_hat[t0]) randn         R/diffusion_trainer.py:106-117, 122-137,
                                                                                R/datasets/__init__.py:8-25
    forward        pred = model(x_t, t, conditioning)  (train-mode BN with per-rank statistics, dropout 0.1)
    loss           mse_weight * sum_{chw}(pred - x0)^2 .mean(batch)              R/models/sal_losses.py:189-192
    backward       hand-written HIP backward of every operator (autograd_ops.py)
    exchange       gradient MEAN over ranks -- the one collective of the path    (DDP, R/model.py:15)
    clip + Adam    clip_grad_norm_(1.0); Adam(lr 1e-4, (0.9, 0.999), eps 1e-8)   R/diffusion_trainer.py:228-235,
                                                                                R/util/utils.py:116-123

MI355X design.  Parameters, gradients and both Adam moments live in four flat fp32 buffers; every
``nn.Parameter`` (and its ``.grad``) is a view into them, so state_dict()/load_state_dict() keep working and
the reference's per-tensor loops collapse into three streaming kernels (csrc/optim.hip) -- norm, then a fused
clip+Adam that reads the norm from device memory (no host sync anywhere in the step).  Parameters are laid out
in REVERSE registration order, so gradients become final roughly front-to-back in the flat buffer during
backward; the buffer is cut into a few large buckets (default 32 MB: xGMI is point-to-point, a ring all-reduce
is per-link bound, so few large messages beat DDP's 25 MB x many) and each bucket's all-reduce (RCCL) is
launched asynchronously from a post-accumulate hook as soon as its last gradient lands, overlapping the rest of
backward.  Only parameters that can receive a gradient are exchanged (the reference buckets 72 M dead ones).
The DDP mean is not a separate pass: 1/world is folded into the norm and Adam kernels.
BatchNorm running statistics follow DDP's ``broadcast_buffers=True``: rank 0's buffers are broadcast (one flat
message) at the start of each step.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor, nn

from .diffusion_utils import get_beta_schedule, to_torch


def _aligned(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class FlatParams:
    """Flat fp32 storage for the trainable parameters of ``module`` (+ grads and optimizer moments).

    Layout: reverse registration order, each tensor padded to 64 elements (256 B) so every view is 16-byte aligned
    for the vectorised kernels.  Padding stays zero in all four buffers (Adam maps 0 -> 0)."""

    def __init__(self, module: nn.Module, bucket_bytes: int = 32 << 20):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("FlatParams: module has no trainable parameters")
        dev, dt = params[0].device, params[0].dtype
        if dt != torch.float32 or any(p.dtype != dt or p.device != dev for p in params):
            raise ValueError("FlatParams: all trainable parameters must be fp32 on one device")
        self.params: List[nn.Parameter] = params[::-1]
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += _aligned(p.numel())
        self.numel = off
        self.live_numel = sum(p.numel() for p in self.params)
        self.flat_p = torch.zeros(off, device=dev)
        self.flat_g = torch.zeros(off, device=dev)
        self.exp_avg = torch.zeros(off, device=dev)
        self.exp_avg_sq = torch.zeros(off, device=dev)
        for p, o in zip(self.params, self.offsets):
            view = self.flat_p[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = None
        # buckets: contiguous ranges of whole tensors, closed once they reach bucket_bytes
        self.buckets: List[range] = []       # element ranges of the flat buffer
        self.bucket_of: List[int] = []       # parameter index -> bucket
        start, cap = 0, max(1, bucket_bytes // 4)
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            self.bucket_of.append(len(self.buckets))
            end = o + _aligned(p.numel())
            if end - start >= cap or i == len(self.params) - 1:
                self.buckets.append(range(start, end))
                start = end
        self.bucket_size = [0] * len(self.buckets)
        self.bucket_members: List[List[int]] = [[] for _ in self.buckets]
        for i, b in enumerate(self.bucket_of):
            self.bucket_size[b] += 1
            self.bucket_members[b].append(i)

    def grad_view(self, i: int) -> Tensor:
        p, o = self.params[i], self.offsets[i]
        return self.flat_g[o:o + p.numel()].view(p.shape)

    def zero_grad(self) -> None:
        """Clear the flat gradient and detach the parameters from it: with ``p.grad is None`` autograd hands over each
        freshly computed gradient tensor without an accumulation kernel; GradReducer gathers them bucket by bucket."""
        self.flat_g.zero_()
        for p in self.params:
            p.grad = None

    def gather(self, indices: Sequence[int]) -> None:
        """Copy the gradients autograd left on ``params[indices]`` into their flat slots (one multi-tensor kernel) and
        re-point ``.grad`` at the flat views.  Parameters without a gradient keep a zero slot."""
        todo = [i for i in indices if self.params[i].grad is not None
                and self.params[i].grad.data_ptr() != self.flat_g.data_ptr() + 4 * self.offsets[i]]
        if not todo:
            return
        if self.flat_g.is_cuda:
            from . import ops

            ops.multi_copy([self.params[i].grad for i in todo], [self.offsets[i] for i in todo], self.flat_g)
        else:  # host tensors: only the gloo tests of the bucket logic come here
            for i in todo:
                self.grad_view(i).copy_(self.params[i].grad)
        for i in todo:
            self.params[i].grad = self.grad_view(i)


class GradReducer:
    """Bucketed gradient gather + asynchronous SUM over the ranks of ``group`` (the mean's 1/world is applied by the
    consumer).

    ``flat.zero_grad()`` then ``arm()`` before backward; when the last gradient of a bucket has been produced its
    per-parameter hook gathers the bucket into the flat buffer (one kernel) and launches the bucket's exchange;
    ``finish()`` handles whatever is left (buckets holding parameters that received no gradient this step) and
    waits for all of them.

    Collectives are issued strictly in bucket order 0, 1, 2, ... on every rank: a bucket that completes while an earlier
    one is still open waits for it.  Ranks whose graphs differ (a parameter that receives no gradient on ONE rank only --
    its bucket closes in ``finish()`` there, in the middle of backward elsewhere) therefore still pair the same buffers in
    the same order; an order taken from hook arrival would pair different buckets across ranks and hang RCCL.

    ``exchange``: "allreduce" -- one ring all-reduce per bucket; "reduce_scatter" -- reduce-scatter of the bucket into
    per-rank shards followed by an all-gather of the shards (same result; on xGMI's point-to-point links the two halves
    can use all seven links of a GPU where a single ring is bound by one link per hop).  Buckets whose
    length is not a multiple of the world size, and backends without reduce-scatter (gloo), use the all-reduce.

    Parameters without a gradient.  The reference wraps the model in ``DistributedDataParallel(...,
    find_unused_parameters=True)`` (R/model.py:15): the set of parameters that receive no gradient may change from step to
    step and from rank to rank.  Here a bucket holding a parameter that never receives a gradient (audio-branch weights in
    visual-only training, an unused head) would stay open until ``finish()`` -- and, the order being fixed, so would every
    later bucket: the whole exchange serialised behind backward.  ``static_unused`` (default on) PREDICTS the set: the
    parameters that received no gradient on ANY rank in the first step (one small MAX all-reduce of the "fired" flags, issued
    by every rank at the end of its first step) are counted as done by ``arm()``, so their buckets close with the last
    gradient that does arrive.  The prediction is verified every step and a wrong one is repaired, on every rank together:
    a gradient that does arrive for such a parameter (on any rank, before or after its bucket went out) is kept out of the
    bucket; ``finish()`` MAX-reduces the per-parameter "fired" flags of the predicted set over the ranks (host data, a
    gloo side group: no device synchronisation), and the flagged parameters' gradients are summed over the ranks by one
    extra all-reduce each and leave the predicted set (``relearned`` lists them).  The result is the same sum DDP's
    ``find_unused_parameters=True`` produces, whatever the arrival order.  ``reset_unused()`` re-learns the set from
    scratch, ``static_unused = False`` turns the prediction off (every bucket with a gradient-less parameter then waits
    for ``finish()``)."""

    static_unused = True

    def __init__(self, flat: FlatParams, group=None, exchange_single_rank: bool = False, exchange: str = "allreduce"):
        if exchange not in ("allreduce", "reduce_scatter"):
            raise ValueError(f"GradReducer: exchange={exchange!r} (allreduce or reduce_scatter)")
        self.flat, self.group, self.mode = flat, group, exchange
        have_group = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if have_group else 1
        self.rank = dist.get_rank(group) if have_group else 0
        # whether the collective is issued: always with more than one rank; with ONE rank only on request (the call path --
        # hook, bucket gather, asynchronous all_reduce, wait -- then runs unchanged, RCCL reducing over a single rank)
        self.exchange = self.world > 1 or (have_group and bool(exchange_single_rank))
        self._native_rs = have_group and dist.get_backend(group) == "nccl"
        self._pending: List[int] = []
        self._launched: List[bool] = []
        self._ready: List[bool] = []
        self._next = 0
        self._work = []
        self._armed = False
        self._fired: List[bool] = []           # per parameter: its gradient hook ran this step
        self._unused = None                    # None: not learnt yet; else the set of parameters without a gradient on any rank
        self._side = None                      # gloo group for the per-step check of the prediction (host flags)
        self.relearned: List[int] = []         # parameters that left the predicted set in the last step
        self.launch_order: List[int] = []
        self.collectives: List[str] = []       # what was issued per bucket in the last step ("allreduce" / "reduce_scatter+all_gather")
        for i, p in enumerate(flat.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i: int) -> Callable:
        def hook(_param):
            if not self._armed:
                return
            b = self.flat.bucket_of[i]
            self._fired[i] = True
            if self._unused and i in self._unused:      # predicted to stay without a gradient, and got one: finish() repairs
                return
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._ready[b] = True
                self._drain()
        return hook

    def _drain(self) -> None:
        while self._next < len(self._ready) and self._ready[self._next]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b: int) -> None:
        if self._launched[b]:
            return
        self._launched[b] = True
        r = self.flat.buckets[b]
        self.launch_order.append(b)
        members = self.flat.bucket_members[b]
        if self._unused:       # a surprise gradient stays out of the bucket on every rank (finish() exchanges it on its own)
            members = [i for i in members if i not in self._unused]
        self.flat.gather(members)
        if not self.exchange:
            return
        buf = self.flat.flat_g[r.start:r.stop]
        n = buf.numel()
        if self.mode == "reduce_scatter" and n % self.world == 0:
            self.collectives.append("reduce_scatter+all_gather")
            shard = n // self.world
            mine = buf[self.rank * shard:(self.rank + 1) * shard]
            if self._native_rs:      # in place: the output is this rank's slice of the input (RCCL's in-place form)
                self._work.append(dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self._work.append(dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=True))
            else:                    # gloo has no reduce-scatter: same arithmetic through its all-reduce (bookkeeping tests)
                self._work.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.collectives.append("allreduce")
            self._work.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def arm(self) -> None:
        self._pending = list(self.flat.bucket_size)
        self._launched = [False] * len(self.flat.buckets)
        self._ready = [False] * len(self.flat.buckets)
        self._next = 0
        self._work, self.launch_order, self.collectives = [], [], []
        self._fired = [False] * len(self.flat.params)
        self.relearned = []
        if self._unused:
            for i in self._unused:
                b = self.flat.bucket_of[i]
                self._pending[b] -= 1
                if self._pending[b] == 0:
                    self._ready[b] = True      # launched by the first drain (first hook, or finish())
        self._armed = True

    def reset_unused(self) -> None:
        """Forget which parameters were found to receive no gradient; the next step learns the set again (call it on EVERY
        rank when the training graph changes, e.g. audio conditioning switched on)."""
        self._unused = None

    def _learn_unused(self) -> None:
        fired = torch.tensor([1.0 if f else 0.0 for f in self._fired], dtype=torch.float32, device=self.flat.flat_g.device)
        if self.world > 1:
            dist.all_reduce(fired, op=dist.ReduceOp.MAX, group=self.group)
        self._unused = {i for i, f in enumerate(fired.tolist()) if f == 0.0}
        if self._unused and self.world > 1 and self._side is None:
            # every rank holds the same set, so every rank comes here together.  The per-step check reduces HOST flags: over
            # the group itself when it is a gloo group, else over a gloo twin of it
            if dist.get_backend(self.group) == "gloo":
                self._side = self.group if self.group is not None else dist.group.WORLD
            else:
                ranks = dist.get_process_group_ranks(self.group if self.group is not None else dist.group.WORLD)
                try:
                    self._side = dist.new_group(ranks=ranks, backend="gloo", use_local_synchronization=True)
                except TypeError:       # older torch: every process of the default group must make the call
                    self._side = dist.new_group(ranks=ranks, backend="gloo")

    def _verify_unused(self) -> None:
        """The prediction against this step's facts, on every rank together; parameters it got wrong are exchanged now."""
        order = sorted(self._unused)
        flags = torch.tensor([1 if self._fired[i] else 0 for i in order], dtype=torch.int32)
        if self.world > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self._side)
        late = [i for i, f in zip(order, flags.tolist()) if f]
        if not late:
            return
        for i in late:
            p, slot = self.flat.params[i], self.flat.grad_view(i)
            if p.grad is not None and p.grad.data_ptr() != slot.data_ptr():
                slot.copy_(p.grad)                       # else: no gradient on this rank, the slot is still zero
            p.grad = slot
            if self.exchange:
                dist.all_reduce(slot, op=dist.ReduceOp.SUM, group=self.group)
            self._unused.discard(i)
        self.relearned = late

    def finish(self) -> None:
        for b in range(len(self.flat.buckets)):     # the rest, in bucket order
            self._ready[b] = True
        self._drain()
        for w in self._work:
            w.wait()
        self._work, self._armed = [], False
        if self._unused:
            self._verify_unused()
        if self.static_unused and self._unused is None:
            self._learn_unused()


class FlatAdam(torch.optim.Optimizer):
    """``torch.optim.Optimizer`` face of ``DiffusionTrainStep``'s flat Adam: ONE parameter group whose ``lr`` / ``betas`` /
    ``eps`` / ``weight_decay`` are what the fused clip + Adam kernel reads at every step, so that the reference's scheduler
    (``optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma)``, R/util/utils.py:116-123, stepped once per epoch at
    R/diffusion_trainer.py:296) -- or any other ``torch.optim.lr_scheduler`` -- drives the learning rate unchanged.
    ``step()`` runs the owner's ``optimizer_step()``; the moments live in the owner's flat buffers (``state_dict()`` /
    ``load_state_dict()`` are the owner's: the ``torch.optim.Adam`` checkpoint format)."""

    def __init__(self, owner: "DiffusionTrainStep", params, lr, betas, eps, weight_decay):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._owner = owner

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError("FlatAdam.step: closures are not supported (the loss is computed by DiffusionTrainStep)")
        self._owner.optimizer_step()

    def zero_grad(self, set_to_none: bool = True):
        self._owner.flat.zero_grad()

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):
            raise RuntimeError("FlatAdam keeps ONE parameter group (one flat buffer, one set of hyper-parameters)")
        super().add_param_group(param_group)

    def state_dict(self):
        return self._owner.state_dict()

    def load_state_dict(self, sd):
        self._owner.load_state_dict(sd)


class DiffusionTrainStep:
    """prepare_data -> forward -> loss -> backward -> gradient mean -> clip -> Adam, for ``model``.

    ``loss_fn(pred, x0) -> scalar tensor | dict with "total"`` replaces the built-in MSE (``mse_weight * sum_chw (pred - x0)^2``
    averaged over the batch) -- pass ``loss_config=config`` to get the reference's ``get_lossv2(config, pred, x0)["total"]``
    (R/diffusion_trainer.py:219, R/models/sal_losses.py:179-259: MSE or KL main term + weighted CC / SIM / NSS, every term with
    its HIP gradient, ``diff_sal_amd.sal_losses``).  ``self.optimizer`` is a ``torch.optim.Optimizer`` (``FlatAdam``) for
    learning-rate schedulers.

    ``model`` is a ``diff_sal_amd.SalUNet`` (called as ``model(x_t, t, feat_list, audio)``) or a
    ``VideoSaliencyModel`` (called as ``model({"img", "input", "audio"}, t)``, R/diffusion_trainer.py:212-218).
    Keyword names follow the YAML fields the reference trainer reads (R/cfgs/diffusion.yml:24-28, 39-60)."""

    def __init__(self, model: nn.Module, *, lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                 weight_decay: float = 0.0, grad_clip: float = 1.0, mse_weight: float = 1.0,
                 beta_schedule: str = "cosine", beta_start: float = 1e-4, beta_end: float = 0.02,
                 num_diffusion_timesteps: int = 1000, gaussian_dequantization: bool = True,
                 bucket_mb: float = 32.0, process_group=None, broadcast_buffers: bool = True,
                 store_clipped_grad: bool = False, exchange_single_rank: bool = False,
                 loss_fn: Optional[Callable] = None, loss_config=None, exchange: str = "allreduce"):
        self.model = model
        self.grad_clip, self.mse_weight = float(grad_clip), float(mse_weight)
        if loss_fn is not None and loss_config is not None:
            raise ValueError("DiffusionTrainStep: give loss_fn or loss_config, not both")
        if loss_config is not None:
            from . import sal_losses

            def loss_fn(pred, x0, _cfg=loss_config):            # R/diffusion_trainer.py:219
                return sal_losses.get_lossv2(_cfg, pred, x0)
        self.loss_fn = loss_fn
        self.last_losses: Optional[Dict] = None      # the loss dictionary of the last step when loss_fn returns one
        self.gaussian_dequantization = bool(gaussian_dequantization)
        self.store_clipped_grad = bool(store_clipped_grad)
        betas = to_torch(get_beta_schedule(beta_schedule, beta_start=beta_start, beta_end=beta_end,
                                           num_diffusion_timesteps=num_diffusion_timesteps))
        alphas_hat = (1.0 - betas).cumprod(dim=0)
        self.sqrt_alphas_hat = torch.sqrt(alphas_hat)
        self.sqrt_one_minus_alphas_hat = torch.sqrt(1.0 - alphas_hat)
        self.num_timesteps = int(betas.shape[0])
        self.flat = FlatParams(model, int(bucket_mb * (1 << 20)))
        self.optimizer = FlatAdam(self, self.flat.params, float(lr), (float(beta1), float(beta2)), float(eps), float(weight_decay))
        self.group = process_group
        self.reducer = GradReducer(self.flat, process_group, exchange_single_rank, exchange)
        self.world = self.reducer.world
        self.broadcast_buffers = bool(broadcast_buffers) and self.reducer.exchange
        self.step_count = 0
        self.last_norm: Optional[Tensor] = None
        self._rng = np.random  # the reference draws t0 from numpy's global generator (diffusion_trainer.py:111)

    # hyper-parameters live in the optimizer face's single parameter group (what a scheduler rewrites)
    def _hp(self, key):
        return self.optimizer.param_groups[0][key]

    lr = property(lambda self: float(self._hp("lr")), lambda self, v: self.optimizer.param_groups[0].__setitem__("lr", float(v)))
    betas = property(lambda self: tuple(float(b) for b in self._hp("betas")),
                     lambda self, v: self.optimizer.param_groups[0].__setitem__("betas", (float(v[0]), float(v[1]))))
    eps = property(lambda self: float(self._hp("eps")), lambda self, v: self.optimizer.param_groups[0].__setitem__("eps", float(v)))
    weight_decay = property(lambda self: float(self._hp("weight_decay")),
                            lambda self, v: self.optimizer.param_groups[0].__setitem__("weight_decay", float(v)))

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    # ---- R/diffusion_trainer.py:78-120 (training branch) ----
    def prepare_data(self, sal_maps: Tensor, *, t0: Optional[int] = None, noise: Optional[Tensor] = None,
                     dequant_noise: Optional[Tensor] = None):
        """sal_maps [B,1,H,W] in [0,1] -> (x0, x_t, t [B] int64, noise)."""
        from . import ops

        x = sal_maps.contiguous().float()
        if self.gaussian_dequantization:
            dq = torch.randn_like(x) if dequant_noise is None else dequant_noise
            x = ops.axpbypcz(x, 1.0, dq, 0.01)
        if noise is None:
            noise = torch.randn_like(x)
        if t0 is None:
            t0 = int(self._rng.randint(0, self.num_timesteps))
        t = torch.full((x.shape[0],), t0, dtype=torch.int64, device=x.device)
        x_t = ops.axpbypcz(x, float(self.sqrt_alphas_hat[t0]), noise, float(self.sqrt_one_minus_alphas_hat[t0]))
        return x, x_t, t, noise

    def _forward(self, x_t: Tensor, t: Tensor, cond: Dict) -> Tensor:
        from .sal_unet import SalUNet

        if isinstance(self.model, SalUNet):
            return self.model(x_t, t, cond["feat_list"], cond.get("audio_feat"))
        data = dict(cond)
        data["input"] = x_t
        return self.model(data, t)

    def _flatten_buffers(self) -> None:
        """Floating-point buffers (BatchNorm running statistics) become views into ONE flat tensor, once: the per-step broadcast
        of rank 0's buffers (DDP's ``broadcast_buffers=True``, R/model.py:15) is then a single collective on that tensor -- no
        gather before it and no copy kernel per buffer after it (the denoiser + encoders hold ~100 such buffers; the copies were
        most of the exchange's exposed time at one rank).  Kernels that update running statistics write through the views."""
        bufs = [b for b in self.model.buffers() if b.dtype == torch.float32 and b.numel() > 0]
        self._flat_buffers = None
        if not bufs or any(b.device != bufs[0].device for b in bufs):
            return
        off, offs = 0, []
        for b in bufs:
            offs.append(off)
            off += _aligned(b.numel())
        flat = torch.zeros(off, device=bufs[0].device, dtype=torch.float32)
        for b, o in zip(bufs, offs):
            view = flat[o:o + b.numel()].view(b.shape)
            view.copy_(b)
            b.data = view
        self._flat_buffers = flat

    def _sync_buffers(self) -> None:
        stale = False
        if getattr(self, "_flat_buffers", None) is not None:       # module.to(...) / a re-registered buffer: the views are gone
            lo = self._flat_buffers.data_ptr()
            hi = lo + 4 * self._flat_buffers.numel()
            stale = any(not (lo <= b.data_ptr() < hi) for b in self.model.buffers() if b.dtype == torch.float32 and b.numel() > 0)
        if getattr(self, "_flat_buffers", "unset") == "unset" or stale:
            self._flatten_buffers()
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        if self._flat_buffers is not None:
            dist.broadcast(self._flat_buffers, src=src, group=self.group)
        others = [b for b in self.model.buffers() if b.dtype.is_floating_point and b.dtype != torch.float32 and b.numel() > 0]
        for b in others:           # none in the reference's model; kept exact for foreign modules
            dist.broadcast(b, src=src, group=self.group)

    def loss_and_backward(self, x0: Tensor, x_t: Tensor, t: Tensor, cond: Dict) -> Tensor:
        """Forward + loss + backward + gradient exchange; leaves the rank-SUMMED gradient in ``flat.flat_g``."""
        from . import autograd_ops as ag

        self.model.train()
        if self.broadcast_buffers:
            self._sync_buffers()
        self.flat.zero_grad()
        self.reducer.arm()
        from . import ops

        with ops.batched_packs():        # every parameter's kernel layouts (forward, data-gradient) rebuilt by one launch
            pred = self._forward(x_t, t, cond)
            if self.loss_fn is None:
                loss = ag.mse_loss(pred, x0, self.mse_weight / x0.shape[0])
                self.last_losses = None
            else:
                out = self.loss_fn(pred, x0)
                if isinstance(out, dict):
                    self.last_losses = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
                    loss = out["total"]
                else:
                    self.last_losses, loss = None, out
                if not torch.is_tensor(loss) or loss.dim() != 0 or not loss.requires_grad:
                    raise RuntimeError("DiffusionTrainStep: loss_fn must return a scalar tensor (or a dict with 'total') that "
                                       "carries a gradient to the prediction")
            loss.backward()
        self.reducer.finish()
        return loss.detach()

    def optimizer_step(self) -> None:
        from . import ops

        self.step_count += 1
        gscale = 1.0 / self.world
        norm = ops.grad_norm(self.flat.flat_g, gscale) if self.grad_clip > 0 else None
        self.last_norm = norm
        ops.adam_step(self.flat.flat_p, self.flat.flat_g, self.flat.exp_avg, self.flat.exp_avg_sq, step=self.step_count,
                      lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, gscale=gscale, norm=norm,
                      max_norm=self.grad_clip, store_clipped_grad=self.store_clipped_grad)
        bump = getattr(self.model, "parameters_updated", None)
        if bump is not None:
            bump()
        for m in self.model.modules():  # nested denoiser inside a VideoSaliencyModel
            if m is not self.model and hasattr(m, "parameters_updated"):
                m.parameters_updated()

    def step(self, sal_maps: Tensor, cond: Dict, *, t0: Optional[int] = None, noise: Optional[Tensor] = None,
             dequant_noise: Optional[Tensor] = None) -> Tensor:
        """One training step on this rank's clips; returns the (detached, device-resident) loss."""
        x0, x_t, t, _ = self.prepare_data(sal_maps, t0=t0, noise=noise, dequant_noise=dequant_noise)
        loss = self.loss_and_backward(x0, x_t, t, cond)
        self.optimizer.step()          # the torch.optim.Optimizer face: step hooks and lr_scheduler bookkeeping see the step
        return loss

    # ---- checkpoint surface of torch.optim.Adam (R/diffusion_trainer.py:187-193, 263-268: "optim_dict") ----
    def _indexed_params(self):
        """(index in ``model.parameters()`` order -- the numbering torch.optim.Adam(model.parameters()) uses --, slot in the
        flat buffers or None for parameters that are not trained)."""
        slot = {id(p): k for k, p in enumerate(self.flat.params)}
        return [(i, p, slot.get(id(p))) for i, p in enumerate(self.model.parameters())]

    def state_dict(self) -> Dict:
        """The ``torch.optim.Adam.state_dict()`` format: ``state[i] = {step, exp_avg, exp_avg_sq}`` per trained parameter
        in ``model.parameters()`` order, one ``param_groups`` entry with lr / betas / eps / weight_decay -- what the
        reference trainer saves as ``optim_dict`` and loads back into ``optim.Adam`` (R/util/utils.py:116-123), so
        checkpoints move both ways.  The moments are gathered out of the flat buffers through ``FlatParams.offsets``."""
        idx = self._indexed_params()
        state = {}
        if self.step_count > 0:          # torch creates a parameter's state at its first step
            for i, p, k in idx:
                if k is None:
                    continue
                o = self.flat.offsets[k]
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.flat.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                            "exp_avg_sq": self.flat.exp_avg_sq[o:o + p.numel()].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": [i for i, _, _ in idx]}
        for k, v in self.optimizer.param_groups[0].items():     # whatever else lives in the group (a scheduler's initial_lr)
            if k != "params" and k not in group:
                group[k] = v
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict) -> None:
        """Inverse of ``state_dict``; also accepts a checkpoint written by ``torch.optim.Adam`` over the same module.
        Validates the parameter count and every moment's shape before touching anything."""
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("load_state_dict expects the torch.optim.Adam format {'state': ..., 'param_groups': [...]}")
        groups = sd["param_groups"]
        idx = self._indexed_params()
        n_saved = sum(len(g["params"]) for g in groups)
        if n_saved != len(idx):
            raise ValueError(f"optimizer checkpoint covers {n_saved} parameters, the module has {len(idx)}")
        if any(g.get("amsgrad", False) for g in groups):
            raise ValueError("amsgrad checkpoints are not supported (the reference sets amsgrad from config, default false)")
        # parameter numbering of the checkpoint = concatenation of the groups' param lists, in module order
        order = [i for g in groups for i in g["params"]]
        state = {int(k): v for k, v in sd["state"].items()}
        steps = set()
        staged = []
        for pos, (i, p, k) in enumerate(idx):
            st = state.get(order[pos])
            if st is None:
                continue
            if k is None:
                raise ValueError(f"checkpoint has optimizer state for parameter {i}, which is not trainable here")
            for name in ("exp_avg", "exp_avg_sq"):
                if tuple(st[name].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {name} of parameter {i}: shape {tuple(st[name].shape)} != {tuple(p.shape)}")
            steps.add(int(float(st["step"])))
            staged.append((k, p, st))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the flat optimizer keeps one step count")
        self.flat.exp_avg.zero_()
        self.flat.exp_avg_sq.zero_()
        for k, p, st in staged:
            o = self.flat.offsets[k]
            self.flat.exp_avg[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
            self.flat.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
        self.step_count = steps.pop() if steps else 0
        g0 = groups[0]
        self.lr = float(g0.get("lr", self.lr))
        self.betas = tuple(float(b) for b in g0.get("betas", self.betas))
        self.eps = float(g0.get("eps", self.eps))
        self.weight_decay = float(g0.get("weight_decay", self.weight_decay))
        for k, v in g0.items():
            if k not in ("params", "lr", "betas", "eps", "weight_decay", "amsgrad", "maximize", "foreach", "capturable",
                         "differentiable", "fused"):
                self.optimizer.param_groups[0][k] = v
