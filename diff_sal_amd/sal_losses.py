"""Evaluation metrics of the reference (R/models/sal_losses.py) on the device: same function names and return values
(a 0-dim tensor = the batch mean), computed by one fused two-pass kernel set (csrc/metrics.hip) instead of ~20 torch
reductions each.  ``saliency_metrics`` returns all four at once -- what the trainer's validation loop needs
(get_kl_cc_sim_loss_wo_weight, R/diffusion_trainer.py:741,797,868) -- for the price of one."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops

_ORDER = {"cc": 0, "sim": 1, "nss": 2, "kl": 3}


def saliency_metrics(s_map: torch.Tensor, gt: torch.Tensor):
    """{'cc', 'sim', 'nss', 'kl'} -> 0-dim device tensors (batch means), plus 'per_image' [B,4] in that order."""
    mean, per = ops.saliency_metrics(s_map, gt)
    out = {k: mean[i] for k, i in _ORDER.items()}
    out["per_image"] = per
    return out


class _SaliencyTermsFn(torch.autograd.Function):
    """(cc, sim, nss, kl) batch means of (pred, gt) as ONE differentiable op: forward = the fused two-pass metric kernels,
    backward = one elementwise kernel that evaluates the four closed-form gradients from the forward's partial sums
    (csrc/metrics.hip::metric_bwd_kernel).  gt receives no gradient (the reference's targets never require one)."""

    @staticmethod
    def forward(ctx, pred, gt):
        mean, _, (p, g, ws) = ops.saliency_metrics(pred, gt, keep_ws=True)
        ctx.save_for_backward(p, g, ws)
        ctx.shape = pred.shape
        return mean

    @staticmethod
    def backward(ctx, dmean):
        p, g, ws = ctx.saved_tensors
        return ops.saliency_metrics_bwd(p, g, ws, dmean.contiguous().float()).view(ctx.shape), None


def saliency_terms(s_map: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """[4] = (cc, sim, nss, kl) batch means; differentiable with respect to ``s_map``."""
    if s_map.requires_grad and torch.is_grad_enabled():
        return _SaliencyTermsFn.apply(s_map, gt)
    return ops.saliency_metrics(s_map, gt)[0]


def cc_s2(s_map, gt):
    """R/models/sal_losses.py:63-97."""
    return saliency_terms(s_map, gt)[0]


def similarity2(s_map, gt):
    """R/models/sal_losses.py:155-176 (with normalize_map2, :134-152)."""
    return saliency_terms(s_map, gt)[1]


def nss2(s_map, gt):
    """R/models/sal_losses.py:14-37."""
    return saliency_terms(s_map, gt)[2]


def kldiv2(s_map, gt):
    """R/models/sal_losses.py:100-131."""
    return saliency_terms(s_map, gt)[3]


def cross_entropy_loss(output, label, weights, batch_average=False, is_reduce=True):
    """R/models/sal_losses.py:48-63: BCE with logits against label / 255, summed per clip, times ``weights`` (a scalar
    ``ce_weight`` in get_kl_cc_sim_loss, or one weight per clip)."""
    b = output.size(0)
    loss = F.binary_cross_entropy_with_logits(output.reshape(b, -1), label.reshape(b, -1) / 255, reduction="none").sum(1)
    loss = loss * weights
    if is_reduce:
        loss = torch.sum(loss)
    if batch_average:
        loss = loss / torch.sum(torch.as_tensor(weights))
    return loss


def get_kl_cc_sim_loss(config, pred_map, gt):
    """R/models/sal_losses.py:179-206: (main, cc, sim, nss) with the configuration's switches and weights.  The main term is KL
    (``loss_kl``), else the weighted MSE (``loss_mse``: the shipped configuration, R/cfgs/diffusion.yml:39-51); every enabled
    term carries its gradient.  One metric launch set serves all four saliency terms.  ``loss_ce`` (:48-63: binary
    cross-entropy with logits on labels / 255, summed over pixels and clips, times ``ce_weight``) is elementwise torch on the
    device tensor -- no configuration of the reference enables it, so it has no kernel of its own."""
    from . import autograd_ops as ag

    lc = config.loss
    zero = torch.zeros((), device=pred_map.device)
    terms = None
    if any(getattr(lc, k, False) for k in ("loss_kl", "loss_cc", "loss_sim", "loss_nss")):
        terms = saliency_terms(pred_map, gt)
    if getattr(lc, "loss_kl", False):
        main = lc.kl_weight * terms[3]
    elif getattr(lc, "loss_ce", False):
        main = cross_entropy_loss(pred_map, gt, lc.ce_weight)
    elif getattr(lc, "loss_mse", False):
        main = ag.mse_loss(pred_map, gt, float(lc.mse_weight) / pred_map.shape[0])
    else:
        main = zero
    cc = lc.cc_weight * terms[0] if getattr(lc, "loss_cc", False) else zero
    sim = lc.sim_weight * terms[1] if getattr(lc, "loss_sim", False) else zero
    nss = lc.nss_weight * terms[2] if getattr(lc, "loss_nss", False) else zero
    return main, cc, sim, nss


def get_lossv2(config, predictions, gt):
    """R/models/sal_losses.py:239-259: the training loss dictionary; ``total`` = main + cc + sim + nss."""
    main, cc, sim, nss = get_kl_cc_sim_loss(config, predictions, gt)
    return {"total": main + cc + sim + nss, "main": main, "cc": cc, "sim": sim, "nss": nss}


def get_kl_cc_sim_loss_wo_weight(config, pred_map, gt):
    """R/models/sal_losses.py:207-236: the validation-loop dictionary (main = KL only if config.loss.loss_kl)."""
    m = saliency_metrics(pred_map, gt)
    kl = m["kl"] if getattr(getattr(config, "loss", None), "loss_kl", False) else torch.zeros((), device=pred_map.device)
    return {"total": m["nss"] + m["cc"] + m["sim"], "main": kl, "cc": m["cc"], "sim": m["sim"], "nss": m["nss"]}
