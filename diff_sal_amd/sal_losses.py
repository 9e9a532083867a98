"""Evaluation metrics of the reference (R/models/sal_losses.py) on the device: same function names and return values
(a 0-dim tensor = the batch mean), computed by one fused two-pass kernel set (csrc/metrics.hip) instead of ~20 torch
reductions each.  ``saliency_metrics`` returns all four at once -- what the trainer's validation loop needs
(get_kl_cc_sim_loss_wo_weight, R/diffusion_trainer.py:741,797,868) -- for the price of one."""
from __future__ import annotations

import torch

from . import ops

_ORDER = {"cc": 0, "sim": 1, "nss": 2, "kl": 3}


def saliency_metrics(s_map: torch.Tensor, gt: torch.Tensor):
    """{'cc', 'sim', 'nss', 'kl'} -> 0-dim device tensors (batch means), plus 'per_image' [B,4] in that order."""
    mean, per = ops.saliency_metrics(s_map, gt)
    out = {k: mean[i] for k, i in _ORDER.items()}
    out["per_image"] = per
    return out


def cc_s2(s_map, gt):
    """R/models/sal_losses.py:63-97."""
    return ops.saliency_metrics(s_map, gt)[0][0]


def similarity2(s_map, gt):
    """R/models/sal_losses.py:155-176 (with normalize_map2, :134-152)."""
    return ops.saliency_metrics(s_map, gt)[0][1]


def nss2(s_map, gt):
    """R/models/sal_losses.py:14-37."""
    return ops.saliency_metrics(s_map, gt)[0][2]


def kldiv2(s_map, gt):
    """R/models/sal_losses.py:100-131."""
    return ops.saliency_metrics(s_map, gt)[0][3]


def get_kl_cc_sim_loss_wo_weight(config, pred_map, gt):
    """R/models/sal_losses.py:207-236: the validation-loop dictionary (main = KL only if config.loss.loss_kl)."""
    m = saliency_metrics(pred_map, gt)
    kl = m["kl"] if getattr(getattr(config, "loss", None), "loss_kl", False) else torch.zeros((), device=pred_map.device)
    return {"total": m["nss"] + m["cc"] + m["sim"], "main": kl, "cc": m["cc"], "sim": m["sim"], "nss": m["nss"]}
