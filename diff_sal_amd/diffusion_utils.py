"""Noise-schedule helpers with the reference's names (R/models/diffusion_decoder/diffusion_utils.py:5-48)."""
import numpy as np
import torch


def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """float64 numpy betas for 'quad' | 'linear' | 'const' | 'jsd' | 'sigmoid' | 'cosine'."""
    n = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(n, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        s = np.linspace(-6, 6, n)
        betas = 1.0 / (np.exp(-s) + 1.0) * (beta_end - beta_start) + beta_start
    elif beta_schedule == "cosine":
        # Nichol & Dhariwal cosine alpha-bar with s = 0.008, evaluated on n+1 *linspace(0, n+1)* knots
        # exactly as the reference does (:40-46), clipped to 0.999
        steps = n + 1
        x = np.linspace(0, steps, steps)
        ac = np.cos(((x / steps) + 0.008) / 1.008 * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        betas = np.clip(1 - ac[1:] / ac[:-1], a_min=0, a_max=0.999)
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (n,)
    return betas


def to_torch(tensor):
    return torch.tensor(tensor, dtype=torch.float32)
