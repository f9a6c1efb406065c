"""Relaxation utilities with the reference's names and numerics (pdm/utils/estimation_utils.py:5-75).

Semantics kept bit-for-bit where the reference is deterministic: the Gumbel noise is drawn on the HOST RNG exactly
as the reference does (``torch.rand`` on CPU; a fresh ``Generator().manual_seed(0)`` per call when ``fixed_seed``),
so seeded runs reproduce the reference's architecture codes.  Everything after the draw runs on the logits' device.
"""
from __future__ import annotations

import torch


def sample_gumbel(shape, eps: float = 1e-20, fixed_seed: bool = False) -> torch.Tensor:
    """G = -log(-log(U + eps) + eps), U ~ U(0,1) on the CPU generator (estimation_utils.py:5-10)."""
    gen = torch.Generator().manual_seed(0) if fixed_seed else None
    u = torch.rand(shape, generator=gen)
    return -torch.log(eps - torch.log(u + eps))


def hard_concrete(out: torch.Tensor) -> torch.Tensor:
    """Straight-through threshold at 0.5 (estimation_utils.py:67-75): value {0,1}, gradient identity."""
    hard = (out >= 0.5).to(out.dtype)
    return (hard - out).detach() + out


def _noisy_sigmoid(logits: torch.Tensor, temperature: float, offset, fixed_seed: bool) -> torch.Tensor:
    noise = sample_gumbel(logits.size(), fixed_seed=fixed_seed).to(logits.device)
    return torch.sigmoid((logits + noise + offset) / temperature)


def vector_gumbel_softmax(logits, temperature, offset=0, force_width_non_zero=False, fixed_seed=False):
    """estimation_utils.py:13-31: rows whose hard version is all-zero get +0.5 on their first entry."""
    y = _noisy_sigmoid(logits, temperature, offset, fixed_seed)
    if not force_width_non_zero:
        return y
    dead = hard_concrete(y).sum(dim=1) == 0
    if not bool(dead.any()):
        return y
    bumped = y.clone()
    bumped[dead, 0] = y[dead, 0] + 0.5
    return bumped


def gumbel_softmax_sample(logits, temperature, offset=0, force_width_non_zero=False, fixed_seed=False):
    """estimation_utils.py:34-46"""
    return vector_gumbel_softmax(logits, temperature, offset, force_width_non_zero, fixed_seed)


def importance_gumbel_softmax_sample(logits, temperature, offset=0, fixed_seed=False, noise=None):
    """estimation_utils.py:49-64: softmax -> cumsum -> flip gives monotone keep-probabilities; logit with eps=1e-6.
    noise: Gumbel noise already drawn (in the reference's order) and moved to the device by the caller."""
    p = torch.flip(torch.cumsum(torch.softmax(logits, dim=1), dim=1), dims=[1])
    eps = 1e-6
    x = torch.log(p + eps) - torch.log1p(-(p - eps))
    if noise is not None:
        return torch.sigmoid((x + noise + offset) / temperature)
    return _noisy_sigmoid(x, temperature, offset, fixed_seed)
