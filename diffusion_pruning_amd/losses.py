"""Losses of the APTP pruning / fine-tuning steps (pdm/losses/contrastive_loss.py:5-22, resource_loss.py:5-23) and the
SNR helper (pdm/utils/metric_utils.py:1-24).  Tiny tensor math: stays PyTorch on the device."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class ContrastiveLoss(nn.Module):
    def __init__(self, arch_vector_temperature: float = 1.0, prompt_embedding_temperature: float = 1.0):
        super().__init__()
        self.arch_vector_temperature = arch_vector_temperature
        self.prompt_embedding_temperature = prompt_embedding_temperature

    def forward(self, prompt_embeddings, arch_vectors, return_similarity: bool = False):
        a = arch_vectors / arch_vectors.norm(dim=1, keepdim=True)
        z = prompt_embeddings / prompt_embeddings.norm(dim=1, keepdim=True)
        arch_sim = F.softmax(a @ a.T / self.arch_vector_temperature, dim=-1)
        text_sim = F.softmax(z @ z.T / self.prompt_embedding_temperature, dim=-1)
        loss = F.binary_cross_entropy(arch_sim.T, text_sim.T, reduction="mean")
        if return_similarity:
            return loss, arch_sim.detach().cpu().numpy()
        return loss


class ResourceLoss(nn.Module):
    def __init__(self, p: float = 0.9, loss_type: str = "log"):
        super().__init__()
        assert loss_type in ["log", "mae", "mse"], f"Unknown loss type {loss_type}"
        self.p = p
        self.loss_type = loss_type

    def forward(self, resource_ratio):
        if self.loss_type == "log":
            if not torch.is_tensor(resource_ratio) or not resource_ratio.is_cuda:
                if resource_ratio > self.p:
                    return torch.log(resource_ratio / self.p)
                return torch.log(self.p / resource_ratio)
            # the same two branches without reading the comparison back on the host (one device -> host wait per step otherwise)
            return torch.where(resource_ratio > self.p, torch.log(resource_ratio / self.p), torch.log(self.p / resource_ratio))
        if self.loss_type == "mae":
            return torch.abs(resource_ratio - self.p)
        return (resource_ratio - self.p) ** 2


def compute_snr(noise_scheduler, timesteps: torch.Tensor) -> torch.Tensor:
    """SNR(t) = alpha_bar_t / (1 - alpha_bar_t) gathered at `timesteps` (metric_utils.py:1-24)."""
    ac = noise_scheduler.alphas_cumprod.to(device=timesteps.device)
    alpha = (ac ** 0.5)[timesteps].float()
    sigma = ((1.0 - ac) ** 0.5)[timesteps].float()
    return (alpha / sigma) ** 2
