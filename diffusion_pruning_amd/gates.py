"""Gate primitives with the reference's names and state (pdm/models/unet/gates.py:9-55).

In the reference a gate is an ``nn.Module`` whose ``forward`` multiplies an activation by the expanded mask — a full
extra read+write of the tensor per gate (164 such passes per U-Net forward).  Here the gate keeps the
same state (``gate_f`` of shape [Bg, width], default ones(1, width), replaced by ``set_structure_value``) but is never
launched on its own: the owning block hands ``gate_f`` to the epilogue of the producing HIP kernel
(aptp_conv_gemm ``colgate`` / ``depth``), or — when the mask is hard and shared by the batch — uses it to compact the
weights so dead channels/heads/blocks are skipped altogether.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn


class VirtualGate(nn.Module):
    def __init__(self, width: int, bs: int = 1):
        super().__init__()
        self.width = width
        self.gate_f = torch.ones(bs, width)       # plain attribute, not a Parameter (gates.py:13)
        self._host: Optional[torch.Tensor] = None  # CPU copy made once per set_structure (no per-forward sync)

    def set_structure_value(self, value: torch.Tensor):
        self.gate_f = value
        self._host = None

    # ---- host-side classification (used to pick compact vs dense execution) ----------------------------------
    def host_value(self) -> torch.Tensor:
        if self._host is None:
            self._host = self.gate_f.detach().float().cpu()
        return self._host

    def set_host_value(self, host: torch.Tensor):
        self._host = host

    def hard_uniform(self) -> Optional[torch.Tensor]:
        """If every entry is exactly 0 or 1 and all batch rows agree, return the 1-D {0,1} mask, else None."""
        h = self.host_value()
        if h.dim() == 1:
            h = h[:, None]
        if not bool(((h == 0) | (h == 1)).all()):
            return None
        if h.shape[0] > 1 and not bool((h == h[:1]).all()):
            return None
        return h[0]

    def forward(self, x):  # pragma: no cover - kept for API parity; the product path never calls it
        raise RuntimeError("gates are fused into the HIP kernels of the owning block; they are not launched standalone")


class WidthGate(VirtualGate):
    pass


class LinearWidthGate(WidthGate):
    pass


class DepthGate(VirtualGate):
    def __init__(self, width: int = 1):
        super().__init__(width)
        self.gate_f = torch.ones(1)
