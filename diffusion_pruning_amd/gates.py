"""Gate primitives with the reference's names and state (pdm/models/unet/gates.py:9-55).

In the reference a gate is an ``nn.Module`` whose ``forward`` multiplies an activation by the expanded mask — a full
extra read+write of the tensor per gate (164 such passes per U-Net forward).  Here the gate keeps the
same state (``gate_f`` of shape [Bg, width], default ones(1, width), replaced by ``set_structure_value``) but is never
launched on its own: the owning block hands ``gate_f`` to the epilogue of the producing HIP kernel
(aptp_conv_gemm ``colgate`` / ``depth``), or — when the mask is hard and shared by the batch — uses it to compact the
weights so dead channels/heads/blocks are skipped altogether.

``forward`` keeps the reference's public semantics for callers that use a gate as a module of its own (analysis scripts in the
style of scripts/other/depth_analysis.py): it runs the same HIP gate kernels the training path uses (``aptp_gate_bwd`` in its
forward form, ``aptp_depth_lerp``), differentiably, on bf16 device tensors -- the product's activation format.  There is no
host / PyTorch fallback: other tensors are refused with a message that says so.
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

    @staticmethod
    def _check(x: torch.Tensor, what: str):
        if not (x.is_cuda and x.dtype == torch.bfloat16):
            raise TypeError(f"{what}: the standalone gate forward runs the HIP gate kernel on bf16 device tensors (got "
                            f"{x.dtype} on {x.device}); there is no host fallback -- inside the U-Net the gates are fused into the "
                            f"producing kernels' epilogues")

    def _gate2d(self, x: torch.Tensor) -> torch.Tensor:
        g = self.gate_f
        if g.dim() == 1:
            g = g[None]
        if x.shape[0] % g.shape[0] != 0:
            raise ValueError(f"gate batch {g.shape[0]} does not divide the activation batch {x.shape[0]}")
        return g.to(device=x.device, dtype=torch.float32)

    def forward(self, x):
        """x [B, C, H, W] * gate_f[b % Bg, c // (C / width)]  (gates.py:15-21; the CFG batch doubling is the b % Bg)"""
        from . import autograd as AG
        self._check(x, type(self).__name__ + ".forward")
        if x.dim() != 4 or x.shape[1] % self.width != 0:
            raise ValueError(f"expected [B, C, H, W] with C a multiple of the gate width {self.width}, got {tuple(x.shape)}")
        y = AG.GateFn.apply(x.permute(0, 2, 3, 1), self._gate2d(x))          # NHWC view (a copy unless x is channels_last)
        return y.permute(0, 3, 1, 2)


class WidthGate(VirtualGate):
    pass


class LinearWidthGate(WidthGate):
    def forward(self, x):
        """x [B, L, C] * gate_f[b % Bg, c // (C / width)]  (gates.py:49-55)"""
        from . import autograd as AG
        self._check(x, "LinearWidthGate.forward")
        if x.dim() != 3 or x.shape[2] % self.width != 0:
            raise ValueError(f"expected [B, L, C] with C a multiple of the gate width {self.width}, got {tuple(x.shape)}")
        return AG.GateFn.apply(x, self._gate2d(x))


class DepthGate(VirtualGate):
    def __init__(self, width: int = 1):
        super().__init__(width)
        self.gate_f = torch.ones(1)

    def forward(self, x):
        """(input_hidden_states, output_tensor) -> (1 - d) * input + d * output, d = gate_f[b % Bg]  (gates.py:36-42)"""
        from . import autograd as AG
        x_in, out = x
        self._check(x_in, "DepthGate.forward")
        self._check(out, "DepthGate.forward")
        if x_in.shape != out.shape or x_in.dim() != 4:
            raise ValueError(f"expected two [B, C, H, W] tensors of one shape, got {tuple(x_in.shape)} and {tuple(out.shape)}")
        d = self.gate_f.reshape(-1).to(device=out.device, dtype=torch.float32)
        if out.shape[0] % d.shape[0] != 0:
            raise ValueError(f"gate batch {d.shape[0]} does not divide the activation batch {out.shape[0]}")
        y = AG.depth_lerp(x_in.permute(0, 2, 3, 1).contiguous(), out.permute(0, 2, 3, 1).contiguous(), d)
        return y.permute(0, 3, 1, 2)
