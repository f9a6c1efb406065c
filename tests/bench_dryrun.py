"""TEST-ONLY support for ``bench.py --dryrun-cpu``: lets the benchmark's launch / rendezvous / timing / reporting logic
run without a GPU (``tests/test_bench_launch.py`` spawns ``bench.py --gpus 2 --dryrun-cpu`` and checks that two gloo ranks
came up).  The HIP ops are replaced by the PyTorch emulator of tests/hip_emulator.py on a tiny configuration; nothing here
is imported by the package, and the printed line is marked as not a measurement."""
import torch

from oracle import unet_oracle as O
from tests import hip_emulator
from diffusion_pruning_amd import ops as real_ops


def _install_emulator():
    for name in ("conv_gemm", "linear", "groupnorm", "layernorm", "attention"):
        setattr(real_ops, name, getattr(hip_emulator, name))


def install_infer() -> dict:
    """route ops.* to the emulator; returns the constructor arguments of the tiny U-Net"""
    _install_emulator()
    cfg = O.TINY
    torch.set_num_threads(2)
    return dict(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                cross_attention_dim=cfg.cross_attention_dim)


def install_train():
    """a reference-API stand-in U-Net whose output depends differentiably on the gates (tests/test_distributed_cpu.py),
    text embedding width, number of codes"""
    from diffusion_pruning_amd.unet import UNet2DConditionModelGated
    from tests.test_distributed_cpu import StubUNet
    torch.set_num_threads(2)
    cfg = O.TINY
    real = UNet2DConditionModelGated(block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.num_heads,
                                     cross_attention_dim=cfg.cross_attention_dim)
    return StubUNet(real), 16, 4


class _StubFineTune:
    """stand-in for train_step.GraphedFineTunerStep on CPU: a small trainable map fitted to a fixed target, one SGD step per
    train_step -- enough for the launcher / rank -> expert mapping / record of ``bench.py --config finetune --dryrun-cpu``"""

    def __init__(self, expert: int):
        g = torch.Generator().manual_seed(1000 + expert)
        self.w = torch.nn.Parameter(torch.randn(8, 8, generator=g) * 0.1)
        self.t = torch.randn(4, 8, generator=g)
        self.x = torch.randn(4, 8, generator=g)

    def train_step(self, optimizer, batch):
        loss = ((self.x @ self.w - self.t) ** 2).mean()
        loss.backward()
        with torch.no_grad():
            self.w -= 0.05 * self.w.grad
            self.w.grad = None
        return {"loss": loss.detach()}


def install_finetune(expert: int, batch: int):
    """(step, batch, trainable parameters, keep ratio) of the stand-in expert"""
    torch.set_num_threads(2)
    step = _StubFineTune(expert)
    return step, {"n": batch}, step.w.numel(), 0.4 + 0.05 * expert
