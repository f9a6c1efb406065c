"""MI355X-native masked-U-Net denoising path for APTP (rezashkv/diffusion_pruning).

Host code is Python on PyTorch-ROCm; all U-Net compute goes through libaptp_hip.so (hand-written gfx950 HIP, C ABI
in include/aptp_hip.h).  There is no CPU or PyTorch fallback for the U-Net path.
"""
__version__ = "0.1.0"
