"""attention forward forms against an fp64 reference on peaked inputs: output and log-sum-exp (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_pruning_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, h, L, amp) in [(2, 2, 512, 1.0), (2, 2, 512, 3.0), (1, 2, 4096, 3.0), (2, 3, 256, 4.0)]:
    q = (torch.randn(B, L, h * 64, device=dev) * amp).bfloat16()
    k = (torch.randn(B, L, h * 64, device=dev) * amp).bfloat16()
    v = torch.randn(B, L, h * 64, device=dev).bfloat16()
    def heads(t): return t.double().view(B, L, h, 64).transpose(1, 2)
    s = heads(q) @ heads(k).transpose(-1, -2) / 8.0
    ref = (torch.softmax(s, -1) @ heads(v)).transpose(1, 2).reshape(B, L, h * 64)
    lse_ref = torch.logsumexp(s, -1) * 1.4426950408889634
    row = []
    for var in (4, 3, 6):
        ops.ATTN_VARIANT = var
        lse = torch.zeros(B, h, L, device=dev)
        o = ops.attention(q, k, v, h, lse=lse)
        e = float((o.double() - ref).norm() / ref.norm())
        el = float((lse.double() - lse_ref).abs().max())
        worst = float((o.double() - ref).abs().max())
        row.append(f"v{var}: rel-L2 {e:.2e} max|err| {worst:.2e} lse max|err| {el:.2e}")
    ops.ATTN_VARIANT = 0
    print(f"B{B} h{h} L{L} amp{amp}: " + "   ".join(row), flush=True)

# where do v6 and v4 differ?
B, h, L, amp = 2, 2, 512, 3.0
q = (torch.randn(B, L, h * 64, device=dev) * amp).bfloat16()
k = (torch.randn(B, L, h * 64, device=dev) * amp).bfloat16()
v = torch.randn(B, L, h * 64, device=dev).bfloat16()
ops.ATTN_VARIANT = 4
o4 = ops.attention(q, k, v, h).float()
ops.ATTN_VARIANT = 6
o6 = ops.attention(q, k, v, h).float()
ops.ATTN_VARIANT = 0
d = (o6 - o4).abs()
nz = (d > 0)
print("elements that differ:", int(nz.sum()), "of", d.numel(), "max", float(d.max()))
print("by d % 64 (count):", nz.reshape(-1, 64).sum(0).tolist())
print("by query % 32 (count):", nz.reshape(B, L // 32, 32, h * 64).sum((0, 1, 3)).tolist())
print("by query // 128 (count):", nz.reshape(B, L // 128, 128, h * 64).sum((0, 2, 3)).tolist())
