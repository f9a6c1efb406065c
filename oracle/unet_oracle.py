"""CPU oracle for APTP's gated SD-2.1 U-Net forward — TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in plain PyTorch CPU ops (fp32 or fp64), of the arithmetic the
reference executes for one denoise step.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package (``diffusion_pruning_amd``) never does.

PARITY PIN STATUS: the reference's U-Net cannot be imported in the build container (it subclasses
``diffusers==0.23.1`` blocks, which are absent and un-vendored), and the reference has no
tests or golden vectors, so the *U-Net arithmetic* of this oracle is **parity unpinned** against a live
reference run.  What pins it instead (tests/test_oracle.py):
  * the exact SD-2.1 parameter count 865,910,724 produced by ``init_params``;
  * gate semantics checked against golden vectors generated from the reference's own ``gates.py``
    (tests/golden/make_golden.py imports /root/reference/pdm/models/unet/gates.py by file path);
  * the invariants of SURVEY App. B.6 (mask==1 == ungated, gated==pruned for attention/FF,
    ResNet gated-pruned == conv2(SiLU(beta_dead)), depth in {0,1} == skip/keep, CFG tiling);
  * fp64-vs-fp32 self-consistency.

Reference call sites restated here (file:line relative to /root/reference):
  gates                 pdm/models/unet/gates.py:9-55
  ResNet block          pdm/models/unet/blocks.py:293-371 (width gated), :482-584 (width+depth gated),
                        prune semantics :424-465, :641-697
  attention             pdm/models/unet/blocks.py:194-280 (head gate :250-255, SDPA :258-260), prune :153-187
  GEGLU / FF            pdm/models/unet/blocks.py:41-50, :121-129
  transformer block     pdm/models/unet/blocks.py:763-851
  Transformer2D         pdm/models/unet/blocks.py:1139-1355 (depth gate :1345-1351, dropped :1190-1194)
  U-Net forward         pdm/models/unet/unet_2d_conditional.py:1415-1726
  structure plumbing    pdm/models/unet/unet_2d_conditional.py:1332-1413, blocks.py:1814-1861
diffusers-0.23.1 behaviours relied upon are listed in SURVEY.md App. E.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # diffusers naming quirk (unet_2d_conditional.py:789-795): "attention_head_dim" is the NUMBER of heads
    num_heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    # which levels carry cross-attention transformers (SD-2.1: first three down levels; mirrored on the way up)
    attn_levels: Tuple[bool, ...] = (True, True, True, False)
    ff_gate_width: int = 32

    @property
    def temb_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def n_levels(self) -> int:
        return len(self.block_out_channels)


SD21 = UNetConfig()
# a small config with the same topology for fast tests (head_dim stays 64 so the HIP attention kernel applies)
TINY = UNetConfig(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=64)


# ----------------------------------------------------------------------------------------------------------------
# architecture walk: module list in the reference's structure order
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class ResnetSpec:
    name: str
    cin: int
    cout: int
    depth_gated: bool
    skip_dim: int = 0  # >0: up-block resnet fed with cat([h, skip]); skip_dim = channels of the skip tensor


@dataclass
class AttnSpec:
    name: str
    ch: int
    heads: int
    depth_gated: bool


@dataclass
class BlockSpec:
    name: str
    kind: str  # "down" | "mid" | "up"
    resnets: List[ResnetSpec] = field(default_factory=list)
    attns: List[AttnSpec] = field(default_factory=list)
    sampler: Optional[str] = None  # parameter prefix of the down/up-sampler conv, if any
    sampler_ch: int = 0


def build_specs(cfg: UNetConfig) -> List[BlockSpec]:
    """Block/module list following diffusers' construction order with the reference's
    gate placement (blocks.py:1717-1807 down, :2554-2736 mid, :2004-2243 / :2419-2550 up)."""
    boc = cfg.block_out_channels
    specs: List[BlockSpec] = []
    out_ch = boc[0]
    for i in range(cfg.n_levels):
        in_ch, out_ch = out_ch, boc[i]
        b = BlockSpec(name=f"down_blocks.{i}", kind="down")
        for j in range(cfg.layers_per_block):
            last = j == cfg.layers_per_block - 1
            b.resnets.append(ResnetSpec(f"down_blocks.{i}.resnets.{j}", in_ch if j == 0 else out_ch, out_ch, last))
            if cfg.attn_levels[i]:
                b.attns.append(AttnSpec(f"down_blocks.{i}.attentions.{j}", out_ch, cfg.num_heads[i], last))
        if i != cfg.n_levels - 1:
            b.sampler, b.sampler_ch = f"down_blocks.{i}.downsamplers.0.conv", out_ch
        specs.append(b)
    mid = BlockSpec(name="mid_block", kind="mid")
    mid.resnets = [ResnetSpec("mid_block.resnets.0", boc[-1], boc[-1], False),
                   ResnetSpec("mid_block.resnets.1", boc[-1], boc[-1], False)]
    mid.attns = [AttnSpec("mid_block.attentions.0", boc[-1], cfg.num_heads[-1], False)]
    specs.append(mid)
    rev = list(reversed(boc))
    rev_heads = list(reversed(cfg.num_heads))
    rev_attn = list(reversed(cfg.attn_levels))
    out_ch = rev[0]
    for i in range(cfg.n_levels):
        prev_out = out_ch
        out_ch = rev[i]
        in_ch = rev[min(i + 1, cfg.n_levels - 1)]
        b = BlockSpec(name=f"up_blocks.{i}", kind="up")
        n = cfg.layers_per_block + 1
        for j in range(n):
            skip = in_ch if j == n - 1 else out_ch
            rin = prev_out if j == 0 else out_ch
            last = j == n - 1
            b.resnets.append(ResnetSpec(f"up_blocks.{i}.resnets.{j}", rin + skip, out_ch, last, skip_dim=skip))
            if rev_attn[i]:
                b.attns.append(AttnSpec(f"up_blocks.{i}.attentions.{j}", out_ch, rev_heads[i], last))
        if i != cfg.n_levels - 1:
            b.sampler, b.sampler_ch = f"up_blocks.{i}.upsamplers.0.conv", out_ch
        specs.append(b)
    return specs


def get_structure(cfg: UNetConfig) -> Dict[str, List[List[int]]]:
    """{"width": [[...]], "depth": [[0|1]]} exactly as UNet2DConditionModelGated.get_structure()
    (unet_2d_conditional.py:1332-1363): per container, all resnets first, then all attentions."""
    width, depth = [], []
    for b in build_specs(cfg):
        for r in b.resnets:
            width.append([cfg.norm_num_groups])
            depth.append([1 if r.depth_gated else 0])
        for a in b.attns:
            width.append([a.heads, a.heads, cfg.ff_gate_width])
            depth.append([1 if a.depth_gated else 0])
    return {"width": width, "depth": depth}


def split_arch_vector(cfg: UNetConfig, arch: torch.Tensor) -> Dict[str, List[torch.Tensor]]:
    """[B, sum(width)+n_depth] -> {"width": [70 x [B,w]], "depth": [14 x [B]]} (hypernet.py:86-101)."""
    st = get_structure(cfg)
    wl = [w for sub in st["width"] for w in sub]
    nd = sum(d for sub in st["depth"] for d in sub)
    assert arch.shape[1] == sum(wl) + nd
    out_w, s = [], 0
    for w in wl:
        out_w.append(arch[:, s:s + w])
        s += w
    out_d = [arch[:, s + i] for i in range(nd)]
    return {"width": out_w, "depth": out_d}


def assign_gates(cfg: UNetConfig, arch_vectors: Dict[str, List[torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Distribute the flat width/depth lists to module names, consuming them in the reference's order
    (set_structure pops from the caller's lists: unet_2d_conditional.py:1365-1413; blocks.py:1833-1861)."""
    w, d = list(arch_vectors["width"]), list(arch_vectors["depth"])
    gates: Dict[str, torch.Tensor] = {}
    for b in build_specs(cfg):
        for r in b.resnets:
            gates[r.name + ".gate"] = w.pop(0)
            if r.depth_gated:
                gates[r.name + ".depth_gate"] = d.pop(0)
        for a in b.attns:
            gates[a.name + ".attn1.gate"] = w.pop(0)
            gates[a.name + ".attn2.gate"] = w.pop(0)
            gates[a.name + ".ff.gate"] = w.pop(0)
            if a.depth_gated:
                gates[a.name + ".depth_gate"] = d.pop(0)
    assert not w and not d, "arch vector lists not fully consumed"
    return gates


# ----------------------------------------------------------------------------------------------------------------
# parameters (diffusers state-dict names and shapes)
# ----------------------------------------------------------------------------------------------------------------
def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    c0, T, X = cfg.block_out_channels[0], cfg.temb_dim, cfg.cross_attention_dim

    def conv(name, cin, cout, k):
        s[name + ".weight"] = (cout, cin, k, k)
        s[name + ".bias"] = (cout,)

    def lin(name, cin, cout, bias=True):
        s[name + ".weight"] = (cout, cin)
        if bias:
            s[name + ".bias"] = (cout,)

    def norm(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    conv("conv_in", cfg.in_channels, c0, 3)
    lin("time_embedding.linear_1", c0, T)
    lin("time_embedding.linear_2", T, T)
    for b in build_specs(cfg):
        for r in b.resnets:
            norm(r.name + ".norm1", r.cin)
            conv(r.name + ".conv1", r.cin, r.cout, 3)
            lin(r.name + ".time_emb_proj", T, r.cout)
            norm(r.name + ".norm2", r.cout)
            conv(r.name + ".conv2", r.cout, r.cout, 3)
            if r.cin != r.cout:
                conv(r.name + ".conv_shortcut", r.cin, r.cout, 1)
        for a in b.attns:
            C = a.ch
            tb = a.name + ".transformer_blocks.0"
            norm(a.name + ".norm", C)
            lin(a.name + ".proj_in", C, C)
            for k in ("norm1", "norm2", "norm3"):
                norm(f"{tb}.{k}", C)
            for an, kv in (("attn1", C), ("attn2", X)):
                lin(f"{tb}.{an}.to_q", C, C, bias=False)
                lin(f"{tb}.{an}.to_k", kv, C, bias=False)
                lin(f"{tb}.{an}.to_v", kv, C, bias=False)
                lin(f"{tb}.{an}.to_out.0", C, C)
            lin(f"{tb}.ff.net.0.proj", C, 8 * C)
            lin(f"{tb}.ff.net.2", 4 * C, C)
            lin(a.name + ".proj_out", C, C)
        if b.sampler:
            conv(b.sampler, b.sampler_ch, b.sampler_ch, 3)
    norm("conv_norm_out", c0)
    conv("conv_out", c0, cfg.out_channels, 3)
    return s


def init_params(cfg: UNetConfig, seed: int = 0, dtype=torch.float32, w_std: float = 0.02,
                beta_std: float = 0.1) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: N(0, w_std) conv/linear weights and biases*0.5, norm gamma=1,
    norm beta ~ N(0, beta_std) (non-zero so the App. B.1 term is exercised)."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    for name, shp in param_shapes(cfg).items():
        leaf = name.rsplit(".", 2)[-2]
        is_norm = leaf.startswith("norm") or leaf == "conv_norm_out"
        if is_norm and name.endswith(".weight"):
            t = torch.ones(shp)
        elif is_norm:
            t = torch.randn(shp, generator=g) * beta_std
        elif name.endswith(".weight"):
            t = torch.randn(shp, generator=g) * w_std
        else:
            t = torch.randn(shp, generator=g) * (0.5 * w_std)
        p[name] = t.to(dtype)
    return p


# ----------------------------------------------------------------------------------------------------------------
# gate primitives (gates.py:9-55)
# ----------------------------------------------------------------------------------------------------------------
def width_gate(x: torch.Tensor, gate_f: torch.Tensor) -> torch.Tensor:
    """VirtualGate/WidthGate.forward: x [B,C,*,*] (or [B,h,L,d]) times gate_f [Bg,W] expanded over C//W channels,
    gate batch tiled B//Bg times for CFG (gates.py:15-21)."""
    width = gate_f.shape[1]
    mask = gate_f.to(x.dtype).repeat_interleave(x.shape[1] // width, dim=1).unsqueeze(-1).unsqueeze(-1)
    if mask.shape[0] != x.shape[0]:
        mask = mask.repeat(x.shape[0] // mask.shape[0], 1, 1, 1)
    return mask.expand_as(x) * x


def linear_width_gate(x: torch.Tensor, gate_f: torch.Tensor) -> torch.Tensor:
    """LinearWidthGate.forward on [B,L,C] (gates.py:49-55)."""
    width = gate_f.shape[1]
    mask = gate_f.to(x.dtype).repeat_interleave(x.shape[-1] // width, dim=1).unsqueeze(1)
    if mask.shape[0] != x.shape[0]:
        mask = mask.repeat(x.shape[0] // mask.shape[0], 1, 1)
    return mask.expand_as(x) * x


def depth_gate(x_in: torch.Tensor, x_out: torch.Tensor, gate_f: torch.Tensor) -> torch.Tensor:
    """DepthGate.forward: (1-d)*x_in + d*x_out with d [Bg] (gates.py:36-42)."""
    mask = gate_f.to(x_out.dtype).reshape(-1, 1, 1, 1)
    if mask.shape[0] != x_out.shape[0]:
        mask = mask.repeat(x_out.shape[0] // mask.shape[0], 1, 1, 1)
    return (1 - mask) * x_in + mask * x_out


def hard(gate: torch.Tensor) -> torch.Tensor:
    """hard_concrete forward value (estimation_utils.py:67-75): >=0.5 -> 1 else 0."""
    return (gate >= 0.5).to(gate.dtype)


# ----------------------------------------------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int, dtype) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps.to(torch.float32)[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)  # flip_sin_to_cos
    return emb.to(dtype)


def resnet_forward(p, r: ResnetSpec, cfg: UNetConfig, x, temb, gates, mode: str):
    """blocks.py:293-371 / :482-584.  mode "gated": multiply by gate (reference UNet2DConditionModelGated);
    mode "pruned": physically slice weights by hard_concrete(gate) (prune(), blocks.py:424-465, 641-697)."""
    n = r.name
    gate = gates.get(n + ".gate")
    dgate = gates.get(n + ".depth_gate") if r.depth_gated else None
    x_in = x[:, :x.shape[1] - r.skip_dim] if (r.depth_gated and r.skip_dim) else x
    if mode == "pruned" and dgate is not None and float(hard(dgate.reshape(-1)[:1])) == 0.0:
        return x_in  # dropped (blocks.py:497-498)
    G = cfg.norm_num_groups
    h = F.group_norm(x, G, p[n + ".norm1.weight"], p[n + ".norm1.bias"], cfg.norm_eps)
    h = F.silu(h)
    w1, b1 = p[n + ".conv1.weight"], p[n + ".conv1.bias"]
    wt, bt = p[n + ".time_emb_proj.weight"], p[n + ".time_emb_proj.bias"]
    g2w, g2b = p[n + ".norm2.weight"], p[n + ".norm2.bias"]
    w2, b2 = p[n + ".conv2.weight"], p[n + ".conv2.bias"]
    G2 = G
    if mode == "pruned" and gate is not None:
        assert gate.shape[0] == 1
        keep = hard(gate)[0].bool().repeat_interleave(r.cout // G)
        w1, b1, wt, bt, g2w, g2b = w1[keep], b1[keep], wt[keep], bt[keep], g2w[keep], g2b[keep]
        w2 = w2[:, keep]
        G2 = int(hard(gate)[0].sum().item())
    h = F.conv2d(h, w1, b1, padding=1)
    t = F.linear(F.silu(temb), wt, bt)[:, :, None, None]
    h = h + t
    if mode == "gated" and gate is not None:
        h = width_gate(h, gate)
    h = F.group_norm(h, G2, g2w, g2b, cfg.norm_eps)
    h = F.silu(h)
    h = F.conv2d(h, w2, b2, padding=1)
    sc = x
    if r.cin != r.cout:
        sc = F.conv2d(x, p[n + ".conv_shortcut.weight"], p[n + ".conv_shortcut.bias"])
    out = sc + h  # output_scale_factor == 1
    if mode == "gated" and dgate is not None:
        out = depth_gate(x_in, out, dgate)
    return out


def attention_forward(p, prefix: str, heads: int, x, ctx, gate, mode: str):
    """HeadGatedAttnProcessor2.__call__ (blocks.py:194-280): q/k/v bias-free linears, per-head gate on q,k,v
    (:250-255), SDPA scale 1/sqrt(64) (:258-260), to_out[0] with bias; dropout 0, rescale 1."""
    B, L, C = x.shape
    kv = x if ctx is None else ctx
    wq, wk, wv = p[prefix + ".to_q.weight"], p[prefix + ".to_k.weight"], p[prefix + ".to_v.weight"]
    wo, bo = p[prefix + ".to_out.0.weight"], p[prefix + ".to_out.0.bias"]
    hd = C // heads
    nh = heads
    if mode == "pruned" and gate is not None:
        assert gate.shape[0] == 1
        keep = hard(gate)[0].bool().repeat_interleave(hd)
        wq, wk, wv, wo = wq[keep], wk[keep], wv[keep], wo[:, keep]
        nh = int(hard(gate)[0].sum().item())
    q = F.linear(x, wq).view(B, -1, nh, hd).transpose(1, 2)
    k = F.linear(kv, wk).view(B, -1, nh, hd).transpose(1, 2)
    v = F.linear(kv, wv).view(B, -1, nh, hd).transpose(1, 2)
    if mode == "gated" and gate is not None:
        q, k, v = width_gate(q, gate), width_gate(k, gate), width_gate(v, gate)
    o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, -1, nh * hd)
    return F.linear(o, wo, bo)


def ff_forward(p, prefix: str, x, gate, mode: str):
    """GEGLUGated.forward (blocks.py:41-50) + net[2] Linear; exact-erf GELU."""
    w0, b0 = p[prefix + ".net.0.proj.weight"], p[prefix + ".net.0.proj.bias"]
    w2, b2 = p[prefix + ".net.2.weight"], p[prefix + ".net.2.bias"]
    inner = w2.shape[1]
    if mode == "pruned" and gate is not None:
        assert gate.shape[0] == 1
        keep = hard(gate)[0].bool().repeat_interleave(inner // gate.shape[1])
        keep2 = torch.cat([keep, keep])
        w0, b0, w2 = w0[keep2], b0[keep2], w2[:, keep]
    hcat = F.linear(x, w0, b0)
    hs, g = hcat.chunk(2, dim=-1)
    if mode == "gated" and gate is not None:
        hs, g = linear_width_gate(hs, gate), linear_width_gate(g, gate)
    return F.linear(hs * F.gelu(g), w2, b2)


def transformer_forward(p, a: AttnSpec, cfg: UNetConfig, x, ctx, gates, mode: str):
    """Transformer2DModel(use_linear_projection) forward with one BasicTransformerBlockWidthGated
    (blocks.py:1139-1355, 763-851)."""
    n = a.name
    dgate = gates.get(n + ".depth_gate") if a.depth_gated else None
    if mode == "pruned" and dgate is not None and float(hard(dgate.reshape(-1)[:1])) == 0.0:
        return x  # dropped (blocks.py:1190-1194)
    B, C, H, W = x.shape
    tb = n + ".transformer_blocks.0"
    h = F.group_norm(x, cfg.norm_num_groups, p[n + ".norm.weight"], p[n + ".norm.bias"], 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = F.linear(h, p[n + ".proj_in.weight"], p[n + ".proj_in.bias"])
    nh = F.layer_norm(h, (C,), p[tb + ".norm1.weight"], p[tb + ".norm1.bias"], 1e-5)
    h = attention_forward(p, tb + ".attn1", a.heads, nh, None, gates.get(n + ".attn1.gate"), mode) + h
    nh = F.layer_norm(h, (C,), p[tb + ".norm2.weight"], p[tb + ".norm2.bias"], 1e-5)
    h = attention_forward(p, tb + ".attn2", a.heads, nh, ctx, gates.get(n + ".attn2.gate"), mode) + h
    nh = F.layer_norm(h, (C,), p[tb + ".norm3.weight"], p[tb + ".norm3.bias"], 1e-5)
    h = ff_forward(p, tb + ".ff", nh, gates.get(n + ".ff.gate"), mode) + h
    h = F.linear(h, p[n + ".proj_out.weight"], p[n + ".proj_out.bias"])
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    out = h + x
    if mode == "gated" and dgate is not None:
        out = depth_gate(x, out, dgate)
    return out


def unet_forward(p: Dict[str, torch.Tensor], cfg: UNetConfig, sample: torch.Tensor, timestep,
                 encoder_hidden_states: torch.Tensor, gates: Optional[Dict[str, torch.Tensor]] = None,
                 mode: str = "gated", return_blocks: bool = False):
    """UNet2DConditionModelGated.forward (unet_2d_conditional.py:1415-1726) for the SD-2.1 configuration.

    gates: output of ``assign_gates`` (missing entries == all-ones gate, the reference default gates.py:13).
    mode: "gated" (mask multiply, reference Gated model) or "pruned" (sliced weights, reference Pruned model).
    return_blocks: also return the outputs of down_blocks / mid_block / up_blocks as the trainer's forward hooks
    see them (trainer.py:496-511)."""
    assert mode in ("gated", "pruned")
    gates = gates or {}
    dt = sample.dtype
    B = sample.shape[0]
    t = timestep
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.int64)
    elif t.ndim == 0:
        t = t[None]
    t = t.expand(B)
    temb = timestep_embedding(t, cfg.block_out_channels[0], dt)
    temb = F.linear(temb, p["time_embedding.linear_1.weight"], p["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), p["time_embedding.linear_2.weight"], p["time_embedding.linear_2.bias"])
    ctx = encoder_hidden_states.to(dt)

    h = F.conv2d(sample, p["conv_in.weight"], p["conv_in.bias"], padding=1)
    skips = [h]
    block_outs = []
    for b in build_specs(cfg):
        if b.kind == "down":
            for j, r in enumerate(b.resnets):
                h = resnet_forward(p, r, cfg, h, temb, gates, mode)
                if b.attns:
                    h = transformer_forward(p, b.attns[j], cfg, h, ctx, gates, mode)
                skips.append(h)
            if b.sampler:
                h = F.conv2d(h, p[b.sampler + ".weight"], p[b.sampler + ".bias"], stride=2, padding=1)
                skips.append(h)
        elif b.kind == "mid":
            h = resnet_forward(p, b.resnets[0], cfg, h, temb, gates, mode)
            h = transformer_forward(p, b.attns[0], cfg, h, ctx, gates, mode)
            h = resnet_forward(p, b.resnets[1], cfg, h, temb, gates, mode)
        else:
            for j, r in enumerate(b.resnets):
                h = torch.cat([h, skips.pop()], dim=1)
                h = resnet_forward(p, r, cfg, h, temb, gates, mode)
                if b.attns:
                    h = transformer_forward(p, b.attns[j], cfg, h, ctx, gates, mode)
            if b.sampler:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = F.conv2d(h, p[b.sampler + ".weight"], p[b.sampler + ".bias"], padding=1)
        block_outs.append(h)
    h = F.group_norm(h, cfg.norm_num_groups, p["conv_norm_out.weight"], p["conv_norm_out.bias"], cfg.norm_eps)
    h = F.silu(h)
    out = F.conv2d(h, p["conv_out.weight"], p["conv_out.bias"], padding=1)
    if return_blocks:
        return out, block_outs
    return out


# ----------------------------------------------------------------------------------------------------------------
# canonical masks / inputs
# ----------------------------------------------------------------------------------------------------------------
def fixed_half_mask(cfg: UNetConfig, batch: int = 1) -> Dict[str, List[torch.Tensor]]:
    """The fixed 50 % mask of BASELINE config 2: 32-wide gates keep even entries, head gates keep the first
    floor(h/2) heads, all depth gates on; same mask for every sample."""
    st = get_structure(cfg)
    width = []
    for sub in st["width"]:
        for w in sub:
            g = torch.zeros(batch, w)
            if w == cfg.norm_num_groups or w == cfg.ff_gate_width:
                g[:, 0::2] = 1.0
            else:
                g[:, :max(1, w // 2)] = 1.0
            width.append(g)
    depth = [torch.ones(batch) for sub in st["depth"] for d in sub if d == 1]
    return {"width": width, "depth": depth}


def ones_mask(cfg: UNetConfig, batch: int = 1) -> Dict[str, List[torch.Tensor]]:
    st = get_structure(cfg)
    width = [torch.ones(batch, w) for sub in st["width"] for w in sub]
    depth = [torch.ones(batch) for sub in st["depth"] for d in sub if d == 1]
    return {"width": width, "depth": depth}


def random_mask(cfg: UNetConfig, keep: float, seed: int, n_depth_off: int = 0, batch: int = 1):
    """Seeded hard mask in the style of HyperStructure.get_random_arch_vector (hypernet.py:131-153)."""
    g = torch.Generator().manual_seed(seed)
    st = get_structure(cfg)
    width = []
    for sub in st["width"]:
        for w in sub:
            m = torch.zeros(batch, w)
            k = max(1, int(keep * w))
            for bi in range(batch):
                m[bi, torch.randperm(w, generator=g)[:k]] = 1.0
            width.append(m)
    nd = sum(d for sub in st["depth"] for d in sub)
    depth = [torch.ones(batch) for _ in range(nd)]
    off = torch.randperm(nd, generator=g)[:n_depth_off].tolist()
    for i in off:
        depth[i] = torch.zeros(batch)
    return {"width": width, "depth": depth}


def synthetic_inputs(cfg: UNetConfig, batch: int, latent: int, seed: int = 1234, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(batch, cfg.in_channels, latent, latent, generator=g).to(dtype)
    ehs = torch.randn(batch, 77, cfg.cross_attention_dim, generator=g).to(dtype)
    t = torch.full((batch,), 500, dtype=torch.int64)
    return sample, t, ehs


def count_macs(cfg: UNetConfig, latent: int, masked: bool = False) -> float:
    """Matmul-only MAC count per sample (SURVEY App. D convention); masked = the fixed 50 % mask."""
    total = 0.0
    H = latent
    res = {}
    cur = H
    total += 9 * cfg.in_channels * cfg.block_out_channels[0] * H * H
    level_res = []
    for i in range(cfg.n_levels):
        level_res.append(cur)
        if i != cfg.n_levels - 1:
            cur //= 2
    for b in build_specs(cfg):
        if b.kind == "down":
            li = int(b.name.split(".")[1]); r_ = level_res[li]
        elif b.kind == "mid":
            r_ = level_res[-1]
        else:
            li = int(b.name.split(".")[1]); r_ = level_res[cfg.n_levels - 1 - li]
        P = r_ * r_
        for r in b.resnets:
            c1o = r.cout // 2 if masked else r.cout
            total += 9 * r.cin * c1o * P + cfg.temb_dim * c1o + 9 * c1o * r.cout * P
            if r.cin != r.cout:
                total += r.cin * r.cout * P
        for a in b.attns:
            C, X = a.ch, cfg.cross_attention_dim
            hl = max(1, a.heads // 2) if masked else a.heads
            Cl = hl * 64
            total += 2 * C * C * P  # proj in/out
            total += 3 * C * Cl * P + Cl * C * P + 2 * hl * P * P * 64  # self
            total += C * Cl * P + 2 * X * Cl * 77 + Cl * C * P + 2 * hl * P * 77 * 64  # cross
            inner = 4 * C // 2 if masked else 4 * C
            total += C * 2 * inner * P + inner * C * P
        if b.sampler:
            if b.kind == "down":
                total += 9 * b.sampler_ch * b.sampler_ch * (r_ // 2) ** 2
            else:
                total += 9 * b.sampler_ch * b.sampler_ch * (r_ * 2) ** 2
    total += 9 * cfg.block_out_channels[0] * cfg.out_channels * H * H
    return total
