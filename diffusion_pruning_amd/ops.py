"""Tensor-level wrappers over the C ABI (include/aptp_hip.h).

Activations are bf16 channels-last tensors ``[B, H, W, C]`` whose last dim is contiguous; the pixel stride
``x.stride(2)`` may exceed C (a channel slice of a wider buffer), which is how skip-concats are consumed without
copies.  Token tensors ``[B, L, C]`` are passed as ``[B, L, 1, C]``.  All launches go to torch's current stream,
so the whole forward can be captured into a HIP graph with ``torch.cuda.graph``.
"""
from __future__ import annotations

import ctypes
import os
import threading
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import UnetEpilogueParams, UnetPrologueParams
from ._lib import (ACT_GEGLU, ACT_NONE, ACT_SILU, AttentionBwdParams, AttentionParams, ConvGemmParams, DepthLerpParams, GateBwdParams, WgradParams, FfTailParams, FoldRowsParams, PackDgradParams, MseParams,
                   GegluParams, GroupNormBwdParams, GroupNormParams, LayerNormBwdParams, LayerNormParams,
                   ColsumParams, LayerNormPgradParams)

BK = 64
# Storage type of activations and packed weights.  The HIP kernels exist for bf16 only (MFMA operands); the test-only CPU
# emulator (tests/hip_emulator.py) can switch both to fp32 to run the SAME host logic -- compaction, GroupNorm-beta
# correction, gate plumbing, batched time / text projections, in-place skip-concats, depth lerp -- without any rounding and
# compare it with the fp32 oracle at 1e-5 (tests/test_host_logic.py); the kernels themselves refuse anything but bf16.
ACT_DTYPE = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _check_act(x: torch.Tensor, name: str):
    # bf16 is the product's activation format; fp32 tensors take the fp32 PARITY instantiations of the kernels (io_f32 of the
    # parameter blocks: same code paths on exact-fp32 arithmetic, tests/test_fp32_parity_gpu.py; never benchmarked)
    if x.dtype not in (torch.bfloat16, torch.float32) or x.dim() != 4 or x.stride(3) != 1 or not x.is_cuda:
        raise ValueError(f"{name}: expected a CUDA bf16 (or, parity path, fp32) [B,H,W,C] tensor with contiguous channels, got "
                         f"{x.dtype} {tuple(x.shape)} strides {x.stride()}")
    B, H, W, C = x.shape
    ld = _ld(x)
    if (W > 1 and x.stride(2) != ld) or (H > 1 and x.stride(1) != W * ld) or (B > 1 and x.stride(0) != H * W * ld):
        raise ValueError(f"{name}: rows must be uniformly strided (ld={ld}), got strides {x.stride()}")
    if ld % 8 != 0 or x.data_ptr() % 16 != 0:
        raise ValueError(f"{name}: pixel stride {ld} must be a multiple of 8 elements and the base 16-byte aligned")
    return ld


def _ld(x: torch.Tensor) -> int:
    # pixel stride of a [B,H,W,C] view (robust to size-1 dims)
    B, H, W, C = x.shape
    if W > 1:
        return x.stride(2)
    if H > 1:
        return x.stride(1)
    if B > 1:
        return x.stride(0)
    return max(x.stride(2), C)


@dataclass
class PackedWeight:
    """bf16 weights in the kernel's [N][taps][cin_pad] layout + fp32 bias (both already gathered/compacted)."""
    w: torch.Tensor            # bf16 [N, KH*KW, cin_pad]
    bias: Optional[torch.Tensor]  # fp32 [N] or None
    N: int
    Cin: int
    KH: int
    KW: int
    geglu: bool = False
    ln_colsum: Optional[torch.Tensor] = None   # fp32 [N]: row sums of the bf16 weights when a LayerNorm is folded in
    Cin2: int = 0                              # second-operand K-segment (pack_weight_cat): channels, padded channels
    cin2_pad: int = 0
    cin_pad_: int = 0                          # cin_pad of the main segment when w is the flat [N, 1, Ktot] of a cat pack

    @property
    def cin_pad(self) -> int:
        return self.cin_pad_ or self.w.shape[2]


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def pack_weight(w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, out_idx: Optional[torch.Tensor] = None,
                in_idx: Optional[torch.Tensor] = None, n_pad_to: int = 8, cin_pad_to: int = 8,
                geglu: bool = False, device=None, ln_gamma: Optional[torch.Tensor] = None,
                ln_beta: Optional[torch.Tensor] = None) -> PackedWeight:
    """Pack an OIHW conv weight (or [out,in] linear weight) for aptp_conv_gemm.

    out_idx / in_idx: live output / input channel indices (architecture-code compaction; the reference's
    prune() slicing, blocks.py:436-463, 159-177, 55-64, 126).  Output rows and input columns are zero-padded to
    multiples of n_pad_to / cin_pad_to (pad rows produce exact zeros), then Cin to a multiple of 64.
    geglu: interleave the two halves of a GEGLU projection in [16 value | 16 gate] row blocks so the epilogue finds
    h and g of one hidden unit in the same lane.
    ln_gamma / ln_beta: fold the LayerNorm that precedes this linear layer into it (include/aptp_hip.h, ln_stats):
    w' = w * gamma, bias' = bias + w @ beta; PackedWeight.ln_colsum = row sums of the bf16-rounded w'."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    device = device or w.device
    w = w.to(device=device, dtype=torch.float32)
    if bias is not None:
        bias = bias.to(device=device, dtype=torch.float32)
    if ln_gamma is not None:
        assert w.shape[2] == 1 and w.shape[3] == 1 and in_idx is None, "a LayerNorm folds into a linear layer over all its channels"
        ga = ln_gamma.to(device=device, dtype=torch.float32)
        be = ln_beta.to(device=device, dtype=torch.float32)
        shift = w[:, :, 0, 0] @ be
        bias = shift if bias is None else bias + shift
        w = w * ga[None, :, None, None]
    if geglu:
        inner = w.shape[0] // 2
        wh, wg = w[:inner], w[inner:]
        bh, bg = (bias[:inner], bias[inner:]) if bias is not None else (None, None)
        if out_idx is not None:
            out_idx = out_idx.to(device)
            wh, wg = wh[out_idx], wg[out_idx]
            if bias is not None:
                bh, bg = bh[out_idx], bg[out_idx]
        if in_idx is not None:
            in_idx = in_idx.to(device)
            wh, wg = wh[:, in_idx], wg[:, in_idx]
        n_live = wh.shape[0]
        n_p = round_up(n_live, 16)

        def padrows(t):
            if t.shape[0] == n_p:
                return t
            return torch.cat([t, t.new_zeros((n_p - t.shape[0],) + tuple(t.shape[1:]))], 0)
        wh, wg = padrows(wh), padrows(wg)
        # [n_p/16, 16, ...] blocks interleaved
        w = torch.stack([wh.reshape(n_p // 16, 16, *wh.shape[1:]), wg.reshape(n_p // 16, 16, *wg.shape[1:])], 1)
        w = w.reshape(2 * n_p, *wh.shape[1:])
        if bias is not None:
            bh, bg = padrows(bh), padrows(bg)
            bias = torch.stack([bh.reshape(n_p // 16, 16), bg.reshape(n_p // 16, 16)], 1).reshape(2 * n_p)
    else:
        if out_idx is not None:
            out_idx = out_idx.to(device)
            w = w[out_idx]
            if bias is not None:
                bias = bias[out_idx]
        if in_idx is not None:
            w = w[:, in_idx.to(device)]
        n_live = w.shape[0]
        n_p = round_up(n_live, n_pad_to)
        if n_p != n_live:
            w = torch.cat([w, w.new_zeros((n_p - n_live,) + tuple(w.shape[1:]))], 0)
            if bias is not None:
                bias = torch.cat([bias, bias.new_zeros(n_p - n_live)], 0)
    N, Cin, KH, KW = w.shape
    cin_live = round_up(Cin, cin_pad_to)
    cin_pad = round_up(cin_live, BK)
    packed = torch.zeros(N, KH * KW, cin_pad, dtype=ACT_DTYPE, device=device)
    packed[:, :, :Cin] = w.permute(0, 2, 3, 1).reshape(N, KH * KW, Cin).to(ACT_DTYPE)
    colsum = packed.float().sum(dim=(1, 2)).contiguous() if ln_gamma is not None else None
    return PackedWeight(packed.contiguous(), None if bias is None else bias.contiguous(), N, cin_live, KH, KW, geglu, colsum)


def pack_weight_cat(pw: PackedWeight, w2: torch.Tensor, bias2: Optional[torch.Tensor] = None, *,
                    in_idx: Optional[torch.Tensor] = None) -> PackedWeight:
    """Append a second-operand K-segment (include/aptp_hip.h, x2) to packed conv weights: w2 [N_logical, Cin2] is the 1x1
    convolution applied to x2 at the output pixel (the resnet's conv_shortcut fused into conv2), bias2 is added to the
    bias.  Rows follow pw's rows (w2 is zero-padded to pw.N rows)."""
    assert not pw.geglu and pw.ln_colsum is None and pw.Cin2 == 0
    dev = pw.w.device
    w2 = w2.to(device=dev, dtype=torch.float32).reshape(w2.shape[0], -1)
    if in_idx is not None:
        w2 = w2[:, in_idx.to(dev)]
    n_live, c2 = w2.shape
    assert n_live <= pw.N
    c2_live = round_up(c2, 8)
    c2_pad = round_up(c2_live, BK)
    ext = torch.zeros(pw.N, c2_pad, dtype=ACT_DTYPE, device=dev)
    ext[:n_live, :c2] = w2.to(ACT_DTYPE)
    flat = torch.cat([pw.w.reshape(pw.N, -1), ext], 1).reshape(pw.N, 1, -1).contiguous()
    bias = pw.bias
    if bias2 is not None:
        b2 = torch.zeros(pw.N, dtype=torch.float32, device=dev)
        b2[:n_live] = bias2.to(device=dev, dtype=torch.float32)
        bias = b2 if bias is None else bias + b2
    return PackedWeight(flat, bias, pw.N, pw.Cin, pw.KH, pw.KW, False, None, c2_live, c2_pad, pw.cin_pad)


_ws_cache = {}
_ws_capture_keep = []      # see _workspace

def tuning_key(M, N, Cin, taps, stride, ups, geglu, Cin2: int = 0) -> str:
    return f"M{M}_N{N}_C{Cin}_T{taps}_s{stride}u{ups}g{int(bool(geglu))}" + (f"x{Cin2}" if Cin2 else "")


def _load_tuning():
    import json
    import os
    # (APTP_TUNING=<file> substitutes another table: A/B timing of tuning runs)
    path = os.environ.get("APTP_TUNING") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning_gfx950.json")
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


# (tile, split_k) per GEMM shape measured on MI355X by tools/tune_convs.py.  The table holds the shapes of the headline
# mask, the dense model and the pruning step; every OTHER architecture code (config 5's experts, real APTP experts with
# irregular widths) produces shapes that are not in it.  Those take the entry of the NEAREST tuned shape of the same class
# (taps, stride, upsampling, GEGLU, second operand) in log-space distance over (M, N, K) -- the tile engine's behaviour
# changes smoothly with the extents, while the library heuristic only knows the generic non-DMA tiles (measured on a
# 55 %-keep expert: 29 us per launch on conv_gemm_kernel<64,128> where the neighbours' DMA tiles take ~20).  Beyond
# TUNING_MAX_DIST the heuristic (aptp_conv_gemm_suggest_split_k / pick_tile) decides.  APTP_TUNING_NEAREST=0 disables.
TUNING = _load_tuning()
TUNING_NEAREST = os.environ.get("APTP_TUNING_NEAREST", "1") != "0"
TUNING_MAX_DIST = 2.0
_HALO_TILES = (43, 44)
_LEAN_TILES = (9, 11, 12, 15, 17, 18, 24, 25, 26, 45, 46, 47, 48, 49, 50, 51, 52, 53)     # tiles csrc/lin_gemm.hip instantiates (aptp_lin_eligible)
LEAN_REMAP = os.environ.get("APTP_LEAN_REMAP", "1") != "0"
SK_TILE_FIRST = 64          # APTP_TILE_SK_*: persistent stream-K macro-tiles (csrc/conv_gemm_sk.hip)
SK_AUTO = os.environ.get("APTP_SK_AUTO", "1") == "1"
SK_AUTO_MIN_OUTPUTS = 256 * 256 * 160
_tuning_index = None
_tuning_near_cache = {}


def _tuning_classes():
    global _tuning_index
    if _tuning_index is None:
        import re
        idx = {}
        pat = re.compile(r"M(\d+)_N(\d+)_C(\d+)_T(\d+)_s(\d+)u(\d+)g(\d+)(?:x(\d+))?$")
        for k, v in TUNING.items():
            m = pat.match(k)
            if not m:
                continue
            M, N, C, T, s_, u, g, x2 = (int(t) if t else 0 for t in m.groups())
            idx.setdefault((T, s_, u, g, x2 > 0), []).append((M, N, T * C + x2, v))
        _tuning_index = idx
    return _tuning_index


def tuning_lookup(M, N, Cin, taps, stride, ups, geglu, Cin2: int = 0):
    """exact table entry, else the nearest tuned shape of the same class (None when there is none close enough)"""
    key = tuning_key(M, N, Cin, taps, stride, ups, geglu, Cin2)
    hit = TUNING.get(key)
    if hit is not None or not TUNING_NEAREST:
        return hit
    if key in _tuning_near_cache:
        return _tuning_near_cache[key]
    import math
    K = taps * Cin + Cin2
    best, bd = None, TUNING_MAX_DIST
    for (M2, N2, K2, v) in _tuning_classes().get((taps, stride, ups, int(bool(geglu)), Cin2 > 0), ()):
        if v["tile"] in _HALO_TILES and M2 != M:
            continue                      # the halo-in-LDS tiles are tied to the map width
        if v["tile"] > 6 and max(Cin, Cin2) > 4032:
            continue                      # every tile past the six register-staged ones is an LDS-DMA tile: channel steps <= 4032 (csrc)
        # (more rows than the tuned shape is the benign direction -- the same tile, more of them: a U-Net batch of 16 takes the
        #  entries tuned at batch 4 instead of falling back to the register-staged heuristic tiles, which bench.py's infer_bs16 leg
        #  measured at 0.08 of the MFMA peak on 48 launches)
        dm = math.log2(M / M2)
        d = (0.5 * dm if dm > 0 else -1.5 * dm) + abs(math.log2(N / N2)) + abs(math.log2(K / K2))
        if d < bd:
            best, bd = v, d
    if best is not None:
        nK = (K + BK - 1) // BK
        best = dict(best)
        best["split_k"] = max(1, min(int(best["split_k"]), nK // 4 if nK >= 8 else 1))
    _tuning_near_cache[key] = best
    return best

# Optional launch recorder used by bench.py's roofline leg: when a list, every aptp_conv_gemm launch appends
# {"params": ConvGemmParams, "flops": algorithmic FLOPs, "keep": tensors referenced by the params}.
LAUNCH_LOG = None
# Same for aptp_groupnorm calls (bench.py's HBM roofline leg): {"params", "bytes": x read once + y written once, "keep"}
GN_LAUNCH_LOG = None
ATTN_LAUNCH_LOG = None      # bench.py: the attention launches of ONE forward (re-timed for roofline_attention)

# Split-K launches combine their K-slices inside the kernel (AptpConvGemmParams.tile_counters) instead of launching
# splitk_reduce_kernel; False restores the two-launch form (A/B timing, tests of both forms)
SPLITK_IN_KERNEL = True
SPLITK_FORCE_IN_KERNEL = os.environ.get("APTP_SPLITK_FORCE_INKERNEL", "0") == "1"
_counters = {}
_N_COUNTERS = 1 << 16


_N_SLABS = int(os.environ.get("APTP_N_SLABS", "16"))
_counters_lock = threading.Lock()


class scratch_domain:
    """``with ops.scratch_domain("teacher"): ...`` -- launches issued (or CAPTURED) inside use their own split-K counters and
    scratch buffers.  Scratch is keyed by the launching stream, which is enough for eager concurrency; but every
    ``torch.cuda.graph`` capture runs on the same capture stream, so two graphs that will REPLAY concurrently on different
    streams (the pruning step's teacher next to the student forward) must be captured under different domains."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = getattr(_tls, "domain", None)
        _tls.domain = self.name
        return self

    def __exit__(self, *exc):
        _tls.domain = self.prev
        return False


def _domain():
    return getattr(_tls, "domain", None)


def _tile_counters(device):
    """Zero-initialised arrival counters of the split-K launches issued on the CURRENT stream (every launch leaves them
    zero).  One pool of _N_SLABS slabs per device, created on first use outside stream capture (a fill recorded into a
    graph would only run at replay); each stream that launches split-K work is handed its own slab on first use -- a
    host-side table lookup, legal during capture -- so forwards running concurrently on different streams never share
    counters.  Returns None (= use the separate reduce launch) when the pool cannot be created yet or is exhausted."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    pool = _counters.get(key)
    if pool is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        with _counters_lock:
            pool = _counters.get(key)
            if pool is None:
                pool = {"buf": torch.zeros(_N_SLABS, _N_COUNTERS, dtype=torch.int32, device=device), "slab": {}}
                _counters[key] = pool
    sid = (torch.cuda.current_stream().cuda_stream, _domain())
    i = pool["slab"].get(sid)
    if i is None:
        with _counters_lock:
            i = pool["slab"].get(sid)
            if i is None:
                if len(pool["slab"]) >= _N_SLABS:
                    return None
                i = pool["slab"][sid] = len(pool["slab"])
    return pool["buf"][i]


# AptpGroupNormParams.variant used when the caller asks for "auto" (0).  0 = the library's own choice (small maps in one
# launch, large maps in three).  3 (large maps in two launches: <= 16 coarse statistics chunks folded by the apply
# workgroups) measured much SLOWER on MI355X (149.5 vs 167.4 steps/s: 64 statistics workgroups cannot stream a level-64
# map fast enough), so it stays opt-in.
GN_DEFAULT_VARIANT = 0
GN_STATS_ONE_LAUNCH = True     # keep_stats on the small maps through the one-launch kernel (False: always three launches; A/B, tests)

# AptpAttentionParams.variant for every launch (1 = staggered wave groups: A/B timing, tests of both forms)
ATTN_VARIANT = int(os.environ.get("APTP_ATTN_VARIANT", "0"))      # AptpAttentionParams.variant: 0 = the library chooses (A/B timing, tests)

# GroupNorm on the large maps: True lets the last statistics workgroup of a sample finalise mean / rstd (two launches
# instead of three).  Measured SLOWER on MI355X (162.5 vs 164.7 steps/s: 128 tickets on one counter + the acquire cost
# more than the 5.5 us launch they replace), so it is off; kept for tests and as a measured negative result.
GN_FUSED_FINALIZE = False

# AptpConvGemmParams.epilogue for every launch: 0 = auto (coalesced epilogue where alignment allows), 1 = force the
# accumulator-layout epilogue (A/B timing, tests of both forms)
EPILOGUE = 0


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): stream-ordered reuse, kernels on one stream serialise; forwards that
    run concurrently on different streams must not share split-K slabs or GroupNorm partials."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.is_current_stream_capturing(),
           torch.cuda.current_stream().cuda_stream, _domain())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        size = max(nbytes, 1 << 20)
        if key[1] and buf is not None:
            size = max(size, buf.numel() + buf.numel() // 2)      # (kept forever, below: grow geometrically so the kept set stays small)
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        if key[1]:
            # Allocated while a stream is capturing: the block comes from that graph's private pool and its address is baked into
            # every launch captured with it -- possibly into graphs captured LATER on the same stream, which may outlive the graph
            # that owns the pool.  torch.cuda.graph.__enter__ empties the cache, so once the owning graph is gone and a larger
            # request has replaced the buffer here, the block would be returned to the device while a younger graph still
            # writes to it (seen as "Memory access fault ... write access to a read-only page" when a code object was mapped
            # there, and only for some test orders).  A capture-time scratch buffer is therefore never released: a few grow
            # steps per (stream, domain), bounded by the largest request.
            _ws_capture_keep.append(buf)
        _ws_cache[key] = buf
    return buf


# GroupNorm statistics emitted by the GEMM that produces a tensor (AptpConvGemmParams.colstat_out) travel with the
# tensor: they are recorded on its root storage owner (the tensor itself, or the concat buffer it is a channel slice of)
# under the view's storage offset, so the consumer finds them through any permute / reshape view, and the two halves of
# a skip-concat buffer keep separate records.  COLSTATS_MIN_HW: only maps large enough for GroupNorm's multi-launch form.
COLSTATS = True
COLSTATS_MIN_HW = 1024


def _root(t: torch.Tensor) -> torch.Tensor:
    return t._base if t._base is not None else t


# Round 4: GroupNorm statistics WITHOUT a finalise launch.  A producer that emits column statistics also adds per-(sample, channel
# unit) fixed-point sums to a slice of a zeroed arena (AptpConvGemmParams.ustat_out: 64-bit integer atomics, order-independent);
# when every producer of a GroupNorm's input left them, the apply pass finishes mean / rstd itself (AptpGroupNormColStats.ustats).
# The arena belongs to ONE forward: the model calls ustat_begin() at the top of its forward (that zeroes it: one fill launch, also
# inside a captured graph) and ustat_end() at the end; outside such a bracket nothing is emitted.  USTAT_UNIT = channels per unit:
# it must divide every GroupNorm group size of the model (SD-2.1: 320 / 32 = 10); 0 = off.
USTAT = os.environ.get("APTP_USTAT", "1") != "0"
# a split-K launch whose output feeds a GroupNorm combines its slices in-kernel (and emits the statistics) up to this many slices
COLS_SPLIT_MAX = int(os.environ.get("APTP_COLS_SPLIT_MAX", "2"))
USTAT_NREP = int(os.environ.get("APTP_USTAT_NREP", "8"))
_USTAT_WORDS = 1 << 17            # int64 words per arena (1 MiB): ~35 producers x 8 replicas x 4 samples x <= 128 units x 2
_ustat_arenas = {}


def ustat_begin(device, unit: int):
    """open the unit-statistics arena of the forward that starts now on the current stream (zeroed here); unit <= 0 or
    APTP_USTAT=0: no arena, producers emit column statistics only"""
    if not USTAT or unit <= 0 or torch.device(device).type != "cuda":
        _tls.ustat = None
        return
    capturing = torch.cuda.is_current_stream_capturing()
    key = (torch.device(device).index, torch.cuda.current_stream().cuda_stream, _domain(), capturing)
    ent = _ustat_arenas.get(key)
    if ent is None:
        # (allocated inside a capture, the zero fill is part of the graph: every replay starts from a clean arena; such a buffer is
        #  never released -- its address is baked into the graph, see _workspace)
        ent = _ustat_arenas[key] = {"buf": torch.zeros(_USTAT_WORDS, dtype=torch.int64, device=device), "used": 0}
        if capturing:
            _ws_capture_keep.append(ent["buf"])
            ent["used"] = _USTAT_WORDS          # a second capture on this stream zeroes everything: it cannot know what the first used
    elif ent["used"]:
        ent["buf"][:ent["used"]].zero_()        # what the previous forward on this arena added to
    _tls.ustat = {"buf": ent["buf"], "off": 0, "unit": int(unit), "ent": ent}


def ustat_end():
    st = getattr(_tls, "ustat", None)
    if st is not None and not st.get("closed"):
        st["closed"] = True
        st["ent"]["used"] = max(st["ent"]["used"], st["off"])


def _ustat_alloc(B: int, nout: int):
    st = getattr(_tls, "ustat", None)
    if st is None or st.get("closed"):
        return None
    unit = st["unit"]
    units = (nout + unit - 1) // unit
    nrep = USTAT_NREP                 # replicas spread the same-address contention; more samples already spread it
    while nrep > 1 and nrep * B > 4 * USTAT_NREP:
        nrep //= 2
    n = nrep * B * units * 2
    if st["off"] + n > st["buf"].numel():
        return None
    u = st["buf"][st["off"]:st["off"] + n]
    st["off"] += n
    return u, unit, units, nrep


def _colstats_drop(out: torch.Tensor):
    d = getattr(_root(out), "_aptp_colstats", None)
    if d:
        d.pop(out.storage_offset(), None)


def _colstats_put(out: torch.Tensor, stats: torch.Tensor, rows_per_block: int, ustat=None):
    root = _root(out)
    d = getattr(root, "_aptp_colstats", None)
    if d is None:
        d = {}
        root._aptp_colstats = d
    d[out.storage_offset()] = (stats, rows_per_block, out.shape[3], (out.shape[0], out.shape[1] * out.shape[2]), _ld(out), ustat)


def _colstats_get(x: torch.Tensor, C: int):
    """producer statistics covering the NHWC view x (C real channels, the rest zero padding): one record, or two adjacent
    ones (skip-concat); returns [(stats, rows_per_block, channels used), ...] or None"""
    d = getattr(_root(x), "_aptp_colstats", None)
    if not d:
        return None
    segs, off, left, pad = [], x.storage_offset(), x.shape[3], x.shape[3] - C
    while left > 0 and len(segs) < 2:
        rec = d.get(off)
        if rec is None or rec[3] != (x.shape[0], x.shape[1] * x.shape[2]) or rec[4] != _ld(x) or rec[2] > left:
            return None
        segs.append([rec[0], rec[1], rec[2], rec[5] if len(rec) > 5 else None])
        off += rec[2]
        left -= rec[2]
    if left != 0 or segs[-1][2] <= pad:
        return None
    segs[-1][2] -= pad
    return segs


class _PrefetchPlan:
    """Order in which one forward reads its packed weights.  The first forward records it; later forwards hand launch i
    the weights of launch i+1 as AptpConvGemmParams.prefetch.  A forward whose order differs from the recorded one
    (another batch of expert masks, other packs) re-records from the point of divergence.
    (Tried and dropped: the same hook in the GroupNorm launch that sits between two convolutions, as eight extra
    prefetch-only workgroups per sample -- 183.7-184.5 vs 184.8 steps/s without it.)"""

    def __init__(self):
        self.seq = []
        self.i = 0

    def begin(self):
        self.i = 0

    def step(self, w):
        i, self.i = self.i, self.i + 1
        if i < len(self.seq) and self.seq[i] is w:
            nxt = self.seq[i + 1] if i + 1 < len(self.seq) else None
        else:
            del self.seq[i:]
            self.seq.append(w)
            nxt = None
        if nxt is not None and not (PREFETCH_MIN_BYTES <= nxt.numel() * nxt.element_size() <= PREFETCH_MAX_BYTES):
            return None
        return nxt


# The plan of the forward in progress is per THREAD (a model installs its own at the top of forward(), see unet.py):
# forwards issued from different host threads -- e.g. one per stream -- never see each other's launch order.
_tls = threading.local()


def set_prefetch_plan(plan: Optional[_PrefetchPlan]):
    _tls.prefetch_plan = plan
    if plan is not None:
        plan.begin()


def _current_prefetch_plan() -> Optional[_PrefetchPlan]:
    return getattr(_tls, "prefetch_plan", None)


# Next-launch weight prefetch (AptpConvGemmParams.prefetch): on for weights that fit the XCD L2s next to the running
# launch's own traffic (<= 12 MB: +1.5 % on the headline forward; all sizes +0.5 %; off: APTP_PREFETCH=0)
PREFETCH_WEIGHTS = os.environ.get("APTP_PREFETCH", "1") == "1"
PREFETCH_MIN_BYTES = int(os.environ.get("APTP_PREFETCH_MIN", "0"))
PREFETCH_MAX_BYTES = int(os.environ.get("APTP_PREFETCH_MAX", str(12 << 20)))


def conv_gemm(x: torch.Tensor, pw: PackedWeight, *, stride: int = 1, pad: Optional[int] = None, ups: int = 0,
              out: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None,
              colgate: Optional[torch.Tensor] = None, gate_group: int = 0, act: int = ACT_NONE,
              corr: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
              depth: Optional[torch.Tensor] = None, depth_in: Optional[torch.Tensor] = None,
              out_f32: bool = False, split_k: Optional[int] = None, tile: int = 0, order: int = 0,
              rowstats: bool = False, ln=None, colstats: bool = False, x2: Optional[torch.Tensor] = None,
              prefetch=None, gn=None):
    """y = epilogue(conv(x, w)); see include/aptp_hip.h for the epilogue order and the reference call sites.
    gn = (gamma, beta, groups, eps, silu, C): return GroupNorm(+SiLU) of the convolution output over its first C channels
    instead of the output itself.  On the small maps, where the launch is split along K anyway, the reduce launch of the
    split applies the normalisation (AptpConvGemmParams.gn_gamma: one launch and one round trip less); otherwise this is
    conv_gemm followed by ops.groupnorm.
    rowstats: also emit the per-row (sum, sumsq) partials of y a following folded LayerNorm needs; returns (y, stats)
    with stats fp32 [slots / 2, M, 4] (two (sum, sumsq) slots per element), or (y, None) when this launch is split along K (the caller then normalises with
    ops.layernorm).  ln = (stats, eps): x is the un-normalised input of a LayerNorm folded into pw (pack_weight ln_gamma)."""
    lib = _lib.load()
    if gn is not None and (colgate is not None or act != ACT_NONE or pw.geglu or corr is not None or residual is not None
                           or depth is not None or out_f32 or rowstats or ln is not None or out is not None):
        raise ValueError("conv_gemm: gn= goes with a plain bf16 convolution (bias / rowbias only)")
    _check_act(x, "conv_gemm x")
    B, Hin, Win, Cx = x.shape
    if Cx != pw.Cin:
        raise ValueError(f"conv_gemm: x has {Cx} channels, packed weight expects {pw.Cin}")
    if pad is None:
        pad = pw.KH // 2
    sh = 1 if ups else 0            # ups 1 = nearest x2, ups 2 = zero-insertion x2 (dgrad of a stride-2 conv)
    HinE, WinE = Hin << sh, Win << sh
    Hout = (HinE + 2 * pad - pw.KH) // stride + 1
    Wout = (WinE + 2 * pad - pw.KW) // stride + 1
    if pw.geglu:
        act = ACT_GEGLU
    nout = pw.N // 2 if act == ACT_GEGLU else pw.N
    f32 = x.dtype == torch.float32                # fp32 parity instantiation: every tensor operand fp32, tiles 1..6, no tables
    gn_f32 = None
    if f32:
        if pw.w.dtype != torch.float32:
            raise ValueError("conv_gemm: fp32 activations need weights packed under ops.ACT_DTYPE = torch.float32")
        out_f32, colstats, prefetch, gn_f32, gn = True, False, False, gn, None
        if tile not in (0, 1, 2, 3, 4, 5, 6):
            raise ValueError("conv_gemm: the fp32 parity path runs on the register-staged tiles 1..6")
    if out is None:
        out = torch.empty(B, Hout, Wout, nout, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    else:
        _colstats_drop(out)          # whatever statistics were recorded for this memory describe its old contents
        if tuple(out.shape) != (B, Hout, Wout, nout) or out.stride(3) != 1:
            raise ValueError(f"conv_gemm: out shape {tuple(out.shape)} != {(B, Hout, Wout, nout)}")
        if out.dtype != (torch.float32 if out_f32 else torch.bfloat16):
            raise ValueError("conv_gemm: out dtype mismatch")
    p = ConvGemmParams()
    p.x, p.ldx = x.data_ptr(), _ld(x)
    p.B, p.Hin, p.Win, p.Cin, p.Hout, p.Wout = B, Hin, Win, Cx, Hout, Wout
    p.KH, p.KW, p.stride, p.pad, p.ups = pw.KH, pw.KW, stride, pad, ups
    p.w, p.N, p.cin_pad = pw.w.data_ptr(), pw.N, pw.cin_pad
    if (x2 is not None) != (pw.Cin2 > 0):
        raise ValueError("conv_gemm: x2 goes with weights packed by pack_weight_cat (and only with them)")
    if x2 is not None:
        _check_act(x2, "conv_gemm x2")
        if tuple(x2.shape) != (B, Hout, Wout, pw.Cin2):
            raise ValueError(f"conv_gemm: x2 shape {tuple(x2.shape)} != {(B, Hout, Wout, pw.Cin2)}")
        p.x2, p.ldx2, p.Cin2, p.cin2_pad = x2.data_ptr(), _ld(x2), pw.Cin2, pw.cin2_pad
    p.bias = None if pw.bias is None else pw.bias.data_ptr()
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.dim() == 2 and rowbias.shape[0] == B and rowbias.stride(1) == 1
        assert rowbias.shape[1] >= pw.N
        p.rowbias, p.ld_rowbias = rowbias.data_ptr(), rowbias.stride(0)
    if colgate is not None:
        assert colgate.dtype == torch.float32 and colgate.dim() == 2 and colgate.is_contiguous()
        assert gate_group > 0 and colgate.shape[1] * gate_group == nout and B % colgate.shape[0] == 0
        p.colgate, p.gate_group, p.gate_B = colgate.data_ptr(), gate_group, colgate.shape[0]
    p.act = act
    if corr is not None:
        assert corr.dtype == torch.float32 and corr.is_contiguous() and corr.shape[1:] == (9, nout)
        assert B % corr.shape[0] == 0
        p.corr, p.corr_B = corr.data_ptr(), corr.shape[0]
    if residual is not None:
        _check_act(residual, "conv_gemm residual")
        assert tuple(residual.shape) == (B, Hout, Wout, nout)
        p.residual, p.ldres = residual.data_ptr(), _ld(residual)
    if depth is not None:
        assert depth.dtype == torch.float32 and depth.dim() == 1 and B % depth.shape[0] == 0 and depth_in is not None
        _check_act(depth_in, "conv_gemm depth_in")
        assert tuple(depth_in.shape) == (B, Hout, Wout, nout)
        p.depth, p.depth_B = depth.data_ptr(), depth.shape[0]
        p.depth_in, p.lddin = depth_in.data_ptr(), _ld(depth_in)
    p.y, p.ldy, p.out_f32 = out.data_ptr(), _ld(out), int(out_f32)
    p.tile = tile
    p.order = order
    p.epilogue = EPILOGUE
    p.io_f32 = int(f32)
    if f32:
        for t_, nm in ((x2, "x2"), (residual, "residual"), (depth_in, "depth_in")):
            if t_ is not None and t_.dtype != torch.float32:
                raise ValueError(f"conv_gemm: fp32 parity path: {nm} must be fp32 too")
    if prefetch is False:             # the "weights" of this launch are a temporary (conv_wgrad): not part of any plan
        prefetch = None
    elif PREFETCH_WEIGHTS and prefetch is None:
        plan = _current_prefetch_plan()
        if plan is not None:
            prefetch = plan.step(pw.w)
    if prefetch is not None:
        p.prefetch, p.prefetch_bytes = prefetch.data_ptr(), prefetch.numel() * prefetch.element_size()
    p.split_k = 1
    in_kernel = None                 # split-K form: tuned per shape; untuned shapes combine in-kernel up to 4 slices
    explicit_split = split_k is not None
    if split_k is None and tile == 0 and not f32:
        tuned = tuning_lookup(B * Hout * Wout, pw.N, Cx, pw.KH * pw.KW, stride, ups, act == ACT_GEGLU, pw.Cin2)
        if tuned is not None and SK_AUTO and B * Hout * Wout * pw.N >= SK_AUTO_MIN_OUTPUTS \
                and pw.KH * pw.KW * (pw.cin_pad // BK) + pw.cin2_pad // BK >= 16 \
                and tuning_key(B * Hout * Wout, pw.N, Cx, pw.KH * pw.KW, stride, ups, act == ACT_GEGLU, pw.Cin2) not in TUNING:
            tuned = None                   # only a NEIGHBOUR's entry, and the shape is in the stream-K macro-tiles' range (below)
        if tuned is not None and tuned["tile"] >= 7 and max(pw.cin_pad, pw.cin2_pad) > (32704 if tuned["tile"] >= SK_TILE_FIRST else 4032):
            tuned = None                   # the LDS-DMA tiles address at most 4032 channels per tap (a neighbour's tile may be one)
        if tuned is not None:
            p.tile, split_k = tuned["tile"], tuned["split_k"]
            in_kernel = bool(tuned.get("in_kernel", 0))
            if LEAN_REMAP and p.tile not in _LEAN_TILES and split_k == 1 and "insitu" not in tuned and pw.KH == 1 and pw.KW == 1 and pw.cin_pad <= 4032 \
                    and stride == 1 and not ups and x2 is None and colgate is None and corr is None and rowbias is None \
                    and depth is None and not out_f32 and gn is None and (act != ACT_GEGLU or (residual is None and not rowstats)):
                # a plain linear layer whose table entry (tuned before csrc/lin_gemm.hip existed: the training steps' shapes) names a
                # tile the lean kernel has no instantiation of -- the 160-wide and the intra-workgroup K-split tiles: take the lean
                # tile the in-situ pass over the inference forward preferred at this row count (profiles/r4_tune_insitu_lean.txt)
                M_ = B * Hout * Wout
                p.tile = 11 if M_ >= 8192 else (18 if M_ >= 2048 else 49)
            if order == 0:
                p.order = tuned.get("order", 1)     # tables tuned before the XCD-aware orders existed mean the legacy order
    if split_k is None and tile == 0 and p.tile == 0 and LEAN_REMAP and not f32 and pw.KH == 1 and pw.KW == 1 and stride == 1 and not ups \
            and x2 is None and colgate is None and corr is None and rowbias is None and depth is None and not out_f32 and gn is None \
            and (act != ACT_GEGLU or (residual is None and not rowstats)) and pw.cin_pad <= 4032:
        # an untuned plain linear layer: a lean tile by row count (csrc/lin_gemm.hip) instead of the library's register-staged pick
        M_ = B * Hout * Wout
        p.tile, split_k = (11 if M_ >= 8192 else (18 if M_ >= 2048 else 49)), 1
        if M_ <= 512 and pw.cin_pad >= 1280:
            p.tile, split_k = 0, None          # (tiny M with a long K wants a K split: the library heuristic decides)
    if split_k is None and tile == 0 and p.tile == 0 and SK_AUTO and not f32:
        # no table entry: contractions with >= SK_AUTO_MIN_OUTPUTS outputs (a chip-filling number of 256 x 160 macro-tiles) and
        # a long K take the persistent stream-K macro-tiles -- measured 1.28-1.42x the best per-tile launch on such shapes
        # (tools/bench_sk.py: 16384 x 1280 x 11520 at 1,093 TFLOP/s; 8192^3 at 995 against 744) -- and nothing below that
        # size does (0.5-0.98x on every launch of the bs=4 forward, which is why the table holds none)
        nK = pw.KH * pw.KW * (pw.cin_pad // BK) + pw.cin2_pad // BK
        if B * Hout * Wout * pw.N >= SK_AUTO_MIN_OUTPUTS and nK >= 16 and max(pw.cin_pad, pw.cin2_pad) <= 32704:
            p.tile = SK_TILE_FIRST + 1 if act == ACT_GEGLU else SK_TILE_FIRST      # 256 x 128 for GEGLU's (h, g) column pairs
            split_k, in_kernel = 2, True
            if order == 0:
                p.order = 3
    if split_k is None:
        split_k = lib.aptp_conv_gemm_suggest_split_k(ctypes.byref(p))
    p.split_k = max(1, int(split_k))
    ws = cnt = None
    gn_fused = False
    if gn is not None and FUSE_GN_REDUCE and p.split_k > 1 and Hout * Wout <= GN_REDUCE_MAX_HW:
        gamma, beta, groups, eps_, silu_, Cn = gn
        cgn = Cn // groups
        gn_fused = Cn % 8 == 0 and Cn % groups == 0 and cgn % 4 == 0 and Hout * Wout * cgn <= 4 * 24 * 256 and Cn <= nout
    if gn_fused:
        in_kernel = False                  # the K-slices are combined by the reduce launch, which also normalises
        colstats = False
        p.gn_gamma, p.gn_beta, p.gn_groups, p.gn_C, p.gn_silu, p.gn_eps = gamma.data_ptr(), beta.data_ptr(), groups, Cn, int(silu_), eps_
    if p.split_k > 1:
        if SPLITK_FORCE_IN_KERNEL and not gn_fused:
            in_kernel = True               # (A/B switch: every split launch combines its slices itself, no reduce launches)
        if in_kernel is None:
            in_kernel = p.split_k <= 4 or explicit_split
        # only the workgroup that combines the slices can emit row / column statistics.  Column statistics are worth the
        # in-kernel form up to 2 slices (beyond that it loses more than the GroupNorm statistics pass it saves: measured
        # +8 us on the 4-slice level-32 convs against a 6.5 us pass + a kernel boundary)
        want_cols = colstats and COLSTATS and Hout * Wout >= COLSTATS_MIN_HW and act != ACT_GEGLU and not out_f32 \
            and (in_kernel or p.split_k <= COLS_SPLIT_MAX)
        if rowstats or want_cols:
            in_kernel = True
        sk_tile = p.tile >= SK_TILE_FIRST     # persistent stream-K tiles: partial tiles are always combined in-kernel
        if (SPLITK_IN_KERNEL or sk_tile) and (in_kernel or sk_tile) and lib.aptp_conv_gemm_tiles(ctypes.byref(p)) <= _N_COUNTERS:
            cnt = _tile_counters(x.device)
            if cnt is not None:
                p.tile_counters = cnt.data_ptr()
        if sk_tile and cnt is None:
            p.split_k = 1                     # no counters at hand (pool exhausted / first use under capture): whole tiles only
        else:
            ws = _workspace(lib.aptp_conv_gemm_workspace_bytes(ctypes.byref(p)), x.device)
            p.workspace = ws.data_ptr()
    stats = None
    if rowstats and (p.split_k == 1 or cnt is not None):
        slots = lib.aptp_conv_gemm_rowstat_slots(ctypes.byref(p))
        assert slots % 2 == 0
        stats = torch.empty(slots // 2, B * Hout * Wout, 4, dtype=torch.float32, device=x.device)
        p.rowstat_out, p.rowstat_slots = stats.data_ptr(), slots
    cstats = ustat = None
    if colstats and COLSTATS and Hout * Wout >= COLSTATS_MIN_HW and act != ACT_GEGLU and not out_f32 \
            and (p.split_k == 1 or cnt is not None) and _ld(out) % 8 == 0 and out.data_ptr() % 16 == 0 and nout % 8 == 0 \
            and (residual is None or (_ld(residual) % 8 == 0 and residual.data_ptr() % 16 == 0)) \
            and (depth_in is None or (_ld(depth_in) % 8 == 0 and depth_in.data_ptr() % 16 == 0)) and p.epilogue == 0:
        rpb = lib.aptp_conv_gemm_colstat_rows(ctypes.byref(p))
        if rpb > 0 and (Hout * Wout) % rpb == 0:
            M = B * Hout * Wout
            nblk = (M + rpb - 1) // rpb + 16          # (+ the padding blocks of the last, partial tile)
            cstats = torch.empty(nblk, nout, 2, dtype=torch.float32, device=x.device)
            p.colstat_out, p.colstat_ld = cstats.data_ptr(), nout
            ustat = _ustat_alloc(B, nout)
            if ustat is not None:
                p.ustat_out, p.ustat_unit, p.ustat_units, p.ustat_nrep = ustat[0].data_ptr(), ustat[1], ustat[2], ustat[3]
    if ln is not None:
        ln_stats, ln_eps = ln
        if pw.ln_colsum is None:
            raise ValueError("conv_gemm: ln= needs weights packed with ln_gamma / ln_beta")
        assert ln_stats.dtype == torch.float32 and ln_stats.is_contiguous() and ln_stats.shape[1] == B * Hout * Wout \
            and ln_stats.shape[2] == 4
        p.ln_stats, p.ln_slots, p.ln_colsum = ln_stats.data_ptr(), 2 * ln_stats.shape[0], pw.ln_colsum.data_ptr()
        p.ln_eps, p.ln_C = ln_eps, Cx
    elif pw.ln_colsum is not None:
        raise ValueError("conv_gemm: weights with a folded LayerNorm need ln=(stats, eps)")
    _lib.check(lib.aptp_conv_gemm(ctypes.byref(p), _stream()), "aptp_conv_gemm")
    if cstats is not None:
        _colstats_put(out, cstats, rpb, ustat)
    if LAUNCH_LOG is not None:
        LAUNCH_LOG.append({"params": p, "flops": 2.0 * B * Hout * Wout * pw.N * (pw.KH * pw.KW * pw.Cin + pw.Cin2),
                           "keep": (x, pw, out, rowbias, colgate, corr, residual, depth, depth_in, ws, stats, ln, cnt, cstats, x2, gn, ustat)})
    if f32 and gn_f32 is not None:
        gn = gn_f32
    if gn is not None and not gn_fused:
        gamma, beta, groups, eps_, silu_, Cn = gn
        return groupnorm(out, gamma, beta, groups, eps_, silu_, C=Cn)
    return (out, stats) if rowstats else out


def linear(x: torch.Tensor, pw: PackedWeight, **kw) -> torch.Tensor:
    """x [B, L, C] -> [B, L, N] through the 1x1 path (tokens = H, W = 1)."""
    B, L, C = x.shape
    out = kw.pop("out", None)
    for k in ("residual", "depth_in"):
        if kw.get(k) is not None:
            kw[k] = kw[k].unsqueeze(2)
    y = conv_gemm(x.unsqueeze(2), pw, pad=0, out=None if out is None else out.unsqueeze(2), **kw)
    if "rowstats" in kw:
        return (y[0].squeeze(2), y[1]) if kw["rowstats"] else (y.squeeze(2), None)
    return y.squeeze(2)


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool,
              C: Optional[int] = None, out: Optional[torch.Tensor] = None, keep_stats: bool = False, variant: int = 0):
    """GroupNorm(+SiLU) of a [B,H,W,Cp] tensor over its first C real channels (Cp = roundup8(C)).
    keep_stats: also return the fp32 [B, nchunk, groups, 2] statistics partials (needed by groupnorm_bwd; variant 4: one
    launch on the small maps, three elsewhere).  variant: see AptpGroupNormParams in include/aptp_hip.h."""
    lib = _lib.load()
    _check_act(x, "groupnorm x")
    B, H, W, Cp = x.shape
    C = Cp if C is None else C
    assert Cp == round_up(C, 8), f"groupnorm: tensor width {Cp} != roundup8({C})"
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C and beta.numel() == C
    if out is None:
        out = torch.empty(B, H, W, Cp, dtype=x.dtype, device=x.device)
    p = GroupNormParams()
    p.x, p.ldx, p.y, p.ldy = x.data_ptr(), _ld(x), out.data_ptr(), _ld(out)
    p.B, p.HW, p.C, p.groups = B, H * W, C, groups
    p.gamma, p.beta, p.eps, p.silu = gamma.data_ptr(), beta.data_ptr(), eps, int(silu)
    if x.dtype == torch.float32:          # fp32 parity path (csrc/parity_f32.hip): three plain launches, no producer statistics
        assert out.dtype == torch.float32 and not keep_stats
        p.io_f32, p.variant = 1, 1
        ws = _workspace(lib.aptp_groupnorm_workspace_bytes(ctypes.byref(p)), x.device)
        p.workspace = ws.data_ptr()
        _lib.check(lib.aptp_groupnorm(ctypes.byref(p), _stream()), "aptp_groupnorm(io_f32)")
        return out
    if keep_stats:
        # [B, nchunk + 1, G, 2]: the partials the backward needs, plus the finalised (mean, rstd) slot the kernels append
        nch = lib.aptp_groupnorm_nchunk(H * W)
        ws_full = torch.empty(B * (nch + 1) * groups * 2, dtype=torch.float32, device=x.device)
        ws = ws_full[:B * nch * groups * 2].view(B, nch, groups, 2)
        p.workspace = ws_full.data_ptr()
        p.variant = 4 if GN_STATS_ONE_LAUNCH else 1
        _lib.check(lib.aptp_groupnorm(ctypes.byref(p), _stream()), "aptp_groupnorm")
        return out, ws
    else:
        ws = _workspace(lib.aptp_groupnorm_workspace_bytes(ctypes.byref(p)), x.device)
    p.variant = variant if variant else GN_DEFAULT_VARIANT
    p.workspace = ws.data_ptr()
    segs = _colstats_get(x, C) if (COLSTATS and variant == 0 and H * W >= COLSTATS_MIN_HW) else None
    if segs is not None:
        for i, (st, rpb, cseg, us) in enumerate(segs):
            p.colstats[i].stats, p.colstats[i].ld, p.colstats[i].rows_per_block, p.colstats[i].C = st.data_ptr(), st.shape[1], rpb, cseg
            if us is not None and us[0].shape[0] == us[3] * B * us[2] * 2:
                p.colstats[i].ustats, p.colstats[i].unit, p.colstats[i].units, p.colstats[i].nrep = us[0].data_ptr(), us[1], us[2], us[3]
        p.variant = 1                       # the multi-launch skeleton (statistics pass replaced by the finalise)
    cnt = _tile_counters(x.device) if (GN_FUSED_FINALIZE and B <= _N_COUNTERS) else None
    if cnt is not None:
        p.counters = cnt.data_ptr()
    _lib.check(lib.aptp_groupnorm(ctypes.byref(p), _stream()), "aptp_groupnorm")
    if GN_LAUNCH_LOG is not None:
        GN_LAUNCH_LOG.append({"params": p, "bytes": 2.0 * B * H * W * (C + Cp), "keep": (x, out, gamma, beta, ws, segs, cnt)})
    return (out, ws) if keep_stats else out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm over the last dim of a bf16 [B, L, C] tensor."""
    lib = _lib.load()
    B, L, C = x.shape
    assert x.dtype in (torch.bfloat16, torch.float32) and x.stride(2) == 1 and x.is_cuda
    ld = x.stride(1) if L > 1 else max(x.stride(1), C)
    assert B == 1 or L == 1 or x.stride(0) == L * ld
    if out is None:
        out = torch.empty(B, L, C, dtype=x.dtype, device=x.device)
    p = LayerNormParams()
    p.io_f32 = int(x.dtype == torch.float32)      # fp32 parity path (csrc/parity_f32.hip)
    p.x, p.ldx, p.y, p.ldy = x.data_ptr(), ld, out.data_ptr(), out.stride(1) if L > 1 else C
    p.rows, p.C = B * L, C
    p.gamma, p.beta, p.eps = gamma.data_ptr(), beta.data_ptr(), eps
    _lib.check(lib.aptp_layernorm(ctypes.byref(p), _stream()), "aptp_layernorm")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None,
              out: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T * scale) v with head_dim 64.  q [B, Lq, >=heads*64], k/v [B, Lk, >=heads*64] are bf16 views
    (e.g. column slices of a fused QKV buffer); heads are laid out as consecutive 64-wide column blocks."""
    lib = _lib.load()
    B, Lq = q.shape[0], q.shape[1]
    Lk = k.shape[1]
    for t in (q, k, v):
        assert t.dtype == q.dtype and t.dtype in (torch.bfloat16, torch.float32) and t.stride(2) == 1 and t.shape[2] == heads * 64 and t.is_cuda
    if out is None:
        out = torch.empty(B, Lq, heads * 64, dtype=q.dtype, device=q.device)
    p = AttentionParams()
    p.io_f32 = int(q.dtype == torch.float32)      # fp32 parity path (csrc/parity_f32.hip)
    p.q, p.q_stride_b, p.q_stride_l = q.data_ptr(), q.stride(0), q.stride(1)
    p.k, p.k_stride_b, p.k_stride_l = k.data_ptr(), k.stride(0), k.stride(1)
    p.v, p.v_stride_b, p.v_stride_l = v.data_ptr(), v.stride(0), v.stride(1)
    p.o, p.o_stride_b, p.o_stride_l = out.data_ptr(), out.stride(0), out.stride(1)
    p.B, p.heads, p.Lq, p.Lk = B, heads, Lq, Lk
    p.scale = (1.0 / 8.0) if scale is None else scale
    p.variant = ATTN_VARIANT
    if lse is not None:
        assert lse.dtype == torch.float32 and lse.is_contiguous() and tuple(lse.shape) == (B, heads, Lq)
        p.lse = lse.data_ptr()
    if ATTN_LAUNCH_LOG is not None and not p.io_f32:
        ATTN_LAUNCH_LOG.append({"params": p, "flops": 4.0 * B * heads * Lq * Lk * 64, "keep": (q, k, v, out, lse), "Lk": Lk})
    _lib.check(lib.aptp_attention(ctypes.byref(p), _stream()), "aptp_attention")
    return out


# GroupNorm applied by the reduce launch of a split-K convolution (conv_gemm(gn=...)) on maps of at most this many pixels
# (SD-2.1 levels 16 and 8: norm2 + SiLU of the 12 resnets there).  Measured on MI355X: it TIES the separate launches
# (187.3-187.7 vs 187.6 steps/s; 327 instead of 339 kernels per step): only groups x B = 64 workgroups exist, and their
# 16 waves each stream the K-slices about as slowly as splitk_reduce_kernel on 256 CUs plus the single-launch GroupNorm take
# together (a 256-thread version lost 3.8 %).  Off by default; APTP_FUSE_GN_REDUCE=1 turns it on (A/B timing, tests).
FUSE_GN_REDUCE = os.environ.get("APTP_FUSE_GN_REDUCE", "0") != "0"
GN_REDUCE_MAX_HW = 256


# Fused transformer tail (aptp_ff_tail): LN3 -> GEGLU projection -> ff.net[2] + residual -> proj_out + residual as one kernel
# per 64-token tile.  Worth it only where a row tile per CU fills the chip and the activations, not the weights, are the
# bytes that matter: M >= FUSE_TAIL_MIN_ROWS (SD-2.1 level 64 at bs >= 4).  APTP_FUSE_TAIL=0 disables; APTP_FUSE_TAIL_MIN_ROWS
# moves the threshold (tests exercise the kernel on small maps).
# Round 4: OFF by default.  With the lean linear kernel (csrc/lin_gemm.hip) the three separate launches are faster than the
# fused tile (same-box A/B on the headline forward: 202.8 steps/s without, 201.6 with; round 2, against the general kernel, the
# fused tile won by 0.3-0.6 %): one workgroup per CU walks 40 dependent K-steps while the launches keep 3-5 workgroups per CU
# in different phases.  APTP_FUSE_TAIL=1 turns it back on (tests of the kernel, A/B timing); it still saves 0.8 GB of HBM
# traffic per step.
FUSE_TAIL = os.environ.get("APTP_FUSE_TAIL", "0") != "0"
FUSE_TAIL_MIN_ROWS = int(os.environ.get("APTP_FUSE_TAIL_MIN_ROWS", "16384"))


def ff_tail_supported(h: torch.Tensor, pw1: PackedWeight, pw2: PackedWeight, pw3: PackedWeight) -> bool:
    if not (FUSE_TAIL and h.is_cuda and h.dtype == torch.bfloat16 and pw1.geglu and pw1.ln_colsum is not None):
        return False
    B, L, C = h.shape
    if B * L < FUSE_TAIL_MIN_ROWS or pw2.N != C or pw3.N != C or pw2.bias is None or pw3.bias is None or pw1.bias is None:
        return False
    return bool(_lib.load().aptp_ff_tail_supported(B * L, C, pw1.N, pw1.cin_pad, pw2.cin_pad, pw3.cin_pad))


def ff_tail(h: torch.Tensor, x: torch.Tensor, pw1: PackedWeight, pw2: PackedWeight, pw3: PackedWeight, eps: float = 1e-5,
            out: Optional[torch.Tensor] = None, colstats: bool = False) -> torch.Tensor:
    """y = proj_out(h + ff2(GEGLU(LN(h) W1'))) + x on token tensors [B, L, C] (bf16, contiguous channels, uniform row stride);
    pw1 = GEGLU projection packed with geglu=True and the LayerNorm folded in, pw2 = ff.net[2], pw3 = proj_out."""
    lib = _lib.load()
    B, L, C = h.shape
    for t, nm in ((h, "h"), (x, "x")):
        assert t.dtype == torch.bfloat16 and t.is_cuda and tuple(t.shape) == (B, L, C) and t.stride(2) == 1, nm
    if out is None:
        out = torch.empty(B, L, C, dtype=torch.bfloat16, device=h.device)
    else:
        _colstats_drop(out.unsqueeze(2))
    assert tuple(out.shape) == (B, L, C) and out.dtype == torch.bfloat16 and out.stride(2) == 1
    p = FfTailParams()
    for name, t in (("h", h), ("x", x), ("y", out)):
        _, _, _, ld = _rows(t)
        assert B == 1 or t.stride(0) == L * ld, "rows must be uniformly strided"
        setattr(p, name, t.data_ptr())
        setattr(p, "ld" + name, ld)
    p.w1, p.b1, p.cs1, p.n1, p.ld1 = pw1.w.data_ptr(), pw1.bias.data_ptr(), pw1.ln_colsum.data_ptr(), pw1.N, pw1.cin_pad
    p.w2, p.b2, p.ld2 = pw2.w.data_ptr(), pw2.bias.data_ptr(), pw2.cin_pad
    p.w3, p.b3, p.ld3 = pw3.w.data_ptr(), pw3.bias.data_ptr(), pw3.cin_pad
    p.M, p.C, p.eps = B * L, C, eps
    cstats = None
    if colstats and COLSTATS and L >= COLSTATS_MIN_HW and L % 64 == 0:
        cstats = torch.empty(B * L // 64, C, 2, dtype=torch.float32, device=h.device)
        p.colstat, p.colstat_ld = cstats.data_ptr(), C
    _lib.check(lib.aptp_ff_tail(ctypes.byref(p), _stream()), "aptp_ff_tail")
    if cstats is not None:
        _colstats_put(out.unsqueeze(2), cstats, 64)
    if LAUNCH_LOG is not None:       # same contraction work as the three aptp_conv_gemm launches it replaces: part of the family
        LAUNCH_LOG.append({"params": p, "fn": "aptp_ff_tail", "flops": 2.0 * B * L * (C * pw1.N + pw2.Cin * C + C * C),
                           "keep": (h, x, out, pw1, pw2, pw3, cstats)})
    return out


def unet_prologue(sample: torch.Tensor, timesteps: torch.Tensor, freqs: torch.Tensor, cin_pad: int):
    """sample [B,C,H,W] (fp32 / bf16, NCHW contiguous) -> (x bf16 [B,H,W,cin_pad] with zero padding channels,
    t_emb bf16 [B, 2*len(freqs)] = [cos | sin] of timesteps * freqs) in one launch (include/aptp_hip.h)."""
    lib = _lib.load()
    assert sample.is_cuda and sample.dim() == 4 and sample.dtype in (torch.float32, torch.bfloat16)
    sample = sample.contiguous()
    B, C, H, W = sample.shape
    t = timesteps.to(device=sample.device, dtype=torch.float32).contiguous()
    assert t.numel() == B and freqs.dtype == torch.float32 and freqs.is_contiguous()
    x = torch.empty(B, H, W, cin_pad, dtype=torch.bfloat16, device=sample.device)
    temb = torch.empty(B, 2 * freqs.numel(), dtype=torch.bfloat16, device=sample.device)
    p = UnetPrologueParams()
    p.sample, p.sample_bf16, p.x = sample.data_ptr(), int(sample.dtype == torch.bfloat16), x.data_ptr()
    p.B, p.C, p.H, p.W, p.cin_pad = B, C, H, W, cin_pad
    p.timesteps, p.freqs, p.half, p.t_emb = t.data_ptr(), freqs.data_ptr(), freqs.numel(), temb.data_ptr()
    _lib.check(lib.aptp_unet_prologue(ctypes.byref(p), _stream()), "aptp_unet_prologue")
    return x, temb


def unet_epilogue(y: torch.Tensor, channels: int, out_dtype: torch.dtype) -> torch.Tensor:
    """conv_out's fp32 [B,H,W,ld] -> [B,channels,H,W] (fp32 or bf16) in one launch."""
    lib = _lib.load()
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and y.is_contiguous() and out_dtype in (torch.float32, torch.bfloat16)
    B, H, W, ld = y.shape
    out = torch.empty(B, channels, H, W, dtype=out_dtype, device=y.device)
    p = UnetEpilogueParams()
    p.y, p.ldy, p.out, p.out_bf16 = y.data_ptr(), ld, out.data_ptr(), int(out_dtype == torch.bfloat16)
    p.B, p.C, p.H, p.W = B, channels, H, W
    _lib.check(lib.aptp_unet_epilogue(ctypes.byref(p), _stream()), "aptp_unet_epilogue")
    return out


# ------------------------------------------------------------------------------------------------------------------
# backward-path wrappers
# ------------------------------------------------------------------------------------------------------------------
def pack_weight_dgrad(w: torch.Tensor, device=None) -> PackedWeight:
    """Packed weights of the data-gradient contraction: OIHW -> IOHW with the filter rotated by 180 degrees
    (a linear weight [out,in] -> its transpose)."""
    if w.dim() == 2:
        return pack_weight(w.detach().t().contiguous(), None, device=device)
    return pack_weight(w.detach().flip(2, 3).transpose(0, 1).contiguous(), None, device=device)


def _rows(t: torch.Tensor):
    """[B, H, W, C] or [B, L, C] -> (B, rows per sample, C, ld)"""
    if t.dim() == 4:
        return t.shape[0], t.shape[1] * t.shape[2], t.shape[3], _ld(t)
    B, L, C = t.shape
    return B, L, C, (t.stride(1) if L > 1 else max(t.stride(1), C))


def _mse_order(t: torch.Tensor):
    """dimension order that sorts t's strides descending (ties by position): the permutation under which a permuted view --
    the NCHW face of a channels-last activation -- becomes row-major again"""
    return tuple(sorted(range(t.dim()), key=lambda d: (-t.stride(d), d)))


def _mse_view(t: torch.Tensor, order=None):
    """(tensor to keep alive, rows, C, ld) of an operand of aptp_mse: any tensor whose elements, taken in dimension order
    `order`, form rows of C contiguous values at a constant row stride (contiguous tensors, NCHW faces of NHWC activations,
    channel slices of NHWC buffers); anything else is copied once."""
    if order is not None and order != tuple(range(t.dim())):
        t = t.permute(order)
    if t.is_contiguous():
        n = t.numel()
        C = t.shape[-1] if (t.dim() > 1 and t.shape[-1] % 8 == 0) else n
        return t, n // C, C, C
    if t.dim() >= 2 and t.stride(-1) == 1 and t.shape[-1] % 8 == 0:
        ld = t.stride(-2)
        ok = ld % 8 == 0 and ld >= t.shape[-1]
        for d in range(t.dim() - 2, 0, -1):               # outer dimensions must collapse onto the row stride
            ok = ok and t.stride(d - 1) == t.stride(d) * t.shape[d]
        if ok:
            return t, t.numel() // t.shape[-1], t.shape[-1], ld
    return _mse_view(t.contiguous())


def _mse_operands(a: torch.Tensor, b: torch.Tensor):
    order = _mse_order(a)
    a2, rows, C, lda = _mse_view(a, order)
    b2, rows_b, C_b, ldb = _mse_view(b, order)
    if (rows_b, C_b) != (rows, C):                         # different row shapes for the same elements: flatten both
        a2, rows, C, lda = _mse_view(a2.contiguous().reshape(-1))
        b2, _, _, ldb = _mse_view(b2.contiguous().reshape(-1))
    assert C % 8 == 0, "aptp_mse needs a multiple of 8 elements per row"
    return order, a2, b2, rows, C, lda, ldb


def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """mean((a - b)^2) as an fp32 device scalar: two launches, no copies, fixed summation order (csrc/loss_ops.hip)"""
    lib = _lib.load()
    assert a.shape == b.shape and a.dtype == b.dtype and a.dtype in (torch.bfloat16, torch.float32) and a.is_cuda
    _, a, b, rows, C, lda, ldb = _mse_operands(a, b)
    nblk = lib.aptp_mse_nblocks(rows, C)
    partial = torch.empty(nblk, dtype=torch.float32, device=a.device)
    out = torch.empty((), dtype=torch.float32, device=a.device)
    p = MseParams()
    p.a, p.lda, p.b, p.ldb, p.rows, p.C, p.f32 = a.data_ptr(), lda, b.data_ptr(), ldb, rows, C, int(a.dtype == torch.float32)
    p.partial, p.out, p.backward = partial.data_ptr(), out.data_ptr(), 0
    _lib.check(lib.aptp_mse(ctypes.byref(p), _stream()), "aptp_mse")
    return out


def mse_bwd(a: torch.Tensor, b: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """d mean((a - b)^2) / da * g = (a - b) * 2 g / n, in a's dtype, shape AND memory order (the gradient of the NCHW face of
    a channels-last activation is channels-last again, so the next backward kernel takes it without a copy)"""
    lib = _lib.load()
    order, a2, b2, rows, C, lda, ldb = _mse_operands(a, b)
    g = g.detach().to(torch.float32).reshape(1)
    pshape = [a.shape[d] for d in order]
    da = torch.empty(pshape, dtype=a.dtype, device=a.device)
    p = MseParams()
    p.a, p.lda, p.b, p.ldb, p.rows, p.C, p.f32 = a2.data_ptr(), lda, b2.data_ptr(), ldb, rows, C, int(a.dtype == torch.float32)
    p.g, p.da, p.ldda, p.backward = g.data_ptr(), da.data_ptr(), C, 1
    _lib.check(lib.aptp_mse(ctypes.byref(p), _stream()), "aptp_mse(bwd)")
    inv = [0] * len(order)
    for i, d in enumerate(order):
        inv[d] = i
    return da.permute(inv)


def fold_rows(partials: torch.Tensor, n_rows: int, C: int, out: Optional[torch.Tensor] = None,
              tail_out: Optional[torch.Tensor] = None, deferrable: bool = False, pair_split: bool = False,
              tail_n: int = 0, pair_out=None) -> torch.Tensor:
    """out[row, c] = sum_r partials[r, row, c] (fp32, fixed order); partials is [R, n_rows (+ tail rows), C] contiguous; `out` may
    be a [n_rows, ld >= C] buffer whose extra columns are left alone (the padded packed layout of a weight gradient).
    tail_out (fp32, contiguous, a multiple of C elements): the rows after the first n_rows are summed into it instead -- the
    bias-gradient slabs that ride behind a weight gradient's slabs, folded by the same launch."""
    lib = _lib.load()
    if pair_split and pair_out is not None:
        # (a, b) pairs straight into two existing gradient tensors (direct-gradient mode): a -> pair_out[0], b -> pair_out[1]
        a_t, b_t = pair_out
        assert n_rows == 1 and C % 2 == 0 and a_t.dtype == b_t.dtype == torch.float32 and a_t.is_contiguous() and b_t.is_contiguous() \
            and a_t.numel() >= C // 2 and b_t.numel() >= C // 2
        assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.numel() % C == 0
        p = FoldRowsParams()
        p.partials, p.out, p.R, p.n_rows, p.C, p.ld_out = partials.data_ptr(), a_t.data_ptr(), partials.numel() // C, 1, C, C
        p.tail_out, p.tail_rows, p.pair_split = b_t.data_ptr(), 0, 1
        if FOLD_DEFER is not None and deferrable:
            FOLD_DEFER.append({"params": p, "keep": (partials, a_t, b_t)})
        else:
            _lib.check(lib.aptp_fold_rows(ctypes.byref(p), _stream()), "aptp_fold_rows")
        return None
    if tail_n:
        # the tail destination holds tail_n values (a bias gradient written in place); the slabs carry whole rows of C
        tail_rows = (tail_n + C - 1) // C
        assert tail_out is not None and tail_out.dtype == torch.float32 and tail_out.is_contiguous() and tail_out.numel() >= tail_n
    else:
        tail_rows = 0 if tail_out is None else tail_out.numel() // C
    total_rows = n_rows + tail_rows
    assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.numel() % (total_rows * C) == 0
    R = partials.numel() // (total_rows * C)
    if pair_split:
        # one row of (a, b) pairs -> out [2, half]: a in row 0, b in row 1 (half = C/2 rounded up to 4: both rows 16-byte aligned)
        assert n_rows == 1 and tail_out is None and out is None and C % 2 == 0
        half = round_up(C // 2, 4)
        out = torch.empty(2, half, dtype=torch.float32, device=partials.device)
        o2 = out
    else:
        if out is None:
            out = torch.empty(n_rows, C, dtype=torch.float32, device=partials.device)
        o2 = out.reshape(n_rows, -1)
        assert o2.dtype == torch.float32 and o2.stride(1) == 1 and o2.shape[1] >= C and (n_rows == 1 or o2.stride(0) == o2.shape[1])
    p = FoldRowsParams()
    p.partials, p.out, p.R, p.n_rows, p.C, p.ld_out = partials.data_ptr(), o2.data_ptr(), R, total_rows, C, o2.shape[1]
    if tail_out is not None:
        assert tail_out.dtype == torch.float32 and tail_out.is_contiguous() and (tail_n or tail_out.numel() == tail_rows * C)
        p.tail_out, p.tail_rows, p.tail_n = tail_out.data_ptr(), tail_rows, int(tail_n)
    p.pair_split = int(pair_split)          # (n_rows == 1: out = [C/2 first-of-pair sums | C/2 second-of-pair sums])
    if FOLD_DEFER is not None and deferrable:
        # deferred: the caller of the backward folds everything at once (FoldBatch); `out` is returned UNWRITTEN and must not be
        # read before that (weight / bias gradients: nothing reads them before the optimizer)
        FOLD_DEFER.append({"params": p, "keep": (partials, out, tail_out)})
        return out
    _lib.check(lib.aptp_fold_rows(ctypes.byref(p), _stream()), "aptp_fold_rows")
    return out


# Deferred slab folds.  The weight-gradient kernel splits the pixel range over workgroups and leaves fp32 slabs; summing them is
# one fold launch per weight -- 357 launches of 7.8 us in the expert fine-tune step, 2.8 ms of 39.  Nothing reads a weight (or
# bias) gradient before the optimizer, so with FOLD_DEFER set to a list the folds of a backward pass are only RECORDED (slabs and
# destinations stay referenced, so a capturing allocator cannot hand their memory to a later kernel), and FoldBatch runs them as
# ONE launch over a descriptor table in device memory -- after the replayed graph, next to the one-launch AdamW
# (train_step.GraphedFineTunerStep).  Bitwise the same sums as the single folds (same kernel body, same order).
FOLD_DEFER = None

# Direct parameter gradients (train_step.GraphedFineTunerStep sets it around its warm-up and capture).  The packed trainable
# tensors own PERSISTENT, zero-initialised .grad buffers, and the weight-gradient kernel, the slab folds and the norm-affine
# folds write their results straight into them; the autograd Functions return None for those inputs.  Autograd's
# AccumulateGrad is out of the loop: it clones any incoming gradient another object still references (a recorded, not yet
# executed fold) -- i.e. it would read the buffer BEFORE the fold wrote it -- and each fresh gradient tensor with pad
# columns cost a zero fill per step (80 launches).  Every parameter is used once per step, so "write" equals "accumulate".
GRAD_DIRECT = False


class FoldBatch:
    def __init__(self, records):
        lib = _lib.load()
        records = list(records)
        assert records
        dev = records[0]["keep"][0].device
        items = (FoldRowsParams * len(records))()
        starts = [0]
        for i, rec in enumerate(records):
            ctypes.memmove(ctypes.byref(items[i]), ctypes.byref(rec["params"]), ctypes.sizeof(FoldRowsParams))
            nb = lib.aptp_fold_rows_blocks(ctypes.byref(items[i]))
            assert nb > 0
            starts.append(starts[-1] + nb)
        assert starts[-1] < 2 ** 31
        self.items = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        self.starts = torch.tensor(starts, dtype=torch.int32).to(dev)
        self.n, self.total = len(records), starts[-1]
        self.keep = [r["keep"] for r in records]

    def run(self):
        lib = _lib.load()
        _lib.check(lib.aptp_fold_rows_many(self.items.data_ptr(), self.starts.data_ptr(), self.n, self.total, _stream()),
                   "aptp_fold_rows_many")


# Batched weight gradients (round 3).  Issued layer by layer, each aptp_conv_wgrad launch has to fill the chip on its own: a
# 176 x 176 projection at 16,384 pixels becomes 9 tiles x 57 pixel slices and 57 fp32 slabs to fold.  With WGRAD_DEFER set to a list
# (the graphed fine-tune step, direct-gradient mode) the stride-1 launches of a backward are only recorded -- operands are kept
# alive by the record -- and WgradBatch runs them as one launch per filter size afterwards, slices of ~WGRAD_BATCH_SLICE pixels:
# most weights need no slabs at all and their gradients go straight into the optimizer's buffer.  Run it BEFORE the FoldBatch of
# the same backward (the remaining slabs are folded there).
WGRAD_DEFER = None
WGRAD_SPLIT_RULE = "launch"        # "batch": per-layer launches use the batch's pixel split too (bitwise comparisons with a batched step)
WGRAD_BATCH_SLICE = int(os.environ.get("APTP_WGRAD_BATCH_SLICE", "2048"))


class WgradBatch:
    def __init__(self, records):
        lib = _lib.load()
        records = list(records)
        assert records
        dev = records[0]["keep"][0].device
        nbytes = int(lib.aptp_conv_wgrad_many_item_bytes())
        self.groups = []
        self.keep = [r["keep"] for r in records]
        for taps in (1, 9):
            recs = [r for r in records if r["taps"] == taps]
            if not recs:
                continue
            # long slices first: the short ones (small maps) fill the tail of the grid
            recs.sort(key=lambda r: -((r["params"].B * r["params"].H * r["params"].W) // r["params"].split_m))
            table = (ctypes.c_uint8 * (nbytes * len(recs)))()
            block_item, first = [], 0
            for i, r in enumerate(recs):
                nb = lib.aptp_conv_wgrad_many_blocks(ctypes.byref(r["params"]))
                assert nb > 0
                _lib.check(lib.aptp_conv_wgrad_many_fill(ctypes.byref(r["params"]), ctypes.addressof(table) + i * nbytes, first),
                           "aptp_conv_wgrad_many_fill")
                block_item.append(torch.full((nb,), i, dtype=torch.int32))
                first += nb
            assert first < 2 ** 31
            items = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
            assert items.data_ptr() % 16 == 0
            self.groups.append((taps, items, torch.cat(block_item).to(dev), len(recs), first))

    def run(self):
        lib = _lib.load()
        for taps, items, block_item, n, total in self.groups:
            _lib.check(lib.aptp_conv_wgrad_many(items.data_ptr(), block_item.data_ptr(), n, total, taps, _stream()),
                       "aptp_conv_wgrad_many")

    def launches(self):
        return len(self.groups)


def pack_dgrad_from_packed(pw: PackedWeight, pwb: PackedWeight) -> PackedWeight:
    """refresh the data-gradient operand `pwb` (ops.pack_weight_dgrad layout) from the forward operand `pw` of the same
    weights, bf16 -> bf16, one launch"""
    lib = _lib.load()
    taps = pw.KH * pw.KW
    p = PackDgradParams()
    p.src, p.dst, p.N, p.C, p.taps = pw.w.data_ptr(), pwb.w.data_ptr(), pw.N, pw.Cin, taps
    p.src_ld, p.dst_ld, p.dst_rows = pw.cin_pad, pwb.cin_pad, pwb.N
    assert pwb.KH == pw.KH and pwb.N >= pw.Cin and pwb.cin_pad >= pw.N and pw.Cin2 == 0
    _lib.check(lib.aptp_pack_dgrad(ctypes.byref(p), _stream()), "aptp_pack_dgrad")
    return pwb


class PackDgradBatch:
    """pack_dgrad_from_packed for a fixed list of (forward operand, data-gradient operand) pairs as ONE launch: the
    descriptors live in device memory (built once; the operands' storage must not move afterwards)."""

    def __init__(self, pairs):
        lib = _lib.load()
        pairs = list(pairs)
        assert pairs
        dev = pairs[0][0].w.device
        items = (PackDgradParams * len(pairs))()
        starts = [0]
        for i, (pw, pwb) in enumerate(pairs):
            assert pwb.KH == pw.KH and pwb.N >= pw.Cin and pwb.cin_pad >= pw.N and pw.Cin2 == 0
            assert pw.cin_pad % 8 == 0 and pwb.cin_pad % 8 == 0 and pw.w.data_ptr() % 16 == 0 and pwb.w.data_ptr() % 16 == 0
            p = items[i]
            p.src, p.dst, p.N, p.C, p.taps = pw.w.data_ptr(), pwb.w.data_ptr(), pw.N, pw.Cin, pw.KH * pw.KW
            p.src_ld, p.dst_ld, p.dst_rows = pw.cin_pad, pwb.cin_pad, pwb.N
            starts.append(starts[-1] + lib.aptp_pack_dgrad_blocks(ctypes.byref(p)))
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        self.items = raw.to(dev)
        self.starts = torch.tensor(starts, dtype=torch.int32).to(dev)
        self.n, self.total = len(pairs), starts[-1]
        self.keep = pairs
        self.key = tuple((pw.w.data_ptr(), pwb.w.data_ptr()) for pw, pwb in pairs)

    def run(self):
        lib = _lib.load()
        _lib.check(lib.aptp_pack_dgrad_many(self.items.data_ptr(), self.starts.data_ptr(), self.n, self.total, _stream()),
                   "aptp_pack_dgrad_many")


def gate_bwd(dy: torch.Tensor, y0: torch.Tensor, gate: torch.Tensor, want_dgate: bool = True):
    """dx = dy * gate (expanded over channel groups, batch tiled), dgate [Bg, G] = sum dy*y0 (fp32).
    want_dgate=False: the forward gate multiply (dy := y0) -- the partials are not folded, dgate is None."""
    lib = _lib.load()
    B, HW, C, lddy = _rows(dy)
    _, _, _, ldy0 = _rows(y0)
    assert dy.dtype == torch.bfloat16 and y0.dtype == torch.bfloat16 and dy.shape == y0.shape
    gate = gate.detach().to(dtype=torch.float32).contiguous()
    Bg, G = gate.shape
    dx = torch.empty_like(dy, memory_format=torch.contiguous_format)
    nchunk = lib.aptp_groupnorm_nchunk(HW)
    part = torch.empty(B, nchunk, G, dtype=torch.float32, device=dy.device)
    dgate = torch.empty(Bg, G, dtype=torch.float32, device=dy.device) if want_dgate else None
    p = GateBwdParams()
    p.dy, p.lddy, p.y0, p.ldy0, p.dx, p.lddx = dy.data_ptr(), lddy, y0.data_ptr(), ldy0, dx.data_ptr(), C
    p.B, p.HW, p.C, p.groups = B, HW, C, G
    p.gate, p.gate_B, p.dgate_partial, p.dgate = gate.data_ptr(), Bg, part.data_ptr(), (dgate.data_ptr() if want_dgate else None)
    _lib.check(lib.aptp_gate_bwd(ctypes.byref(p), _stream()), "aptp_gate_bwd")   # (the partials are folded by the same call)
    return dx, dgate


def geglu_fwd(hg: torch.Tensor, gate: Optional[torch.Tensor]) -> torch.Tensor:
    lib = _lib.load()
    B, HW, C2, ld = _rows(hg)
    C = C2 // 2
    out = torch.empty(*hg.shape[:-1], C, dtype=torch.bfloat16, device=hg.device)
    p = GegluParams()
    p.hg, p.ldhg, p.out, p.ldout = hg.data_ptr(), ld, out.data_ptr(), C
    p.B, p.HW, p.C = B, HW, C
    if gate is not None:
        gate = gate.detach().to(dtype=torch.float32).contiguous()
        p.gate, p.gate_B, p.groups = gate.data_ptr(), gate.shape[0], gate.shape[1]
    else:
        p.groups = 32 if C % 32 == 0 else 1
    p.backward = 0
    _lib.check(lib.aptp_geglu(ctypes.byref(p), _stream()), "aptp_geglu")
    return out


def geglu_bwd(hg: torch.Tensor, dout: torch.Tensor, gate: Optional[torch.Tensor]):
    lib = _lib.load()
    B, HW, C2, ld = _rows(hg)
    C = C2 // 2
    _, _, _, lddout = _rows(dout)
    dhg = torch.empty(*hg.shape, dtype=torch.bfloat16, device=hg.device)
    p = GegluParams()
    p.hg, p.ldhg, p.dout, p.lddout, p.dhg, p.lddhg = hg.data_ptr(), ld, dout.data_ptr(), lddout, dhg.data_ptr(), C2
    p.B, p.HW, p.C = B, HW, C
    Bg, G = (gate.shape if gate is not None else (1, 32 if C % 32 == 0 else 1))
    if gate is not None:
        gate = gate.detach().to(dtype=torch.float32).contiguous()
        p.gate, p.gate_B = gate.data_ptr(), Bg
    p.groups = G
    part = torch.empty(B, lib.aptp_groupnorm_nchunk(HW), G, dtype=torch.float32, device=hg.device)
    p.dgate_partial = part.data_ptr()
    dgate = None
    if gate is not None:
        dgate = torch.empty(Bg, G, dtype=torch.float32, device=hg.device)
        p.dgate = dgate.data_ptr()
    p.backward = 1
    _lib.check(lib.aptp_geglu(ctypes.byref(p), _stream()), "aptp_geglu(bwd)")
    return dhg, dgate


def depth_lerp(x_in: torch.Tensor, x_out: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """DepthGate.forward in one launch: (1 - d[b % dB]) * x_in + d[b % dB] * x_out; x_in may be a channel slice of a wider
    buffer (the un-sliced skip-concat of an up-block resnet, blocks.py:485-495)."""
    lib = _lib.load()
    B, HW, C, ldin = _rows(x_in)
    _, _, _, ldout = _rows(x_out)
    assert x_in.dtype == torch.bfloat16 and x_out.dtype == torch.bfloat16 and x_in.shape == x_out.shape and x_in.is_cuda
    d = d.detach().to(dtype=torch.float32).reshape(-1).contiguous()
    y = torch.empty(x_out.shape, dtype=torch.bfloat16, device=x_out.device)
    p = DepthLerpParams()
    p.x_in, p.ld_in, p.x_out, p.ld_out, p.y, p.ld_y = x_in.data_ptr(), ldin, x_out.data_ptr(), ldout, y.data_ptr(), C
    p.B, p.HW, p.C, p.d, p.d_B, p.backward = B, HW, C, d.data_ptr(), d.numel(), 0
    _lib.check(lib.aptp_depth_lerp(ctypes.byref(p), _stream()), "aptp_depth_lerp")
    return y


def depth_lerp_bwd(dy: torch.Tensor, x_in: torch.Tensor, x_out: torch.Tensor, d: torch.Tensor):
    """(d_in, d_out, dd [dB] fp32) of depth_lerp"""
    lib = _lib.load()
    B, HW, C, ldin = _rows(x_in)
    _, _, _, ldout = _rows(x_out)
    _, _, _, lddy = _rows(dy)
    d = d.detach().to(dtype=torch.float32).reshape(-1).contiguous()
    d_in = torch.empty(x_out.shape, dtype=torch.bfloat16, device=dy.device)
    d_out = torch.empty(x_out.shape, dtype=torch.bfloat16, device=dy.device)
    part = torch.empty(B, lib.aptp_groupnorm_nchunk(HW), dtype=torch.float32, device=dy.device)
    dd = torch.empty(d.numel(), dtype=torch.float32, device=dy.device)
    p = DepthLerpParams()
    p.x_in, p.ld_in, p.x_out, p.ld_out = x_in.data_ptr(), ldin, x_out.data_ptr(), ldout
    p.dy, p.ld_dy, p.d_in, p.ld_d_in, p.d_out, p.ld_d_out = dy.data_ptr(), lddy, d_in.data_ptr(), C, d_out.data_ptr(), C
    p.B, p.HW, p.C, p.d, p.d_B = B, HW, C, d.data_ptr(), d.numel()
    p.dd_partial, p.dd, p.backward = part.data_ptr(), dd.data_ptr(), 1
    _lib.check(lib.aptp_depth_lerp(ctypes.byref(p), _stream()), "aptp_depth_lerp(bwd)")
    return d_in, d_out, dd


def groupnorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                  silu: bool, stats: torch.Tensor, C: Optional[int] = None, want_pgrad: bool = False, pgrad_out=None,
                  add: Optional[torch.Tensor] = None):
    """dx (and, with want_pgrad, (dgamma, dbeta) fp32 [C]) of GroupNorm(+SiLU); C = real channel count when the
    tensor is padded to a multiple of 8.  add (bf16, x's shape): the gradient arriving over the residual path that forked
    at x -- summed into dx in fp32 inside the kernel instead of by a separate elementwise launch."""
    lib = _lib.load()
    B, HW, Cp, ldx = _rows(x)
    C = Cp if C is None else C
    _, _, _, lddy = _rows(dy)
    dx = torch.zeros(*x.shape, dtype=torch.bfloat16, device=x.device) if C != Cp else \
        torch.empty(*x.shape, dtype=torch.bfloat16, device=x.device)
    p = GroupNormBwdParams()
    p.x, p.ldx, p.dy, p.lddy, p.dx, p.lddx = x.data_ptr(), ldx, dy.data_ptr(), lddy, dx.data_ptr(), Cp
    p.B, p.HW, p.C, p.groups = B, HW, C, groups
    p.gamma, p.beta, p.eps, p.silu = gamma.data_ptr(), beta.data_ptr(), eps, int(silu)
    p.fwd_stats = stats.data_ptr()
    ws = torch.empty_like(stats)
    p.workspace = ws.data_ptr()
    if add is not None:
        assert add.dtype == torch.bfloat16 and add.shape == x.shape
        p.add, p.ldadd = add.data_ptr(), _rows(add)[3]
    part = None
    if want_pgrad:
        part = torch.empty(B, stats.shape[1], C, 2, dtype=torch.float32, device=x.device)
        p.pgrad_partial = part.data_ptr()
    _lib.check(lib.aptp_groupnorm_bwd(ctypes.byref(p), _stream()), "aptp_groupnorm_bwd")
    if want_pgrad:
        # (dbeta, dgamma) pairs -> two contiguous [C] gradients straight from the fold (was: one [C, 2] tensor + two strided copies)
        if pgrad_out is not None:          # (dgamma, dbeta) destinations: the parameters' own gradient buffers
            fold_rows(part.view(-1, 1, C * 2), 1, C * 2, deferrable=True, pair_split=True, pair_out=(pgrad_out[1], pgrad_out[0]))
            return dx, None, None
        pg = fold_rows(part.view(-1, 1, C * 2), 1, C * 2, deferrable=True, pair_split=True)
        return dx, pg[1, :C], pg[0, :C]
    return dx


def layernorm_pgrad(x: torch.Tensor, dy: torch.Tensor, eps: float = 1e-5, pgrad_out=None):
    """(dgamma, dbeta) fp32 [C] of LayerNorm over the last dim."""
    lib = _lib.load()
    B, L, C, ldx = _rows(x)
    _, _, _, lddy = _rows(dy)
    nchunk = lib.aptp_rows_nchunk(B * L)
    part = torch.empty(nchunk, C, 2, dtype=torch.float32, device=x.device)
    p = LayerNormPgradParams()
    p.x, p.ldx, p.dy, p.lddy, p.rows, p.C, p.eps, p.partial = x.data_ptr(), ldx, dy.data_ptr(), lddy, B * L, C, eps, part.data_ptr()
    _lib.check(lib.aptp_layernorm_pgrad(ctypes.byref(p), _stream()), "aptp_layernorm_pgrad")
    if pgrad_out is not None:
        fold_rows(part.view(-1, 1, C * 2), 1, C * 2, deferrable=True, pair_split=True, pair_out=(pgrad_out[1], pgrad_out[0]))
        return None, None
    pg = fold_rows(part.view(-1, 1, C * 2), 1, C * 2, deferrable=True, pair_split=True)
    return pg[1, :C], pg[0, :C]


def colsum(x: torch.Tensor, per_sample: bool = False) -> torch.Tensor:
    """fp32 column sums of a bf16 [..., C] tensor with contiguous last dim and uniform row stride (bias gradients): [C];
    per_sample: one sum per leading index, [B, C] (the gradient of a per-sample output bias), still two launches."""
    lib = _lib.load()
    C = x.shape[-1]
    B = x.shape[0] if per_sample else 1
    x2 = x.reshape(-1, C)
    rows = x2.shape[0] // B
    nchunk = lib.aptp_groupnorm_nchunk(rows) if B > 1 else lib.aptp_rows_nchunk(rows)
    part = torch.empty(nchunk, B, C, dtype=torch.float32, device=x.device)
    p = ColsumParams()
    p.x, p.ldx, p.rows, p.C, p.partial, p.batch = x2.data_ptr(), (x2.stride(0) if x2.shape[0] > 1 else C), rows, C, part.data_ptr(), B
    _lib.check(lib.aptp_colsum(ctypes.byref(p), _stream()), "aptp_colsum")
    out = fold_rows(part, B, C)
    return out if per_sample else out.view(C)


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, eps: float = 1e-5,
                  add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """add: as in groupnorm_bwd (the residual-path gradient of the fork at x, summed inside the kernel)"""
    lib = _lib.load()
    B, L, C, ldx = _rows(x)
    _, _, _, lddy = _rows(dy)
    dx = torch.empty(*x.shape, dtype=torch.bfloat16, device=x.device)
    p = LayerNormBwdParams()
    p.x, p.ldx, p.dy, p.lddy, p.dx, p.lddx = x.data_ptr(), ldx, dy.data_ptr(), lddy, dx.data_ptr(), C
    p.rows, p.C, p.gamma, p.eps = B * L, C, gamma.data_ptr(), eps
    if add is not None:
        assert add.dtype == torch.bfloat16 and add.shape == x.shape
        p.add, p.ldadd = add.data_ptr(), _rows(add)[3]
    _lib.check(lib.aptp_layernorm_bwd(ctypes.byref(p), _stream()), "aptp_layernorm_bwd")
    return dx


ATTN_BWD_Q_SPLIT = True      # False: never split the dK/dV kernel's query range (A/B timing, tests of both forms)


def attention_bwd(q, k, v, o, dout, lse, heads: int, dq, dk, dv, scale: Optional[float] = None, q_split: Optional[int] = None):
    """dq/dk/dv are preallocated bf16 views (e.g. column slices of one fused gradient buffer), written in place.
    q_split: slices of the query range in the dK/dV kernel (None: the library's rule -- only where few keys leave the chip idle)"""
    lib = _lib.load()
    B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
    delta = torch.empty(B, heads, Lq, dtype=torch.float32, device=q.device)
    p = AttentionBwdParams()
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o), ("dout", dout), ("dq", dq), ("dk", dk), ("dv", dv)):
        assert t.dtype == torch.bfloat16 and t.stride(2) == 1 and t.shape[2] == heads * 64, name
        setattr(p, name, t.data_ptr())
        setattr(p, name + "_stride_b", t.stride(0))
        setattr(p, name + "_stride_l", t.stride(1))
    p.lse, p.delta = lse.data_ptr(), delta.data_ptr()
    p.B, p.heads, p.Lq, p.Lk = B, heads, Lq, Lk
    p.scale = (1.0 / 8.0) if scale is None else scale
    if q_split is None:
        q_split = lib.aptp_attention_bwd_q_split(ctypes.byref(p)) if ATTN_BWD_Q_SPLIT else 1
    ws = None
    if q_split > 1:
        ws = torch.empty(lib.aptp_attention_bwd_workspace_bytes(ctypes.byref(p), q_split), dtype=torch.uint8, device=q.device)
        p.q_split, p.workspace = q_split, ws.data_ptr()
    _lib.check(lib.aptp_attention_bwd(ctypes.byref(p), _stream()), "aptp_attention_bwd")
    return dq, dk, dv


# ------------------------------------------------------------------------------------------------------------------
# weight gradients (expert fine-tuning): dW[n, tap, c] = sum_m dy[m, n] * x[pix(m, tap), c]
# Stride-1 3x3 / 1x1 / linear layers (all but the six down/up-sampler convolutions of SD-2.1) run aptp_conv_wgrad: the
# operands stay in their forward layout and the kernel reads its LDS tiles K-major (transposed LDS reads), one input halo
# per 32-pixel step serving all nine taps.  The remaining geometries fall back to the implicit-GEMM kernel on transposed
# copies ("activations" = dy^T, "weights" = im2col(x)^T; the copies are torch strided copies).
# WGRAD_KERNEL=False forces the fallback everywhere (A/B timing, tests of both forms).
# ------------------------------------------------------------------------------------------------------------------
WGRAD_KERNEL = os.environ.get("APTP_WGRAD_KERNEL", "1") != "0"
WGRAD_SPLIT_TARGET = int(os.environ.get("APTP_WGRAD_SPLIT_TARGET", "0"))
WGRAD_PARITY = True       # _wgrad_parity where its rule selects it (False: copies + GEMM for every resampling convolution)
WGRAD_RESAMPLE = os.environ.get("APTP_WGRAD_RESAMPLE", "1") != "0"    # stride-2 / nearest-x2 3x3 layers through aptp_conv_wgrad itself


def _wgrad_direct(x: torch.Tensor, dy: torch.Tensor, KH: int, KW: int, split_m: Optional[int] = None,
                  out: Optional[torch.Tensor] = None, want_db: bool = False, stride: int = 1, ups: int = 0,
                  db_out: Optional[torch.Tensor] = None):
    """x [B,H,W,C], dy [B,H,W,N] (bf16, channels contiguous, uniform pixel stride) -> fp32 [N, KH*KW, C] or None when
    the geometry is not handled by aptp_conv_wgrad.  out: an fp32 [N, KH*KW, ld >= C] buffer (a packed-layout gradient) that
    receives the result in its first C columns (the rest is left alone); returned instead of a fresh tensor."""
    lib = _lib.load()
    B, C = x.shape[0], x.shape[3]
    H, W, N = dy.shape[1], dy.shape[2], dy.shape[3]           # (the OUTPUT map: the pixels the contraction runs over)
    p = WgradParams()
    p.x, p.ldx, p.dy, p.lddy = x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy)
    p.B, p.H, p.W, p.C, p.N, p.KH, p.KW = B, H, W, C, N, KH, KW
    p.stride, p.ups = stride, ups
    if x.data_ptr() % 16 or dy.data_ptr() % 16 or not lib.aptp_conv_wgrad_supported(ctypes.byref(p)):
        return None
    _check_act(x, "conv_wgrad x")
    _check_act(dy, "conv_wgrad dy")
    # batched mode (WGRAD_DEFER): the launch is only RECORDED -- every stride-1 weight gradient of the backward runs as one launch
    # per filter size when the backward is complete (WgradBatch), so a pixel slice only has to be worth a workgroup
    defer = WGRAD_DEFER is not None and out is not None and stride == 1 and (not want_db or db_out is not None)
    p.split_m = int(split_m) if split_m else lib.aptp_conv_wgrad_suggest_split(ctypes.byref(p))
    if not split_m and stride == 1 and (defer or WGRAD_SPLIT_RULE == "batch"):
        p.split_m = max(1, min(64, (B * H * W + WGRAD_BATCH_SLICE // 2) // WGRAD_BATCH_SLICE))

    def launch():
        if defer:
            WGRAD_DEFER.append({"params": p, "taps": KH * KW, "keep": (x, dy, out, db_out)})
        else:
            _lib.check(lib.aptp_conv_wgrad(ctypes.byref(p), _stream()), "aptp_conv_wgrad")
    if not split_m and WGRAD_SPLIT_TARGET and not defer and WGRAD_SPLIT_RULE != "batch":
        # (A/B of the pixel-range split: workgroups wanted per launch; the library's own rule is 512)
        tiles = ((N + 63) // 64) * ((C + 63) // 64)
        nsteps = (B * H * W + 31) // 32
        p.split_m = max(1, min(-(-WGRAD_SPLIT_TARGET // tiles), max(1, nsteps // 4), 64))
    if out is not None:
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape[:2]) == (N, KH * KW) and out.shape[2] >= C
    if db_out is not None:
        assert want_db and db_out.dtype == torch.float32 and db_out.is_contiguous() and db_out.numel() >= N
    if p.split_m == 1:
        db = (db_out if db_out is not None else torch.empty(N, dtype=torch.float32, device=x.device)) if want_db else None
        if want_db:
            p.db = db.data_ptr()
        res = out if out is not None else torch.empty(N, KH * KW, C, dtype=torch.float32, device=x.device)
        p.dw, p.ld_dw = res.data_ptr(), res.shape[2]
        launch()
        return (res, db) if want_db else res
    # slabs: [split][N*taps rows of C | ceil(N / C) more rows holding the slice's N bias-gradient sums]; ONE fold for both
    rows, dbrows = N * KH * KW, ((N + C - 1) // C if want_db else 0)
    slabs = torch.empty(p.split_m, rows + dbrows, C, dtype=torch.float32, device=x.device)
    p.dw = slabs.data_ptr()
    tail = None
    if want_db:
        p.slab_stride = (rows + dbrows) * C
        p.db, p.db_stride = slabs.data_ptr() + rows * C * 4, (rows + dbrows) * C
        tail = torch.empty(dbrows * C, dtype=torch.float32, device=x.device) if db_out is None else db_out
    launch()
    if defer:
        WGRAD_DEFER[-1]["keep"] += (slabs,)
    # (tail entries past N are sums of unwritten slab floats: unused, and never written when the destination is db_out)
    res = fold_rows(slabs, rows, C, out=out, tail_out=tail, deferrable=True, tail_n=(N if db_out is not None else 0))
    res = res if out is not None else res.view(N, KH * KW, C)
    return (res, tail[:N]) if want_db else res


def _im2col_T(x: torch.Tensor, KH: int, KW: int, stride: int, pad: int, ups: int) -> torch.Tensor:
    """x [B,H,W,C] bf16 -> [KH*KW*C, M] (M = B*Hout*Wout), taps-major then channels, matching the packed weight order."""
    B, H, W, C = x.shape
    if ups == 1:
        x = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
        H, W = 2 * H, 2 * W
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    if KH == 1 and KW == 1 and stride == 1 and pad == 0:
        return x.reshape(B * H * W, C).t().contiguous()
    xp = torch.nn.functional.pad(x, (0, 0, pad, pad, pad, pad))
    cols = torch.empty(KH * KW, C, B, Ho, Wo, dtype=x.dtype, device=x.device)
    for ky in range(KH):
        for kx in range(KW):
            v = xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride, :]
            cols[ky * KW + kx].copy_(v.permute(3, 0, 1, 2))
    return cols.reshape(KH * KW * C, B * Ho * Wo)


_PARITY_IDX = {}


def _wgrad_parity(x: torch.Tensor, dy: torch.Tensor, stride: int, ups: int, out: Optional[torch.Tensor], want_db: bool):
    """Weight gradient of the six resampling convolutions of the U-Net (3x3 / stride 2 / pad 1 downsamplers; 3x3 on the
    nearest-x2 upsampled input) through the stride-1 kernel.  Splitting the FINE grid by pixel parity turns either one into
    four stride-1 correlations on the coarse grid:
      stride 2:   x[2*oy + ky - 1] = x_p[oy + s],  (p, s) = (1, -1), (0, 0), (1, 0) for ky = 0, 1, 2   (x_p = rows of parity p)
      upsample:   xu[2*i + a + ky - 1] = x[i + s], s = floor((a + ky - 1) / 2): (-1, 0, 0) for a = 0, (0, 0, 1) for a = 1
    and the same along x, so dW[ky][kx] is a gather (stride 2) or a four-term sum (upsample) of taps of the four 3x3 results.
    One parity-split copy of the fine operand replaces nine im2col copies, a transposed dy and a GEMM on them.  Returns None
    when the coarse geometry is not handled by aptp_conv_wgrad."""
    B, H, W, C = x.shape
    N = dy.shape[-1]
    dev = x.device
    if stride == 2:
        if H % 2 or W % 2 or tuple(dy.shape[1:3]) != (H // 2, W // 2):
            return None
        fine, coarse_other = x, dy
    else:
        if tuple(dy.shape[1:3]) != (2 * H, 2 * W):
            return None
        fine, coarse_other = dy, x
    # the four parity planes of the fine operand, contiguous: [2, 2, B, h, w, ch]
    Bf, Hf, Wf, Cf = fine.shape
    planes = fine.reshape(Bf, Hf // 2, 2, Wf // 2, 2, Cf).permute(2, 4, 0, 1, 3, 5).contiguous()
    key = (stride, ups, str(dev))
    idx = _PARITY_IDX.get(key)
    if idx is None:
        if stride == 2:
            par, sh = (1, 0, 1), (0, 1, 1)               # ky -> (parity of the source row, tap row of the stride-1 result)
            dst = {(p_, q_): [] for p_ in (0, 1) for q_ in (0, 1)}
            src = {(p_, q_): [] for p_ in (0, 1) for q_ in (0, 1)}
            for ky in range(3):
                for kx in range(3):
                    dst[(par[ky], par[kx])].append(ky * 3 + kx)
                    src[(par[ky], par[kx])].append(sh[ky] * 3 + sh[kx])
            idx = {k: (torch.tensor(dst[k], device=dev), torch.tensor(src[k], device=dev)) for k in dst}
        else:
            ty = ((0, 1, 1), (1, 1, 2))                  # output-row parity a, ky -> tap row of the stride-1 result
            idx = {(a, b): torch.tensor([ty[a][ky] * 3 + ty[b][kx] for ky in range(3) for kx in range(3)], device=dev)
                   for a in (0, 1) for b in (0, 1)}
        _PARITY_IDX[key] = idx
    dW = torch.zeros(N, 9, C, dtype=torch.float32, device=dev) if stride != 2 else torch.empty(N, 9, C, dtype=torch.float32, device=dev)
    db = None
    for a in (0, 1):
        for b in (0, 1):
            xs, dys = (planes[a, b], coarse_other) if stride == 2 else (coarse_other, planes[a, b])
            first = (a, b) == (0, 0)
            r = _wgrad_direct(xs, dys, 3, 3, want_db=want_db and (first or stride != 2))
            if r is None:
                return None
            if want_db and (first or stride != 2):
                r, dbp = r
                db = dbp if db is None else db + dbp      # upsample: the four parity planes of dy together are dy
            if stride == 2:
                d_i, s_i = idx[(a, b)]
                dW[:, d_i] = r[:, s_i]
            else:
                dW += r[:, idx[(a, b)]]
    if out is not None:
        out[:, :, :C] = dW
        dW = out
    return (dW, db) if want_db else dW


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, KH: int, KW: int, stride: int = 1, pad: int = 0, ups: int = 0,
               out: Optional[torch.Tensor] = None, want_db: bool = False, db_out: Optional[torch.Tensor] = None):
    """Weight gradient of y = conv(x, w): returns fp32 [N, KH*KW, C] (the packed-weight order, unpadded), or `out` (fp32
    [N, KH*KW, ld >= C], e.g. a zero-initialised packed-layout gradient) with the result in its first C columns.
    x [B,H,W,C] (or tokens [B,L,C] with KH=KW=1), dy [B,Ho,Wo,N] (or [B,L,N]); both bf16.
    want_db: also return the bias gradient (fp32 [N] column sums of dy) -- a by-product of the weight-gradient kernel, which
    stages every dy tile anyway; a separate column-sum pass only on the fallback geometries."""
    if x.dim() == 3:
        # tokens: one image of B*L x 1 "pixels" (a linear layer has no spatial structure)
        x, dy = x.reshape(1, -1, 1, x.shape[-1]), dy.reshape(1, -1, 1, dy.shape[-1])
    C, N = x.shape[-1], dy.shape[-1]
    if WGRAD_KERNEL and stride == 1 and ups == 0 and KH == KW and pad == KH // 2 and x.shape[:3] == dy.shape[:3]:
        g = _wgrad_direct(x, dy, KH, KW, out=out, want_db=want_db, db_out=db_out)
        if g is not None:
            return g
    # the six resampling convolutions of the U-Net (3x3: stride-2 down-samplers, nearest-x2 up-samplers): the same kernel with
    # a strided / up-sampled halo (csrc/wgrad.hip)
    if (WGRAD_KERNEL and WGRAD_RESAMPLE and KH == 3 and KW == 3 and pad == 1 and x.dim() == 4 and x.shape[0] == dy.shape[0]
            and ((stride == 2 and ups == 0 and x.shape[1] == 2 * dy.shape[1] and x.shape[2] == 2 * dy.shape[2])
                 or (stride == 1 and ups == 1 and dy.shape[1] == 2 * x.shape[1] and dy.shape[2] == 2 * x.shape[2]))):
        g = _wgrad_direct(x, dy, KH, KW, out=out, want_db=want_db, stride=stride, ups=ups, db_out=db_out)
        if g is not None:
            return g
    # resampling convolutions: the parity split only pays where the fine grid is large and the weight small -- it runs the
    # nine-tap kernel four times and uses 9 of the 36 tap results (tools/bench_wgrad_resample.py, us, parity vs copies+GEMM:
    # upsample to 64x64 at 640 / 352 channels 449 vs 810 / 259 vs 396; every other resampling layer of SD-2.1 is 1.2-3.5x
    # SLOWER that way, e.g. 704 vs 199 at 8x8 / 1280 channels), so the rule is narrow
    if (WGRAD_KERNEL and WGRAD_PARITY and KH == 3 and KW == 3 and pad == 1 and x.dim() == 4 and stride == 1 and ups == 1
            and dy.shape[0] * dy.shape[1] * dy.shape[2] >= 16384 and x.shape[-1] * dy.shape[-1] <= 640 * 640 and db_out is None):
        g = _wgrad_parity(x, dy, stride, ups, out, want_db)
        if g is not None:
            return g
    xt = _im2col_T(x, KH, KW, stride, pad, ups)                   # [K, M]
    K, M = xt.shape
    dyt = dy.reshape(M, N).t().contiguous()                         # [N, M]
    Mp = round_up(M, BK)
    if Mp != M:                                                     # ragged reduction length (e.g. 4*77 text tokens)
        xt = torch.nn.functional.pad(xt, (0, Mp - M))
        dyt = torch.nn.functional.pad(dyt, (0, Mp - M))
    Kp = round_up(K, 8)
    if Kp != K:
        xt = torch.nn.functional.pad(xt, (0, 0, 0, Kp - K))
    pw = PackedWeight(xt.view(Kp, 1, Mp), None, Kp, Mp, 1, 1)
    Np = round_up(N, 8)
    if Np != N:
        dyt = torch.nn.functional.pad(dyt, (0, 0, 0, Np - N))
    res = conv_gemm(dyt.view(1, Np, 1, Mp), pw, pad=0, out_f32=True, prefetch=False)            # [1, Np, 1, Kp] fp32
    res = res.view(Np, Kp)[:N, :K].reshape(N, KH * KW, C)
    if out is not None:
        out[:, :, :C] = res
        res = out
    if want_db and db_out is not None:
        db_out[:N].copy_(colsum(dy))
        return res, db_out[:N]
    return (res, colsum(dy)) if want_db else res
