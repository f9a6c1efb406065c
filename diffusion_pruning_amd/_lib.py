"""ctypes binding of libaptp_hip.so (C ABI declared in include/aptp_hip.h).

The library is the ONLY compute path of this package: there is no PyTorch/CPU fallback.  Loading fails loudly if
the shared object is missing (build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C diffusion_pruning_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (APTP_LIB=<file> loads another build of the same ABI: A/B timing of kernel changes on one box)
LIB_PATH = os.environ.get("APTP_LIB") or os.path.join(_HERE, "csrc", "libaptp_hip.so")

ACT_NONE, ACT_SILU, ACT_GEGLU = 0, 1, 2
TILE_AUTO, TILE_128x128, TILE_128x160, TILE_64x128, TILE_64x160, TILE_128x64, TILE_64x64 = range(7)
(TILE_DMA_128x128, TILE_DMA_128x160, TILE_DMA_64x128, TILE_DMA_64x160, TILE_DMA_128x64, TILE_DMA_64x64) = range(7, 13)
(TILE_DMA3_128x128, TILE_DMA3_128x160, TILE_DMA3_64x128, TILE_DMA3_64x160, TILE_DMA3_128x64, TILE_DMA3_64x64) = range(13, 19)


class ConvGemmParams(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64),
        ("B", c_int32), ("Hin", c_int32), ("Win", c_int32), ("Cin", c_int32),
        ("Hout", c_int32), ("Wout", c_int32),
        ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad", c_int32), ("ups", c_int32),
        ("w", c_void_p),
        ("N", c_int32), ("cin_pad", c_int32),
        ("bias", c_void_p),
        ("rowbias", c_void_p), ("ld_rowbias", c_int32),
        ("colgate", c_void_p), ("gate_group", c_int32), ("gate_B", c_int32),
        ("act", c_int32),
        ("corr", c_void_p), ("corr_B", c_int32),
        ("residual", c_void_p), ("ldres", c_int64),
        ("depth", c_void_p), ("depth_B", c_int32),
        ("depth_in", c_void_p), ("lddin", c_int64),
        ("y", c_void_p), ("ldy", c_int64),
        ("out_f32", c_int32), ("split_k", c_int32),
        ("workspace", c_void_p),
        ("tile", c_int32), ("order", c_int32),
        ("rowstat_out", c_void_p), ("rowstat_slots", c_int32),
        ("ln_stats", c_void_p), ("ln_slots", c_int32),
        ("ln_colsum", c_void_p), ("ln_eps", c_float), ("ln_C", c_int32),
        ("colstat_out", c_void_p), ("colstat_ld", c_int32),
        ("tile_counters", c_void_p),
        ("x2", c_void_p), ("ldx2", c_int64), ("Cin2", c_int32), ("cin2_pad", c_int32),
        ("prefetch", c_void_p), ("prefetch_bytes", c_int64),
        ("epilogue", c_int32),
        ("gn_gamma", c_void_p), ("gn_beta", c_void_p), ("gn_groups", c_int32), ("gn_C", c_int32), ("gn_silu", c_int32),
        ("gn_eps", c_float),
        ("io_f32", c_int32),
        ("ustat_out", c_void_p), ("ustat_unit", c_int32), ("ustat_units", c_int32), ("ustat_nrep", c_int32),
    ]


class GroupNormColStats(Structure):
    _fields_ = [("stats", c_void_p), ("ld", c_int32), ("rows_per_block", c_int32), ("C", c_int32),
                ("ustats", c_void_p), ("unit", c_int32), ("units", c_int32), ("nrep", c_int32)]


class GroupNormParams(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64),
        ("y", c_void_p), ("ldy", c_int64),
        ("B", c_int32), ("HW", c_int32), ("C", c_int32), ("groups", c_int32),
        ("gamma", c_void_p), ("beta", c_void_p),
        ("eps", c_float), ("silu", c_int32),
        ("workspace", c_void_p), ("variant", c_int32),
        ("counters", c_void_p),
        ("colstats", GroupNormColStats * 2),
        ("io_f32", c_int32),
    ]


class LayerNormParams(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64),
        ("y", c_void_p), ("ldy", c_int64),
        ("rows", c_int32), ("C", c_int32),
        ("gamma", c_void_p), ("beta", c_void_p),
        ("eps", c_float),
        ("io_f32", c_int32),
    ]


class AttentionParams(Structure):
    _fields_ = [
        ("q", c_void_p), ("q_stride_b", c_int64), ("q_stride_l", c_int64),
        ("k", c_void_p), ("k_stride_b", c_int64), ("k_stride_l", c_int64),
        ("v", c_void_p), ("v_stride_b", c_int64), ("v_stride_l", c_int64),
        ("o", c_void_p), ("o_stride_b", c_int64), ("o_stride_l", c_int64),
        ("B", c_int32), ("heads", c_int32), ("Lq", c_int32), ("Lk", c_int32),
        ("scale", c_float),
        ("lse", c_void_p),
        ("variant", c_int32),
        ("io_f32", c_int32),
    ]


class GateBwdParams(Structure):
    _fields_ = [
        ("dy", c_void_p), ("lddy", c_int64), ("y0", c_void_p), ("ldy0", c_int64), ("dx", c_void_p), ("lddx", c_int64),
        ("B", c_int32), ("HW", c_int32), ("C", c_int32), ("groups", c_int32),
        ("gate", c_void_p), ("gate_B", c_int32),
        ("dgate_partial", c_void_p),
        ("dgate", c_void_p),
    ]


class GegluParams(Structure):
    _fields_ = [
        ("hg", c_void_p), ("ldhg", c_int64), ("out", c_void_p), ("ldout", c_int64),
        ("dout", c_void_p), ("lddout", c_int64), ("dhg", c_void_p), ("lddhg", c_int64),
        ("B", c_int32), ("HW", c_int32), ("C", c_int32), ("groups", c_int32),
        ("gate", c_void_p), ("gate_B", c_int32),
        ("dgate_partial", c_void_p),
        ("backward", c_int32),
        ("dgate", c_void_p),
    ]


class FfTailParams(Structure):
    _fields_ = [
        ("h", c_void_p), ("ldh", c_int64), ("x", c_void_p), ("ldx", c_int64), ("y", c_void_p), ("ldy", c_int64),
        ("w1", c_void_p), ("b1", c_void_p), ("cs1", c_void_p), ("n1", c_int32), ("ld1", c_int32),
        ("w2", c_void_p), ("b2", c_void_p), ("ld2", c_int32),
        ("w3", c_void_p), ("b3", c_void_p), ("ld3", c_int32),
        ("colstat", c_void_p), ("colstat_ld", c_int32),
        ("M", c_int32), ("C", c_int32),
        ("eps", c_float),
    ]


class WgradParams(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64), ("dy", c_void_p), ("lddy", c_int64), ("dw", c_void_p),
        ("B", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32), ("N", c_int32), ("KH", c_int32), ("KW", c_int32),
        ("split_m", c_int32), ("ld_dw", c_int32), ("db", c_void_p), ("slab_stride", c_int64), ("db_stride", c_int64),
        ("stride", c_int32), ("ups", c_int32),
    ]


class FoldRowsParams(Structure):
    _fields_ = [("partials", c_void_p), ("out", c_void_p), ("R", c_int32), ("n_rows", c_int32), ("C", c_int32), ("ld_out", c_int32),
                ("tail_out", c_void_p), ("tail_rows", c_int32), ("pair_split", c_int32), ("tail_n", c_int32)]


class PackDgradParams(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("N", c_int32), ("C", c_int32), ("taps", c_int32), ("src_ld", c_int32),
                ("dst_ld", c_int32), ("dst_rows", c_int32)]


class DepthLerpParams(Structure):
    _fields_ = [
        ("x_in", c_void_p), ("ld_in", c_int64), ("x_out", c_void_p), ("ld_out", c_int64), ("y", c_void_p), ("ld_y", c_int64),
        ("dy", c_void_p), ("ld_dy", c_int64), ("d_in", c_void_p), ("ld_d_in", c_int64), ("d_out", c_void_p), ("ld_d_out", c_int64),
        ("B", c_int32), ("HW", c_int32), ("C", c_int32),
        ("d", c_void_p), ("d_B", c_int32),
        ("dd_partial", c_void_p),
        ("dd", c_void_p),
        ("backward", c_int32),
    ]


class AdamWItem(Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("shadow", c_void_p), ("n", c_int64)]


class AdamWParams(Structure):
    _fields_ = [("items_dev", c_void_p), ("starts_dev", c_void_p), ("n_items", c_int32), ("total_blocks", c_int32),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float),
                ("step_dev", c_void_p), ("gate_dev", c_void_p), ("grad_scale", c_float)]


class MseParams(Structure):
    _fields_ = [
        ("a", c_void_p), ("lda", c_int64), ("b", c_void_p), ("ldb", c_int64),
        ("rows", c_int64), ("C", c_int32), ("f32", c_int32),
        ("partial", c_void_p), ("out", c_void_p),
        ("g", c_void_p), ("da", c_void_p), ("ldda", c_int64),
        ("backward", c_int32),
    ]


class GroupNormBwdParams(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64), ("dy", c_void_p), ("lddy", c_int64), ("dx", c_void_p), ("lddx", c_int64),
        ("B", c_int32), ("HW", c_int32), ("C", c_int32), ("groups", c_int32),
        ("gamma", c_void_p), ("beta", c_void_p),
        ("eps", c_float), ("silu", c_int32),
        ("fwd_stats", c_void_p),
        ("workspace", c_void_p),
        ("pgrad_partial", c_void_p),
        ("add", c_void_p), ("ldadd", c_int64),
    ]


class UnetPrologueParams(Structure):
    _fields_ = [
        ("sample", c_void_p), ("sample_bf16", c_int32), ("x", c_void_p),
        ("B", c_int32), ("C", c_int32), ("H", c_int32), ("W", c_int32), ("cin_pad", c_int32),
        ("timesteps", c_void_p), ("freqs", c_void_p), ("half", c_int32), ("t_emb", c_void_p),
    ]


class UnetEpilogueParams(Structure):
    _fields_ = [
        ("y", c_void_p), ("ldy", c_int64), ("out", c_void_p), ("out_bf16", c_int32),
        ("B", c_int32), ("C", c_int32), ("H", c_int32), ("W", c_int32),
    ]


class ColsumParams(Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_int64), ("rows", c_int32), ("C", c_int32), ("partial", c_void_p), ("batch", c_int32)]


class LayerNormPgradParams(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64), ("dy", c_void_p), ("lddy", c_int64),
        ("rows", c_int32), ("C", c_int32), ("eps", c_float), ("partial", c_void_p),
    ]


class LayerNormBwdParams(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64), ("dy", c_void_p), ("lddy", c_int64), ("dx", c_void_p), ("lddx", c_int64),
        ("rows", c_int32), ("C", c_int32),
        ("gamma", c_void_p),
        ("eps", c_float),
        ("add", c_void_p), ("ldadd", c_int64),
    ]


class AttentionBwdParams(Structure):
    _fields_ = [
        ("q", c_void_p), ("q_stride_b", c_int64), ("q_stride_l", c_int64),
        ("k", c_void_p), ("k_stride_b", c_int64), ("k_stride_l", c_int64),
        ("v", c_void_p), ("v_stride_b", c_int64), ("v_stride_l", c_int64),
        ("o", c_void_p), ("o_stride_b", c_int64), ("o_stride_l", c_int64),
        ("dout", c_void_p), ("dout_stride_b", c_int64), ("dout_stride_l", c_int64),
        ("dq", c_void_p), ("dq_stride_b", c_int64), ("dq_stride_l", c_int64),
        ("dk", c_void_p), ("dk_stride_b", c_int64), ("dk_stride_l", c_int64),
        ("dv", c_void_p), ("dv_stride_b", c_int64), ("dv_stride_l", c_int64),
        ("lse", c_void_p), ("delta", c_void_p),
        ("B", c_int32), ("heads", c_int32), ("Lq", c_int32), ("Lk", c_int32),
        ("scale", c_float), ("q_split", c_int32), ("workspace", c_void_p),
    ]


# every symbol include/aptp_hip.h declares: (name, restype, argtypes)
EXPORTS = [
    ("aptp_conv_gemm", c_int, [POINTER(ConvGemmParams), c_void_p]),
    ("aptp_conv_gemm_workspace_bytes", c_int64, [POINTER(ConvGemmParams)]),
    ("aptp_conv_gemm_suggest_split_k", c_int, [POINTER(ConvGemmParams)]),
    ("aptp_conv_gemm_rowstat_slots", c_int, [POINTER(ConvGemmParams)]),
    ("aptp_conv_gemm_tiles", c_int, [POINTER(ConvGemmParams)]),
    ("aptp_conv_gemm_colstat_rows", c_int, [POINTER(ConvGemmParams)]),
    ("aptp_groupnorm", c_int, [POINTER(GroupNormParams), c_void_p]),
    ("aptp_groupnorm_nchunk", c_int, [c_int]),
    ("aptp_rows_nchunk", c_int, [c_int]),
    ("aptp_groupnorm_workspace_bytes", c_int64, [POINTER(GroupNormParams)]),
    ("aptp_layernorm", c_int, [POINTER(LayerNormParams), c_void_p]),
    ("aptp_attention", c_int, [POINTER(AttentionParams), c_void_p]),
    ("aptp_gate_bwd", c_int, [POINTER(GateBwdParams), c_void_p]),
    ("aptp_geglu", c_int, [POINTER(GegluParams), c_void_p]),
    ("aptp_depth_lerp", c_int, [POINTER(DepthLerpParams), c_void_p]),
    ("aptp_ff_tail_supported", c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    ("aptp_ff_tail", c_int, [POINTER(FfTailParams), c_void_p]),
    ("aptp_conv_wgrad_supported", c_int, [POINTER(WgradParams)]),
    ("aptp_conv_wgrad_suggest_split", c_int, [POINTER(WgradParams)]),
    ("aptp_conv_wgrad", c_int, [POINTER(WgradParams), c_void_p]),
    ("aptp_conv_wgrad_many_item_bytes", ctypes.c_int64, []),
    ("aptp_conv_wgrad_many_blocks", c_int, [POINTER(WgradParams)]),
    ("aptp_conv_wgrad_many_fill", c_int, [POINTER(WgradParams), c_void_p, ctypes.c_int32]),
    ("aptp_conv_wgrad_many", c_int, [c_void_p, c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_void_p]),
    ("aptp_fold_rows", c_int, [POINTER(FoldRowsParams), c_void_p]),
    ("aptp_pack_dgrad", c_int, [POINTER(PackDgradParams), c_void_p]),
    ("aptp_fold_rows_blocks", c_int, [POINTER(FoldRowsParams)]),
    ("aptp_fold_rows_many", c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    ("aptp_pack_dgrad_blocks", c_int, [POINTER(PackDgradParams)]),
    ("aptp_pack_dgrad_many", c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    ("aptp_adamw_blocks", c_int, [c_int64]),
    ("aptp_adamw_many", c_int, [POINTER(AdamWParams), c_void_p]),
    ("aptp_mse_nblocks", c_int, [c_int64, c_int32]),
    ("aptp_mse", c_int, [POINTER(MseParams), c_void_p]),
    ("aptp_groupnorm_bwd", c_int, [POINTER(GroupNormBwdParams), c_void_p]),
    ("aptp_layernorm_bwd", c_int, [POINTER(LayerNormBwdParams), c_void_p]),
    ("aptp_attention_bwd", c_int, [POINTER(AttentionBwdParams), c_void_p]),
    ("aptp_attention_bwd_q_split", c_int, [POINTER(AttentionBwdParams)]),
    ("aptp_attention_bwd_workspace_bytes", c_size_t, [POINTER(AttentionBwdParams), c_int32]),
    ("aptp_colsum", c_int, [POINTER(ColsumParams), c_void_p]),
    ("aptp_layernorm_pgrad", c_int, [POINTER(LayerNormPgradParams), c_void_p]),
    ("aptp_unet_prologue", c_int, [POINTER(UnetPrologueParams), c_void_p]),
    ("aptp_unet_epilogue", c_int, [POINTER(UnetEpilogueParams), c_void_p]),
    ("aptp_last_error", c_char_p, []),
    ("aptp_version", c_int, []),
]

_lib = None


class AptpError(RuntimeError):
    pass


def load():
    """Load libaptp_hip.so (once) and declare the prototypes.  Raises if the library is missing: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles the HIP runtime (torch/lib/libamdhip64.so), and the process must end up with ONE
    # runtime.  Loaded after torch, libaptp_hip.so's libamdhip64.so.7 dependency resolves to the copy torch already
    # mapped; loaded before it, /opt/rocm's copy comes in as a second runtime and every launch on a torch stream fails with
    # "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AptpError(
            f"{LIB_PATH} not found: the HIP extension is the only compute path of diffusion_pruning_amd. "
            "Build it with `make -C diffusion_pruning_amd/csrc` (hipcc, --offload-arch=gfx950).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in EXPORTS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().aptp_last_error()
        raise AptpError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
