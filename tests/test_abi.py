"""CPU checks of the C-ABI boundary: the library loads, exports every symbol include/aptp_hip.h declares, the ctypes
mirrors have the C compiler's layout, and argument validation fails loudly without launching anything."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aptp_hip.h")


@pytest.fixture(scope="module")
def lib():
    from diffusion_pruning_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aptp_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from diffusion_pruning_amd import _lib
    names = declared_functions()
    assert len(names) >= 10
    bound = {n for n, _, _ in _lib.EXPORTS}
    assert set(names) == bound, set(names) ^ bound
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.aptp_version() >= 100


def test_ctypes_layout_matches_the_c_header():
    from diffusion_pruning_amd import _lib
    structs = {"AptpConvGemmParams": _lib.ConvGemmParams, "AptpGroupNormParams": _lib.GroupNormParams,
               "AptpLayerNormParams": _lib.LayerNormParams, "AptpAttentionParams": _lib.AttentionParams,
               "AptpGateBwdParams": _lib.GateBwdParams, "AptpGegluParams": _lib.GegluParams,
               "AptpGroupNormBwdParams": _lib.GroupNormBwdParams, "AptpLayerNormBwdParams": _lib.LayerNormBwdParams,
               "AptpAttentionBwdParams": _lib.AttentionBwdParams, "AptpColsumParams": _lib.ColsumParams,
               "AptpLayerNormPgradParams": _lib.LayerNormPgradParams, "AptpDepthLerpParams": _lib.DepthLerpParams,
               "AptpWgradParams": _lib.WgradParams,
               "AptpFfTailParams": _lib.FfTailParams, "AptpFoldRowsParams": _lib.FoldRowsParams,
               "AptpPackDgradParams": _lib.PackDgradParams, "AptpMseParams": _lib.MseParams,
               "AptpAdamWItem": _lib.AdamWItem, "AptpAdamWParams": _lib.AdamWParams}
    body = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, cls in structs.items():
        body.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            body.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    body.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write("\n".join(body))
        subprocess.run(["gcc", "-std=c99", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_validation_errors_do_not_launch(lib):
    from diffusion_pruning_amd import _lib
    p = _lib.ConvGemmParams()
    rc = lib.aptp_conv_gemm(ctypes.byref(p), None)
    assert rc == -1 and b"null pointer" in lib.aptp_last_error()
    g = _lib.GroupNormParams()
    assert lib.aptp_groupnorm(ctypes.byref(g), None) == -1
    a = _lib.AttentionParams()
    assert lib.aptp_attention(ctypes.byref(a), None) == -1
    ln = _lib.LayerNormParams()
    assert lib.aptp_layernorm(ctypes.byref(ln), None) == -1
    assert lib.aptp_groupnorm_nchunk(4096) == 128 and lib.aptp_groupnorm_nchunk(4) == 1


def test_missing_library_fails_loudly(monkeypatch):
    from diffusion_pruning_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libaptp_hip.so")
    with pytest.raises(_lib.AptpError):
        _lib.load()
