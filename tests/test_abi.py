"""The C-ABI library loads and exports every symbol include/diffsal.h declares (no compute calls)."""
import ctypes
import os
import re

from diff_sal_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "diffsal.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(diffsal_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in diffsal.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.diffsal_version() >= 1


def test_conv_desc_layout_matches_header():
    text = open(os.path.join(ROOT, "include", "diffsal.h")).read()
    body = re.search(r"typedef struct diffsal_conv_desc \{(.*?)\} diffsal_conv_desc;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int ([^;]+);", body) for f in decl.split(",")]
    assert fields == [n for n, _ in _lib.ConvDesc._fields_]
    assert ctypes.sizeof(_lib.ConvDesc) == 4 * len(fields)


def test_argument_errors_are_reported_without_a_gpu():
    """Validation happens before any launch, so it is checkable on a CPU-only host."""
    lib = _lib.load()
    d = _lib.ConvDesc(1, 4, 4, 24, 4, 4, 8, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0)
    rc = lib.diffsal_conv_igemm(ctypes.byref(d), 16, 16, None, None, None, None, None, 16, None, 0, None)
    assert rc == -1 and b"multiple of 32" in lib.diffsal_last_error()
    # arithmetic mode and storage type are per-call descriptor fields, validated before any launch
    d = _lib.ConvDesc(1, 4, 4, 32, 4, 4, 8, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, 0, 7, 0)
    assert lib.diffsal_conv_igemm(ctypes.byref(d), 16, 16, None, None, None, None, None, 16, None, 0, None) == -4
    assert b"precision" in lib.diffsal_last_error()
    d = _lib.ConvDesc(1, 4, 4, 32, 4, 4, 8, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0, 0, 1, 1)   # bf16x3 arithmetic on bf16 storage
    assert lib.diffsal_conv_igemm(ctypes.byref(d), 16, 16, None, None, None, None, None, 16, None, 0, None) == -4
    rc = lib.diffsal_layernorm(None, None, None, None, 4, 32, 1e-5, 0, None)
    assert rc == -4
    assert lib.diffsal_layernorm(16, 16, 16, 16, 4, 32, 1e-5, 9, None) == -4 and b"dtype" in lib.diffsal_last_error()
    assert not hasattr(lib, "diffsal_set_gemm_precision")      # no process-wide mode in the library
    assert lib.diffsal_groupnorm_ws_bytes(4, 32) == 4 * 128 * 32 * 2 * 8      # sized for the most chunks the tuning knob allows


def test_integration_md_binding_matches_the_header():
    """INTEGRATION.md shows the ctypes struct a maintainer would copy: it must list exactly the header's fields."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class ConvDesc\(ctypes\.Structure\):.*?_fields_ = \[\(n, ctypes\.c_int\) for n in \((.*?)\)\]", md, re.S)
    assert m, "INTEGRATION.md no longer shows the ConvDesc binding"
    names = re.findall(r'"([A-Za-z_]+)"', m.group(1))
    assert names == [n for n, _ in _lib.ConvDesc._fields_]
    call = re.search(r"d = ConvDesc\(([^)]*)\)", md).group(1)
    assert len([a for a in call.split(",") if a.strip()]) == len(names)


def test_abi_version_guard():
    assert _lib.load().diffsal_version() == _lib.ABI_VERSION


def test_workspace_bytes_is_the_typed_functions_behind_one_entry_point():
    """SURVEY 8b names ONE `diffsal_workspace_bytes(op, dims...)`: it must return what the typed *_ws_bytes functions return
    (host arithmetic only: no GPU needed), and 0 for an unknown operator or a wrong argument count."""
    lib = _lib.load()
    d = _lib.ConvDesc(4, 28, 48, 384, 28, 48, 384, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0)
    ref = ctypes.byref(d)
    assert _lib.workspace_bytes(_lib.WS_GROUPNORM, dims=(4, 32)) == lib.diffsal_groupnorm_ws_bytes(4, 32) > 0
    assert _lib.workspace_bytes(_lib.WS_CONV_IGEMM, d) == lib.diffsal_conv_igemm_ws_bytes(ref)
    assert _lib.workspace_bytes(_lib.WS_CONV_WINO, d) == lib.diffsal_conv_wino_ws_bytes(ref) > 0
    assert _lib.workspace_bytes(_lib.WS_CONV_WINO4, d) == lib.diffsal_conv_wino4_ws_bytes(ref) == 36 * 4 * 7 * 12 * (384 + 384) * 4
    assert _lib.workspace_bytes(_lib.WS_CONV_WINO4_STATS, d, (32,)) == lib.diffsal_conv_wino4_stats_bytes(ref, 32) > 0
    assert _lib.workspace_bytes(_lib.WS_CONV_WGRAD, d) == lib.diffsal_conv_wgrad_ws_bytes(ref)
    assert _lib.workspace_bytes(_lib.WS_WGRAD_SEGMENTED, dims=(36, 336, 192, 18)) == lib.diffsal_wgrad_segmented_ws_bytes(36, 336, 192, 18)
    assert _lib.workspace_bytes(_lib.WS_TAPSUM_BWD, dims=(36, 96, 96, 28)) == lib.diffsal_tapsum_bwd_ws_bytes(36, 96, 96, 28) > 0
    assert _lib.workspace_bytes(_lib.WS_SALIENCY_METRICS, dims=(4,)) == lib.diffsal_saliency_metrics_ws_bytes(4) > 0
    assert _lib.workspace_bytes(_lib.WS_ATTENTION_TAIL, dims=(4, 4, 2352, 296, 96)) == 4 * lib.diffsal_attention_general_tail_floats(4, 4, 2352, 296, 96)
    assert _lib.workspace_bytes(99, d) == 0 and _lib.workspace_bytes(_lib.WS_GROUPNORM, dims=(4,)) == 0
