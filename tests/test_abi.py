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
    d = _lib.ConvDesc(1, 4, 4, 24, 4, 4, 8, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0)
    rc = lib.diffsal_conv_igemm(ctypes.byref(d), 16, 16, None, None, None, None, None, 16, None, 0, None)
    assert rc == -1 and b"multiple of 32" in lib.diffsal_last_error()
    rc = lib.diffsal_layernorm(None, None, None, None, 4, 32, 1e-5, None)
    assert rc == -4
    assert lib.diffsal_groupnorm_ws_bytes(4, 32) == 4 * 32 * 32 * 2 * 8
