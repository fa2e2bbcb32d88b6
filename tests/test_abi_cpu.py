"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/dclip_hip.h declares."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(REPO, "include", "dclip_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dclip_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from dclip_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES), "include/dclip_hip.h and dclip_amd/_lib.py differ"


def test_library_exports_every_declared_symbol():
    from dclip_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} not exported"
    assert _lib.load().dclip_abi_version() == 1


def test_argument_errors_do_not_need_a_gpu():
    from dclip_amd import _lib
    lib = _lib.load()
    rc = lib.dclip_gemm_f32(None, None, None, None, None, None, 4, 4, 4, 4, 4, 4, 3, 0, 1.0, 0, None, 0, None)
    assert rc == -1
    assert b"null operand" in lib.dclip_last_error()
    with pytest.raises(_lib.DclipError):
        _lib.check(rc, "gemm")


def test_product_path_does_not_import_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "dclip_amd")):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M):
                    bad.append(f)
    assert not bad, f"product modules import the oracle: {bad}"
