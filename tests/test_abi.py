"""CPU checks of the drop-in boundary: the shared library loads and exports every symbol include/zonos_hip.h
declares, the ctypes binding covers them all, and argument errors come back as statuses (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from zonos_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "zonos_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from zonos_amd import build
    build.build(verbose=False)
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in zonos_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)


def test_abi_version_and_kv_bytes(lib):
    assert lib.zn_abi_version() == _lib.ZN_ABI_VERSION
    zc = _lib.zn_config(d_model=2048, n_layer=26, n_heads=16, n_heads_kv=4, d_ff=8192, n_codebooks=9, vocab_head=1025,
                        vocab_embed=1032, eos_id=1024, mask_id=1025, rope_positions=16384, double_out_proj=1, norm_eps=1e-5)
    # 53 248 B per row-position over 26 layers
    assert lib.zn_kv_bytes_per_layer(C.byref(zc), 1, 1) * 26 == 53248


def test_bad_arguments_return_status_not_crash(lib):
    h = C.c_void_p()
    zc = _lib.zn_config(d_model=100, n_layer=1, n_heads=3, n_heads_kv=1, d_ff=64, n_codebooks=9, vocab_head=1025,
                        vocab_embed=1032, eos_id=1024, mask_id=1025, rope_positions=16384, double_out_proj=1, norm_eps=1e-5)
    w = _lib.zn_weights()
    assert lib.zn_create(C.byref(zc), C.byref(w), 2, C.byref(h)) < 0
    assert b"divide" in lib.zn_last_error(None)
    assert lib.zn_decode_steps(None, 1, None) < 0
    assert lib.zn_destroy(None) == 0
