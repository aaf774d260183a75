"""CPU checks of the drop-in boundary: libsup3r_hip.so loads without a GPU and
exports every function include/sup3r_hip.h declares; the ctypes structs match
the C layout; no compute is called here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'sup3r_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(s3_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from sup3r_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libsup3r_hip.so not built (run __graft_entry__.build())')
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f'{n} declared in sup3r_hip.h but not exported'
    assert sorted(_lib.EXPORTS) == names
    assert b'gfx950' in L.s3_version()


def test_struct_layouts():
    from sup3r_amd import _lib
    assert C.sizeof(_lib.TensorDesc) == 40
    # 7 ints + 4*3 ints + pad_mode, act + float + d2s, rep, bcast + 4 reserved
    assert C.sizeof(_lib.OpDesc) == 4 * (7 + 12 + 2 + 1 + 3 + 4)


def test_missing_library_fails_loudly(monkeypatch):
    from sup3r_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libsup3r_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.lib()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from sup3r_amd.engine import Device
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        Device(0)


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle."""
    pkg = os.path.join(ROOT, 'sup3r_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src,
                                     flags=re.M), f
