"""CPU: the C-ABI library loads and exports every symbol include/qcnet_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from helpers import ROOT

from tempenslearnagg_qcnet_amd import _lib

HEADER = os.path.join(ROOT, 'include', 'qcnet_hip.h')


def _declared():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(qcn_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 10
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), 'libqcnet_hip.so does not export %s' % n
    assert set(names) == set(_lib.EXPORTED_SYMBOLS)


def test_host_only_entry_points():
    lib = _lib.lib()
    assert lib.qcn_version() >= 100
    assert lib.qcn_status_string(0) == b'ok'
    assert b'workspace' in lib.qcn_status_string(-3)
    assert lib.qcn_scan_workspace_bytes(0) > 0
    assert lib.qcn_csr_workspace_bytes(0, 10) > 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libqcnet_hip.so')
    with pytest.raises(_lib.QcnError, match='no fallback'):
        _lib.lib()


def test_ops_refuse_cpu_tensors():
    import torch
    from tempenslearnagg_qcnet_amd import ops
    x = torch.rand(4, 2)
    p = torch.tensor([0, 4])
    with pytest.raises(_lib.QcnError, match='GPU only'):
        ops.radius_pairs(x, x, 1.0, p, p, 8)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'tempenslearnagg-qcnet_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                for line in text.splitlines():
                    if 'import' in line:
                        assert 'oracle' not in line, (dirpath, f, line)
