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


def test_every_entry_point_is_a_registered_torch_op():
    """north_star / SURVEY 8(b): the kernels are driven through PyTorch custom ops.  Each op is visible to the dispatcher with
    a schema and its CPU key refuses instead of falling back (the fake implementations are exercised on the GPU box:
    tests/test_gpu_torch_ops.py::test_ops_trace_under_fake_tensors)."""
    import torch
    from tempenslearnagg_qcnet_amd import ops
    assert len(ops.OP_NAMES) >= 20
    for name in ops.OP_NAMES:
        packet = getattr(torch.ops.qcnet_hip, name)
        assert str(packet.default._schema).startswith('qcnet_hip::%s(' % name)
        assert torch._C._dispatch_has_kernel_for_dispatch_key('qcnet_hip::' + name, 'CUDA')
        assert torch._C._dispatch_has_kernel_for_dispatch_key('qcnet_hip::' + name, 'CPU')      # the refusing kernel
    with pytest.raises(_lib.QcnError, match='no CPU fallback'):
        torch.ops.qcnet_hip.split_rows(torch.zeros(2, 128), 11)


def test_matmul_mode_is_python_state_not_library_state():
    from tempenslearnagg_qcnet_amd import ops
    handle = ctypes.CDLL(_lib.LIB_PATH)
    assert not hasattr(handle, 'qcn_set_matmul_mode') and not hasattr(handle, 'qcn_prof_enable')
    mode = ops.matmul_mode()
    try:
        ops.set_matmul_mode(ops.MATMUL_BF16)
        assert ops.matmul_mode() == ops.MATMUL_BF16
        with pytest.raises(ValueError):
            ops.set_matmul_mode(7)
    finally:
        ops.set_matmul_mode(mode)


def test_pack_chain_matrix_bf16_layout():
    """bf16 pack (include/qcnet_hip.h 'Chain-packed 128x128 matrix'): P[j][t][lane][i] = bf16(W[32 t + (lane & 31)][kin(8 j + i, lane >> 5)])."""
    import torch
    from tempenslearnagg_qcnet_amd import ops
    w = torch.randn(128, 128, generator=torch.Generator().manual_seed(1))
    packed, sc = ops.pack_chain_matrix(w, mode=ops.MATMUL_BF16)
    assert packed.numel() == 128 * 128 and sc.tolist() == [1.0, 1.0]
    frag = packed[:8192].view(torch.bfloat16).view(8, 4, 64, 8)
    kin = ops.chain_feature_order()
    for j, t, lane, i in ((0, 0, 0, 0), (3, 2, 37, 5), (7, 3, 63, 7), (5, 1, 31, 2)):
        assert frag[j, t, lane, i] == w[32 * t + (lane & 31), kin[8 * j + i, lane >> 5]].bfloat16()
    assert bool((packed[8192:] == 0).all())
