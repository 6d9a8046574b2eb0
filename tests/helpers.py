"""Shared test utilities (the only place, with bench.py's cpu_baseline leg and smoke(), that touches oracle/)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import qcnet_oracle as oracle  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
_ORACLE_SO = os.path.join(ROOT, 'oracle', '_build', 'liboracle.so')


def oracle_clib():
    if not os.path.exists(_ORACLE_SO):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    lib = ctypes.CDLL(_ORACLE_SO)
    P, I64 = ctypes.c_void_p, ctypes.c_int64
    lib.qcn_oracle_radius.restype = I64
    lib.qcn_oracle_radius.argtypes = [P, P, P, P, I64, I64, ctypes.c_float, I64, P, P, I64]
    lib.qcn_oracle_attn.restype = None
    lib.qcn_oracle_attn.argtypes = [P, P, P, P, P, P, P, I64, I64, ctypes.c_int, ctypes.c_int, P, P, P]
    return lib


def c_radius(x, y, r, ptr_x, ptr_y, max_num_neighbors):
    """oracle/radius_ref.c: [2, E] int64 (row0 = y, row1 = x)."""
    lib = oracle_clib()
    x = np.ascontiguousarray(x.cpu().numpy(), dtype=np.float32)
    y = np.ascontiguousarray(y.cpu().numpy(), dtype=np.float32)
    px = np.ascontiguousarray(ptr_x.cpu().numpy(), dtype=np.int64)
    py = np.ascontiguousarray(ptr_y.cpu().numpy(), dtype=np.int64)
    cap = int(y.shape[0]) * int(max_num_neighbors) + 1
    row = np.empty(cap, dtype=np.int64)
    col = np.empty(cap, dtype=np.int64)
    r2 = np.float32(np.float64(r) * np.float64(r))
    n = lib.qcn_oracle_radius(x.ctypes.data, y.ctypes.data, px.ctypes.data, py.ctypes.data, len(px) - 1, x.shape[1],
                              ctypes.c_float(r2), int(max_num_neighbors), row.ctypes.data, col.ctypes.data, cap)
    assert n >= 0
    return torch.from_numpy(np.stack([row[:n], col[:n]]))


def c_attention(q, k, v, kr, vr, edge_index):
    """oracle/radius_ref.c::qcn_oracle_attn — fp64 reference of one propagate call, returns float64 [n_dst, 128]."""
    lib = oracle_clib()
    f = lambda t: None if t is None else np.ascontiguousarray(t.cpu().numpy(), dtype=np.float32)
    qn, kn, vn, krn, vrn = f(q), f(k), f(v), f(kr), f(vr)
    src = np.ascontiguousarray(edge_index[0].cpu().numpy(), dtype=np.int64)
    dst = np.ascontiguousarray(edge_index[1].cpu().numpy(), dtype=np.int64)
    nd = qn.shape[0]
    smax = np.empty(nd * 8, dtype=np.float64)
    ssum = np.empty(nd * 8, dtype=np.float64)
    out = np.zeros((nd, 128), dtype=np.float64)
    p = lambda a: None if a is None else a.ctypes.data
    lib.qcn_oracle_attn(p(qn), p(kn), p(vn), p(krn), p(vrn), p(src), p(dst), len(src), nd, 8, 16, p(smax), p(ssum),
                        p(out))
    return torch.from_numpy(out)


def randomize_(module: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    """weight_init leaves biases at 0 and LayerNorms at identity, which hides bias / affine bugs: move every
    1-D parameter off its trivial value and widen the Fourier frequencies (deterministic)."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in module.named_parameters():
            if prm.dim() == 1:
                prm.add_(torch.randn(prm.shape, generator=gen).to(prm.device) * 0.1)
            elif 'freqs' in name:
                prm.copy_(torch.randn(prm.shape, generator=gen).to(prm.device) * 0.05)
            elif 'traj_emb_h0' in name:
                prm.copy_(torch.randn(prm.shape, generator=gen).to(prm.device) * 0.1)
    return module


def cpu_state_dict(module: torch.nn.Module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    """max |got - want| / max |want|  — the "1e-4 relative" bar of BASELINE.json is applied to this."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    denom = float(want.abs().max()) if want.numel() else 1.0
    return float((got - want).abs().max()) / max(denom, 1e-12) if want.numel() else 0.0


ATOL_SCALE = 2e-6          # predictions (trajectories, scales, logits, probabilities)
ATOL_SCALE_HIDDEN = 5e-6   # 128-wide hidden states (x_a, x_pl, m, final_m): see assert_close
HIDDEN_KEYS = ('m', 'final_m', 'x_a', 'x_pl')


def assert_close(got: torch.Tensor, want: torch.Tensor, name: str = '', rtol: float = 1e-4, atol: float = None) -> float:
    """Per-element parity bar of the model-level tests:  |got - want| <= rtol |want| + atol  for EVERY
    element, rtol = 1e-4 (BASELINE.json: "within 1e-4 relative").  ``atol`` is the absolute floor for elements near zero;
    default ATOL_SCALE * max|want| = 2e-6 of the tensor's scale — fp32 dot products over 128 terms of that scale carry an
    absolute rounding error of about that size whatever the size of their result, so a purely relative bound on an element
    that happens to cancel to ~0 is not meaningful.  (The old bar, max|got - want| / max|want| < 1e-4, is 50x looser on
    such elements.)  Finding of round 2 (full-size runs): every prediction tensor sits at <= 0.65 of this bar; the hidden
    mode embeddings ``m`` (outputs of 28 stacked attention layers + the GRU, |m| up to ~10) have elements near zero whose
    absolute error reaches 3.4e-5 = 4e-6 of the tensor's scale (max-norm error 4e-6 .. 7e-6), so hidden states
    (``HIDDEN_KEYS``) get the floor ATOL_SCALE_HIDDEN = 5e-6 of their scale.
    Returns the worst ratio |got - want| / (rtol |want| + atol)."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (name, tuple(got.shape), tuple(want.shape))
    if want.numel() == 0:
        return 0.0
    assert bool(torch.isfinite(got).all()), name
    if atol is None:
        atol = (ATOL_SCALE_HIDDEN if name in HIDDEN_KEYS else ATOL_SCALE) * float(want.abs().max())
    bound = rtol * want.abs() + atol
    ratio = (got - want).abs() / bound.clamp_min(1e-300)
    worst = float(ratio.max())
    if worst > 1.0:
        i = int(ratio.argmax())
        raise AssertionError('%s: element %d got %.9g want %.9g: |diff| %.3g > rtol |want| + atol = %.3g (x%.2f; max-norm rel err %.3g)'
                             % (name, i, float(got.reshape(-1)[i]), float(want.reshape(-1)[i]),
                                float((got - want).abs().reshape(-1)[i]), float(bound.reshape(-1)[i]), worst,
                                rel_err(got, want)))
    return worst


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
