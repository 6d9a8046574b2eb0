"""PyTorch custom ops over the C ABI (include/qcnet_hip.h).

Every kernel entry point is registered with the dispatcher as ``torch.ops.qcnet_hip.<name>`` (``torch.library``):
* the ``CUDA``-key implementation (ROCm tensors carry that key) validates shapes / dtypes / devices on the host before a
  kernel sees a pointer (a kernel that faults can take the whole GPU host down), allocates outputs and workspaces with
  torch (plumbing only) and launches through ``libqcnet_hip.so`` on torch's current stream;
* a fake (meta) implementation gives output shapes, so the ops trace under FakeTensor / ``torch.compile``;
* the ``CPU`` key REFUSES loudly: the hot path has no CPU or eager fallback (the CPU restatement lives in ``oracle/`` and
  is test infrastructure only).
The functions of this module are the typed Python face of those ops (defaults, argument normalisation) and the host-side
weight packers.  The arithmetic mode of the MFMA chain kernels is an ARGUMENT of every call (the library keeps no state);
``matmul_mode()`` / ``set_matmul_mode()`` hold the process default the layers use when none is given.
"""
import ctypes
import os
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

C = 128  # QCN_CHANNELS

_LIBRARY = torch.library.Library('qcnet_hip', 'DEF')
OP_NAMES: List[str] = []


def _register(name: str, schema: str, fake=None):
    """Decorator: define ``qcnet_hip::<name><schema>``, register the decorated function for the CUDA key, a refusing CPU
    kernel, and ``fake`` as the meta implementation."""
    def deco(fn):
        _LIBRARY.define(name + schema)
        _LIBRARY.impl(name, fn, 'CUDA')

        def refuse(*args, **kwargs):
            raise _lib.QcnError('qcnet_hip::%s got CPU tensors: the QCNet hot path runs on the GPU only (no CPU fallback)'
                                % name)
        _LIBRARY.impl(name, refuse, 'CPU')
        if fake is not None:
            torch.library.register_fake('qcnet_hip::' + name, fake, lib=_LIBRARY)
        OP_NAMES.append(name)
        return fn
    return deco


def _op(name: str):
    return getattr(torch.ops.qcnet_hip, name)


def _dyn():
    return torch.library.get_ctx().new_dynamic_size()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device: torch.device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require(t: torch.Tensor, name: str, dtype: torch.dtype, device: torch.device, shape=None) -> None:
    if not torch.is_tensor(t):
        raise TypeError('%s must be a tensor' % name)
    if not t.is_cuda:
        raise _lib.QcnError('%s is on %s: the QCNet hot path runs on the GPU only (no CPU fallback)' % (name, t.device))
    if t.device != device:
        raise ValueError('%s is on %s, expected %s' % (name, t.device, device))
    if t.dtype != dtype:
        raise TypeError('%s has dtype %s, expected %s' % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and int(d) != int(s) for d, s in zip(t.shape, shape)):
            raise ValueError('%s has shape %s, expected %s' % (name, tuple(t.shape), tuple(shape)))


# ------------------------------------------------------------------------------------------------
# radius search
# ------------------------------------------------------------------------------------------------
def _radius_pairs_fake(x, y, r, ptr_x, ptr_y, max_num_neighbors, drop_self, mask_x, mask_y, y_first):
    return x.new_empty((2, _dyn()), dtype=torch.int64), x.new_empty((y.size(0) + 1,), dtype=torch.int64)


def radius_pairs(x: torch.Tensor, y: torch.Tensor, r: float, ptr_x: torch.Tensor, ptr_y: torch.Tensor,
                 max_num_neighbors: int, drop_self: bool = False, mask_x: Optional[torch.Tensor] = None,
                 mask_y: Optional[torch.Tensor] = None, y_first: bool = True
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """All (y, x) pairs of the same example with ||x - y||^2 < r^2 (qcn_radius_count / _fill).

    Returns ``(edge_index [2, E] int64, offset [n_y + 1] int64)``; ``edge_index`` rows are (y, x) when
    ``y_first`` else (x, y); pairs are grouped by ascending y, ``offset`` is the CSR row pointer over y.
    Costs one host synchronisation (E must be known to allocate the output).
    """
    return _op('radius_pairs')(x, y, float(r), ptr_x, ptr_y, int(max_num_neighbors), bool(drop_self), mask_x, mask_y,
                               bool(y_first))


@_register('radius_pairs', '(Tensor x, Tensor y, float r, Tensor ptr_x, Tensor ptr_y, int max_num_neighbors, bool drop_self, '
           'Tensor? mask_x, Tensor? mask_y, bool y_first) -> (Tensor, Tensor)', _radius_pairs_fake)
def _radius_pairs(x, y, r, ptr_x, ptr_y, max_num_neighbors, drop_self, mask_x, mask_y, y_first):
    dev = x.device
    dim = x.size(1) if x.dim() == 2 else -1
    _require(x, 'x', torch.float32, dev, (None, dim))
    _require(y, 'y', torch.float32, dev, (None, dim))
    if dim not in (2, 3):
        raise _lib.QcnError('radius search supports 2-D or 3-D points, got dim=%d' % dim)
    num_examples = ptr_x.numel() - 1
    _require(ptr_x, 'ptr_x', torch.int64, dev, (num_examples + 1,))
    _require(ptr_y, 'ptr_y', torch.int64, dev, (num_examples + 1,))
    n_x, n_y = x.size(0), y.size(0)
    mx = my = None
    if mask_x is not None:
        mx = mask_x.view(torch.uint8) if mask_x.dtype == torch.bool else mask_x
        _require(mx, 'mask_x', torch.uint8, dev, (n_x,))
    if mask_y is not None:
        my = mask_y.view(torch.uint8) if mask_y.dtype == torch.bool else mask_y
        _require(my, 'mask_y', torch.uint8, dev, (n_y,))
    if drop_self and n_x != n_y:
        raise ValueError('drop_self needs x and y to be the same point set')
    r2 = float(torch.tensor(float(r) * float(r), dtype=torch.float64).float())  # (scalar_t)(r * r), as upstream
    lib = _lib.lib()
    st = _stream(dev)
    count = torch.zeros(n_y + 1, dtype=torch.int32, device=dev)
    offset = torch.empty(n_y + 1, dtype=torch.int64, device=dev)
    common = (_ptr(x), _ptr(y), _ptr(ptr_x), _ptr(ptr_y), num_examples, n_y, dim, r2, int(max_num_neighbors),
              int(bool(drop_self)), _ptr(mx), _ptr(my))
    _lib.check(lib.qcn_radius_count(*common, _ptr(count), st), 'qcn_radius_count')
    ws_bytes = lib.qcn_scan_workspace_bytes(n_y)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.qcn_exclusive_scan_i32(_ptr(count), n_y, _ptr(offset), _ptr(ws), ws_bytes, st),
               'qcn_exclusive_scan_i32')
    num_edges = int(offset[-1].item())
    edge_index = torch.empty(2, num_edges, dtype=torch.int64, device=dev)
    if num_edges > 0:
        row, col = (edge_index[0], edge_index[1]) if y_first else (edge_index[1], edge_index[0])
        _lib.check(lib.qcn_radius_fill(*common, _ptr(offset), _ptr(row), _ptr(col), st), 'qcn_radius_fill')
    return edge_index, offset


# ------------------------------------------------------------------------------------------------
# CSR
# ------------------------------------------------------------------------------------------------
def csr_build(edge_index: torch.Tensor, n_dst: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Stable destination sort of a COO edge list → (rowptr int32 [n_dst+1], col int32 [E], eid int32 [E])."""
    return _op('csr_build')(edge_index, int(n_dst))


def _csr_build_fake(edge_index, n_dst):
    e = edge_index.size(1)
    mk = lambda n: edge_index.new_empty((n,), dtype=torch.int32)
    return mk(n_dst + 1), mk(e), mk(e)


@_register('csr_build', '(Tensor edge_index, int n_dst) -> (Tensor, Tensor, Tensor)', _csr_build_fake)
def _csr_build(edge_index, n_dst):
    dev = edge_index.device
    _require(edge_index, 'edge_index', torch.int64, dev, (2, None))
    num_edges = edge_index.size(1)
    rowptr = torch.empty(n_dst + 1, dtype=torch.int32, device=dev)
    col = torch.empty(num_edges, dtype=torch.int32, device=dev)
    eid = torch.empty(num_edges, dtype=torch.int32, device=dev)
    lib = _lib.lib()
    ws_bytes = lib.qcn_csr_workspace_bytes(num_edges, n_dst)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.qcn_csr_build(_ptr(edge_index[0]), _ptr(edge_index[1]), num_edges, n_dst, _ptr(rowptr), _ptr(col),
                                 _ptr(eid), _ptr(ws), ws_bytes, _stream(dev)), 'qcn_csr_build')
    return rowptr, col, eid


def csr_static_max_edges() -> int:
    return int(_lib.lib().qcn_csr_static_max_edges())


def csr_build_static(edge_index: torch.Tensor, n_dst: int, rowptr: torch.Tensor, col: torch.Tensor,
                     eid: torch.Tensor) -> None:
    """qcn_csr_build_static: csr_build into CALLER-provided outputs with the sort workspace inside the library — the call
    allocates nothing (for code under HIP graph capture / serving loops).  At most csr_static_max_edges() edges; calls
    that overlap on one GPU race (they share the library's per-GPU scratch array)."""
    _op('csr_build_static')(edge_index, int(n_dst), rowptr, col, eid)


@_register('csr_build_static', '(Tensor edge_index, int n_dst, Tensor(a!) rowptr, Tensor(b!) col, Tensor(c!) eid) -> ()',
           lambda edge_index, n_dst, rowptr, col, eid: None)
def _csr_build_static(edge_index, n_dst, rowptr, col, eid):
    dev = edge_index.device
    _require(edge_index, 'edge_index', torch.int64, dev, (2, None))
    num_edges = edge_index.size(1)
    _require(rowptr, 'rowptr', torch.int32, dev, (n_dst + 1,))
    _require(col, 'col', torch.int32, dev, (num_edges,))
    _require(eid, 'eid', torch.int32, dev, (num_edges,))
    if torch.cuda.current_device() != dev.index:
        raise ValueError('csr_build_static uses the scratch array of the CURRENT device; make %s current' % dev)
    src = ctypes.c_void_p(edge_index.data_ptr())
    dst = ctypes.c_void_p(edge_index.data_ptr() + 8 * num_edges)
    _lib.check(_lib.lib().qcn_csr_build_static(src, dst, num_edges, n_dst, _ptr(rowptr), _ptr(col), _ptr(eid),
                                               _stream(dev)), 'qcn_csr_build_static')


# ------------------------------------------------------------------------------------------------
# attention propagate
# ------------------------------------------------------------------------------------------------
def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, kr: Optional[torch.Tensor],
             vr: Optional[torch.Tensor], rowptr: torch.Tensor, col: torch.Tensor,
             eid: Optional[torch.Tensor]) -> torch.Tensor:
    """qcn_attn_fwd_f32: gather + segment softmax + weighted scatter-add, [n_dst, 128] fp32."""
    return _op('attn_fwd')(q, k, v, kr, vr, rowptr, col, eid)


@_register('attn_fwd', '(Tensor q, Tensor k, Tensor v, Tensor? kr, Tensor? vr, Tensor rowptr, Tensor col, Tensor? eid) -> Tensor',
           lambda q, k, v, kr, vr, rowptr, col, eid: torch.empty_like(q))
def _attn_fwd(q, k, v, kr, vr, rowptr, col, eid):
    dev = q.device
    n_dst, n_src, num_edges = q.size(0), k.size(0), col.numel()
    _require(q, 'q', torch.float32, dev, (n_dst, C))
    _require(k, 'k', torch.float32, dev, (n_src, C))
    _require(v, 'v', torch.float32, dev, (n_src, C))
    _require(rowptr, 'rowptr', torch.int32, dev, (n_dst + 1,))
    _require(col, 'col', torch.int32, dev, (num_edges,))
    if (kr is None) != (vr is None):
        raise ValueError('kr and vr must be given together')
    if kr is not None:
        _require(kr, 'kr', torch.float32, dev, (num_edges, C))
        _require(vr, 'vr', torch.float32, dev, (num_edges, C))
    if eid is not None:
        _require(eid, 'eid', torch.int32, dev, (num_edges,))
    out = torch.empty(n_dst, C, dtype=torch.float32, device=dev)
    tok = _prof_begin(PROF_ATTN, (2064.0 if kr is not None else 1040.0) * num_edges + 1024.0 * n_dst, dev)
    _lib.check(_lib.lib().qcn_attn_fwd_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(kr), _ptr(vr), _ptr(rowptr), _ptr(col),
                                           _ptr(eid), n_dst, n_src, num_edges, _ptr(out), _stream(dev)),
               'qcn_attn_fwd_f32')
    _prof_end(tok, dev)
    return out


def row_normalize(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """qcn_row_normalize_f32: (x - mean) / sqrt(var + eps) per 128-wide row (LayerNorm without affine)."""
    return _op('row_normalize')(x, float(eps))


@_register('row_normalize', '(Tensor x, float eps) -> Tensor', lambda x, eps: torch.empty_like(x))
def _row_normalize(x, eps):
    dev = x.device
    _require(x, 'x', torch.float32, dev, (None, C))
    y = torch.empty_like(x)
    _lib.check(_lib.lib().qcn_row_normalize_f32(_ptr(x), _ptr(y), x.size(0), float(eps), _stream(dev)),
               'qcn_row_normalize_f32')
    return y


_CHAIN_INDEX = {}


def chain_feature_order(device=None) -> torch.Tensor:
    """[64, 2] long: the input feature that k-step ``s`` of the feature-major MFMA chain takes from lane half ``h``
    (csrc/chain_tile.hip.h: element (t, r) of a row vector is feature 32 t + (r & 3) + 8 (r >> 2) + 4 h, s = 16 t + r)."""
    s = torch.arange(64).unsqueeze(1)
    h = torch.arange(2).unsqueeze(0)
    r = s & 15
    return (32 * (s >> 4) + (r & 3) + 8 * (r >> 2) + 4 * h).to(device)


MATMUL_EXACT_F32, MATMUL_SPLIT_F16, MATMUL_BF16 = 0, 1, 2
MATMUL_MODE_NAMES = {MATMUL_EXACT_F32: 'exact-f32', MATMUL_SPLIT_F16: 'split-f16x3', MATMUL_BF16: 'bf16'}
CHAIN_MATRIX_FLOATS = C * C

# Process default of the Python layer (QCN_MATMUL_MODE=exact|split|bf16; QCN_MATMUL_EXACT_F32=1 is the old spelling).  The
# C library itself is stateless: every chain entry point takes the mode as an argument.
_DEFAULT_MODE = {'exact': MATMUL_EXACT_F32, 'split': MATMUL_SPLIT_F16, 'bf16': MATMUL_BF16}[
    os.environ.get('QCN_MATMUL_MODE', 'exact' if os.environ.get('QCN_MATMUL_EXACT_F32', '0') == '1' else 'split')]


def matmul_mode() -> int:
    """Default arithmetic mode of the MFMA chain kernels (include/qcnet_hip.h: QCN_MATMUL_*) for calls that name none."""
    return _DEFAULT_MODE


def set_matmul_mode(mode: int) -> None:
    global _DEFAULT_MODE
    if int(mode) not in MATMUL_MODE_NAMES:
        raise ValueError('unknown matmul mode %r' % (mode,))
    _DEFAULT_MODE = int(mode)


def _mode(mode: Optional[int]) -> int:
    mode = _DEFAULT_MODE if mode is None else int(mode)
    if mode not in MATMUL_MODE_NAMES:
        raise ValueError('unknown matmul mode %r' % (mode,))
    return mode


def split_matrix_scale(weight: torch.Tensor) -> float:
    """The power of two that puts max|W| into [2^14, 2^15) (1.0 for a zero matrix): the split mode's matrix scale."""
    m = float(weight.detach().abs().max())
    if not m > 0:
        return 1.0
    _, e = torch.frexp(torch.tensor(m))                 # m = f * 2^e, f in [1/2, 1)
    return 2.0 ** (15 - int(e))


def static_operand_scale(bound: float) -> float:
    """Power-of-two operand scale for a row vector with |x| <= ``bound`` (csrc/chain_tile.hip.h make_operand_static):
    bound * scale lands in (2^14, 2^15], a factor two below the f16 overflow threshold."""
    if not bound > 0:
        return 1.0
    _, e = torch.frexp(torch.tensor(float(bound)))      # bound <= 2^e
    return 2.0 ** (15 - int(e))


LN_MAX_ABS = 127.0 ** 0.5   # |(x - mean) / sqrt(var + eps)| <= sqrt(n - 1) for n = 128 features


def layer_norm_bound(ln_weight: torch.Tensor, ln_bias: torch.Tensor) -> float:
    """Rigorous bound on |LayerNorm(x)| over all inputs: sqrt(127) max|gamma| + max|beta| (1 % slack for rounding)."""
    return 1.01 * (LN_MAX_ABS * float(ln_weight.detach().abs().max()) + float(ln_bias.detach().abs().max()))


def pack_chain_matrix(weight: torch.Tensor, in_index: Optional[torch.Tensor] = None,
                      mode: Optional[int] = None, scale: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """A ``nn.Linear(128, 128).weight`` ([out, in]) in the order the chain kernels consume it, for the matmul mode in
    force: (16384 floats, {scale, 1 / scale}) — layouts in include/qcnet_hip.h, 'Chain-packed 128x128 matrix'.
    ``in_index`` [64, 2]: the input column that register position s of lane half h stands for; defaults to
    :func:`chain_feature_order` (the operand is a register-resident row vector).  ``scale`` (split mode): a power of two
    not above :func:`split_matrix_scale` to use instead of it (sums of GEMMs that share one accumulator scale)."""
    if tuple(weight.shape) != (C, C):
        raise ValueError('pack_chain_matrix expects a [128, 128] weight, got %s' % (tuple(weight.shape),))
    mode = _mode(mode)
    dev = weight.device
    if in_index is None:
        key = str(dev)
        if key not in _CHAIN_INDEX:
            _CHAIN_INDEX[key] = chain_feature_order(dev)
        in_index = _CHAIN_INDEX[key]
    in_index = in_index.to(dev)
    w = weight.detach().float()
    if not bool(torch.isfinite(w).all()):
        raise ValueError('pack_chain_matrix: weight has non-finite entries')
    out_index = (32 * torch.arange(4, device=dev).view(1, 4) + torch.arange(32, device=dev).view(32, 1))  # [m, t]
    if mode == MATMUL_EXACT_F32:
        packed = w[out_index.view(1, 1, 32, 4), in_index.view(64, 2, 1, 1)]          # P[s, h, m, t]
        return packed.reshape(-1).contiguous(), torch.ones(2, dtype=torch.float32, device=dev)
    if mode == MATMUL_BF16:
        kin = in_index.view(8, 8, 2).permute(0, 2, 1)                                 # [j, h, i] = in_index[8 j + i, h]
        rows = out_index.t().reshape(1, 4, 1, 32, 1)                                  # [., t, ., m, .]
        cols = kin.reshape(8, 1, 2, 1, 8)                                             # [j, ., h, ., i]
        frag = w.bfloat16()[rows, cols].contiguous().view(-1).view(torch.float32)     # [j, t, lane = 32 h + m, i], RNE
        packed = torch.cat([frag, frag.new_zeros(CHAIN_MATRIX_FLOATS - frag.numel())])
        return packed.contiguous(), torch.ones(2, dtype=torch.float32, device=dev)
    if mode != MATMUL_SPLIT_F16:
        raise ValueError('unknown matmul mode %r' % (mode,))
    auto_scale = split_matrix_scale(w)
    if scale is None:
        scale = auto_scale
    elif not 0 < scale <= auto_scale:
        raise ValueError('pack_chain_matrix: scale %r would overflow the f16 planes (at most %r)' % (scale, auto_scale))
    ws = w * scale                                      # exact (power of two); max|ws| < 2^15
    hi = ws.half()
    lo = (ws - hi.float()).half()                       # ws - hi is exact in fp32
    kin = in_index.view(8, 8, 2).permute(0, 2, 1)       # [j, h, i] = in_index[8 j + i, h]
    rows = out_index.t().reshape(1, 4, 1, 1, 32, 1)     # [., t, ., ., m, .]
    cols = kin.reshape(8, 1, 1, 2, 1, 8)                # [j, ., ., h, ., i]
    planes = torch.stack([hi, lo])                      # [p, out, in]
    pidx = torch.arange(2, device=dev).view(1, 1, 2, 1, 1, 1)
    packed = planes[pidx, rows, cols]                   # [j, t, p, h, m, i] f16
    packed = packed.contiguous().view(-1).view(torch.float32)
    assert packed.numel() == CHAIN_MATRIX_FLOATS
    return packed.contiguous(), torch.tensor([scale, 1.0 / scale], dtype=torch.float32, device=dev)


def pack_chain_set(weights, in_indices=None, mode: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """([n, 16384] chain-packed matrices, [n, 2] scale pairs) for a list of weights."""
    in_indices = in_indices or [None] * len(weights)
    pairs = [pack_chain_matrix(w, ix, mode) for w, ix in zip(weights, in_indices)]
    return torch.stack([p[0] for p in pairs]).contiguous(), torch.stack([p[1] for p in pairs]).contiguous()


def pack_chain_matrices(weights, mode: Optional[int] = None) -> torch.Tensor:
    """The ``wt`` argument of qcn_ln_linear_f32: n chain-packed matrices followed by their n scale pairs, flat."""
    mats, scales = pack_chain_set(weights, mode=mode)
    return torch.cat([mats.reshape(-1), scales.reshape(-1)]).contiguous()


def scale_table(scales: torch.Tensor, size: int = 32) -> torch.Tensor:
    """Scale pairs padded to the fixed-size table at the end of the Fourier / attn_post packs."""
    flat = scales.reshape(-1)
    return torch.cat([flat, flat.new_ones(size - flat.numel())])


def pack_fourier(freqs: Optional[torch.Tensor], mlps: Sequence[Sequence[torch.Tensor]], to_out: Sequence[torch.Tensor],
                 mode: Optional[int] = None) -> torch.Tensor:
    """The ``pack`` of qcn_fourier_embed_f32 (include/qcnet_hip.h) for one FourierEmbedding (layers/fourier_embedding.py:25-46).
    ``freqs`` [input_dim, 64] (None for input_dim 0); ``mlps[i]`` = (mlps.i.0.weight [128, 129], mlps.i.0.bias, mlps.i.1.weight,
    mlps.i.1.bias, mlps.i.3.weight, mlps.i.3.bias); ``to_out`` = (to_out.0.weight, to_out.0.bias, to_out.2.weight, to_out.2.bias).

    Two exact rewrites happen here instead of in the kernel's vector ALU:
    * the first Linear of every per-dimension MLP is CENTRED over its output features (rows of (I - 11^T / 128) W, centred
      bias): LayerNorm's mean of W f + b is linear in f, so the kernel's LayerNorm starts at the variance (formed in fp64);
    * every LayerNorm affine carries the static power-of-two operand scale of the GEMM that consumes it (split mode), and
      the second Linear layers share ONE accumulator scale, so their sum over input dims needs no work between the GEMMs."""
    mode = _mode(mode)
    split = mode == MATMUL_SPLIT_F16
    input_dim = len(mlps)
    dev = to_out[2].device
    f32 = lambda t: t.detach().to(dev, torch.float32)
    s_idx = torch.arange(64).unsqueeze(1)
    h_idx = torch.arange(2).unsqueeze(0)
    # k-step s of a first layer takes, from lane half h, cos (s < 32) or sin (s >= 32) of frequency 32 h + (s mod 32):
    # input column 32 h + s resp. 64 + 32 h + (s - 32) of the reference's cat order (layers/fourier_embedding.py:59)
    first_layer_order = torch.where(s_idx < 32, 32 * h_idx + s_idx, 64 + 32 * h_idx + (s_idx - 32))
    # (not in the bf16 mode: its operands are the bf16 roundings of the reference's own weights, the kernels keep the mean pass)
    centre = ((lambda t: t) if mode == MATMUL_BF16 else
              (lambda t: (t.double() - t.double().mean(dim=0, keepdim=True)).float()))   # over the output features

    s1 = [static_operand_scale(layer_norm_bound(m[2], m[3])) if split else 1.0 for m in mlps]
    so = static_operand_scale(layer_norm_bound(to_out[0], to_out[1])) if split else 1.0
    # common accumulator scale of the second layers: operand scale x matrix scale, the smallest over the input dims
    w2_scale = [split_matrix_scale(centre(f32(m[4]))) if split else 1.0 for m in mlps]
    acc2 = min([a * b for a, b in zip(s1, w2_scale)], default=1.0)

    mats, table, vecs = [], torch.ones(32, dtype=torch.float32, device=dev), []
    for i, (w1, b1, g, be, w2, b2) in enumerate(mlps):
        w1c = centre(f32(w1))
        p1, sc1 = pack_chain_matrix(w1c[:, :C], first_layer_order, mode)
        p2, _ = pack_chain_matrix(centre(f32(w2)), None, mode, scale=acc2 / s1[i] if split else None)
        mats += [p1, p2]
        table[2 * i] = float(sc1[1])                       # sines / cosines: operand scale 1
        fr = f32(freqs[i])
        fmx = torch.stack([fr[:32].abs().max(), fr[32:].abs().max(), fr.new_zeros(()), fr.new_zeros(())])
        vecs += [fr, w1c[:, C].contiguous(), centre(f32(b1)), f32(g) * s1[i], f32(be) * s1[i], fmx]
    g, be, w, b = to_out
    p, sc = pack_chain_matrix(f32(w), None, mode)
    mats.append(p)
    table[2 * input_dim] = float(sc[1]) / so
    table[24], table[25], table[26] = so, 1.0 / so, 1.0 / acc2
    b2_sum = torch.stack([f32(m[5]).double() for m in mlps]).sum(dim=0).float() if input_dim else torch.zeros(C, device=dev)
    b2_sum = centre(b2_sum)     # second layers centred too: without categorical embeddings to_out's LayerNorm sees zero-mean rows
    vecs += [b2_sum, f32(g) * so, f32(be) * so, f32(b)]
    return torch.cat([m.reshape(-1) for m in mats] + [v.reshape(-1) for v in vecs] + [table]).contiguous()


def fourier_embed(x: Optional[torch.Tensor], input_dim: int, pack: torch.Tensor, extra: Optional[torch.Tensor] = None,
                  extra_div: int = 1, normalize_out=False, n_rows: Optional[int] = None,
                  mode: Optional[int] = None) -> torch.Tensor:
    """qcn_fourier_embed_f32: the whole FourierEmbedding forward; ``pack`` from ``FourierEmbedding.packed_weights`` for the
    same ``mode``.  ``normalize_out``: False / True (row-normalised) / 'split' (row-normalised split-f16 rows, the operand
    of attn_modes) / 'bf16' (row-normalised bf16 rows [n, 128], the operand of attn_modes in the bf16 mode)."""
    if input_dim == 0 and extra is None:
        raise ValueError('Both continuous_inputs and categorical_embs are None')
    if input_dim > 0:
        n_rows = x.size(0)
    elif n_rows is None:
        n_rows = extra.size(0) * extra_div
    norm = {'split': 2, 'bf16': 3}.get(normalize_out, int(bool(normalize_out)))
    return _op('fourier_embed')(x, int(input_dim), pack, extra, int(extra_div), norm, int(n_rows), _mode(mode))


def _fourier_flops(n_rows: int, input_dim: int, tail_mats: int) -> float:
    return float(n_rows) * (input_dim * 2.0 * C * (129.0 + C) + 2.0 * C * C * tail_mats)


def _fourier_embed_fake(x, input_dim, pack, extra, extra_div, normalize_out, n_rows, mode):
    if normalize_out == 2:
        return pack.new_empty((n_rows, 2 * C), dtype=torch.float16)
    if normalize_out == 3:
        return pack.new_empty((n_rows, C), dtype=torch.bfloat16)
    return pack.new_empty((n_rows, C))


@_register('fourier_embed', '(Tensor? x, int input_dim, Tensor pack, Tensor? extra, int extra_div, int normalize_out, int n_rows, '
           'int mode) -> Tensor', _fourier_embed_fake)
def _fourier_embed(x, input_dim, pack, extra, extra_div, normalize_out, n_rows, mode):
    dev = pack.device
    if input_dim > 0:
        _require(x, 'continuous_inputs', torch.float32, dev, (n_rows, input_dim))
    need = _lib.lib().qcn_fourier_pack_floats(int(input_dim))
    _require(pack, 'pack', torch.float32, dev, (need,))
    if extra is not None:
        _require(extra, 'categorical embedding sum', torch.float32, dev, ((n_rows + extra_div - 1) // extra_div, C))
    elif input_dim == 0:
        raise ValueError('Both continuous_inputs and categorical_embs are None')
    if normalize_out == 2:         # row-normalised AND in the split-f16 operand format of attn_modes
        y = torch.empty(n_rows, 2 * C, dtype=torch.float16, device=dev)
    elif normalize_out == 3:       # ... in its bf16 operand format
        y = torch.empty(n_rows, C, dtype=torch.bfloat16, device=dev)
    else:
        y = torch.empty(n_rows, C, dtype=torch.float32, device=dev)
    tok = _prof_begin(PROF_FOURIER, _fourier_flops(n_rows, input_dim, 1), dev)
    _lib.check(_lib.lib().qcn_fourier_embed_f32(_ptr(x) if input_dim > 0 else None, int(input_dim), _ptr(extra),
                                                int(extra_div), _ptr(pack), n_rows, int(normalize_out), _ptr(y), int(mode),
                                                _stream(dev)), 'qcn_fourier_embed_f32')
    _prof_end(tok, dev)
    return y


def fourier_linear(x: torch.Tensor, input_dim: int, pack: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor],
                   folded: bool = False, mode: Optional[int] = None) -> List[torch.Tensor]:
    """qcn_fourier_linear_f32: [W_j FourierEmbedding(x) + b_j] for the n_out <= 3 layers of ``wt`` (pack_chain_matrices).
    ``folded``: ``wt`` / ``bias`` already contain the embedding's last Linear (qcn_fourier_folded_linear_f32,
    :func:`fold_linear`)."""
    return list(_op('fourier_linear')(x, int(input_dim), pack, wt, bias, bool(folded), _mode(mode)))


def _fourier_linear_fake(x, input_dim, pack, wt, bias, folded, mode):
    return [x.new_empty((x.size(0), C)) for _ in range(wt.numel() // (C * C + 2))]


@_register('fourier_linear', '(Tensor x, int input_dim, Tensor pack, Tensor wt, Tensor? bias, bool folded, int mode) -> Tensor[]',
           _fourier_linear_fake)
def _fourier_linear(x, input_dim, pack, wt, bias, folded, mode):
    dev = pack.device
    _require(x, 'continuous_inputs', torch.float32, dev, (None, input_dim))
    n = x.size(0)
    _require(pack, 'pack', torch.float32, dev, (int(_lib.lib().qcn_fourier_pack_floats(int(input_dim))),))
    n_out = wt.numel() // (C * C + 2)
    if not 1 <= n_out <= 3:
        raise ValueError('n_out must be 1..3')
    _require(wt, 'wt', torch.float32, dev, (n_out * (C * C + 2),))
    if bias is not None:
        _require(bias, 'bias', torch.float32, dev, (n_out, C))
    ys = [torch.empty(n, C, dtype=torch.float32, device=dev) for _ in range(n_out)]
    yp = [_ptr(y) for y in ys] + [None] * (3 - n_out)
    fn = _lib.lib().qcn_fourier_folded_linear_f32 if folded else _lib.lib().qcn_fourier_linear_f32
    tok = _prof_begin(PROF_FOURIER, _fourier_flops(n, input_dim, n_out + (0 if folded else 1)), dev)
    _lib.check(fn(_ptr(x), int(input_dim), None, 1, _ptr(pack), n, _ptr(wt), _ptr(bias), n_out,
                  yp[0], yp[1], yp[2], int(mode), _stream(dev)),
               'qcn_fourier_folded_linear_f32' if folded else 'qcn_fourier_linear_f32')
    _prof_end(tok, dev)
    return ys


def trajectory_gru(x: torch.Tensor, input_dim: int, n_steps: int, pack: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor,
                   folded: bool, whh_pack: torch.Tensor, b_hh: torch.Tensor, h0: torch.Tensor, mode: Optional[int] = None
                   ) -> torch.Tensor:
    """The decoder's trajectory re-encoding (modules/qcnet_decoder.py:255-257) as two launches that hand the GRU's input
    halves over in the recurrence's own blocked layout (qcn_fourier_gru_input_f32 + qcn_gru_blocked_f32): same values as
    ``fourier_linear`` + ``gru_recurrence``, bit for bit.  ``x`` [n_seq * n_steps, input_dim], row = seq * n_steps + t;
    split-f16 / bf16 modes only (the exact mode's recurrence kernel has another thread mapping)."""
    return _op('trajectory_gru')(x, int(input_dim), int(n_steps), pack, wt, bias, bool(folded), whh_pack, b_hh, h0, _mode(mode))


@_register('trajectory_gru', '(Tensor x, int input_dim, int n_steps, Tensor pack, Tensor wt, Tensor bias, bool folded, Tensor whh_pack, '
           'Tensor b_hh, Tensor h0, int mode) -> Tensor',
           lambda x, input_dim, n_steps, pack, wt, bias, folded, whh_pack, b_hh, h0, mode: x.new_empty((x.size(0) // n_steps, C)))
def _trajectory_gru(x, input_dim, n_steps, pack, wt, bias, folded, whh_pack, b_hh, h0, mode):
    dev = pack.device
    lib = _lib.lib()
    if mode == MATMUL_EXACT_F32:
        raise ValueError('trajectory_gru: split-f16 / bf16 modes only')
    _require(x, 'x', torch.float32, dev, (None, input_dim))
    if n_steps < 1 or x.size(0) % n_steps:
        raise ValueError('x rows (%d) not a multiple of n_steps (%d)' % (x.size(0), n_steps))
    n_seq = x.size(0) // n_steps
    _require(pack, 'pack', torch.float32, dev, (int(lib.qcn_fourier_pack_floats(int(input_dim))),))
    _require(wt, 'wt', torch.float32, dev, (3 * (C * C + 2),))
    _require(bias, 'bias', torch.float32, dev, (3, C))
    _require(whh_pack, 'whh_pack', torch.float32, dev, (int(lib.qcn_gru_pack_floats()),))
    _require(b_hh, 'b_hh', torch.float32, dev, (3 * C,))
    _require(h0, 'h0', torch.float32, dev, (C,))
    h = torch.empty(n_seq, C, dtype=torch.float32, device=dev)
    if n_seq == 0:
        return h
    rows = int(lib.qcn_gru_blocked_rows(n_seq, int(n_steps)))
    gi = torch.empty(3, rows, C, dtype=torch.float32, device=dev)
    tok = _prof_begin(PROF_FOURIER, _fourier_flops(x.size(0), input_dim, 3 + (0 if folded else 1)), dev)
    _lib.check(lib.qcn_fourier_gru_input_f32(_ptr(x), int(input_dim), _ptr(pack), n_seq, int(n_steps), _ptr(wt), _ptr(bias),
                                             int(bool(folded)), _ptr(gi[0]), _ptr(gi[1]), _ptr(gi[2]), int(mode), _stream(dev)),
               'qcn_fourier_gru_input_f32')
    _prof_end(tok, dev)
    _lib.check(lib.qcn_gru_blocked_f32(_ptr(gi[0]), _ptr(gi[1]), _ptr(gi[2]), _ptr(whh_pack), _ptr(b_hh), _ptr(h0), n_seq,
                                       int(n_steps), _ptr(h), _stream(dev)), 'qcn_gru_blocked_f32')
    return h


def fold_linear(weights, biases, w_inner: torch.Tensor, b_inner: torch.Tensor):
    """(W_j W_inner, W_j b_inner + b_j) for a stack of Linear layers applied to the output of Linear(w_inner, b_inner)
    with nothing in between; products in fp64, results fp32."""
    wi, bi = w_inner.detach().double(), b_inner.detach().double()
    ws = [(w.detach().double() @ wi).float() for w in weights]
    bs = [(w.detach().double() @ bi + b.detach().double()).float() for w, b in zip(weights, biases)]
    return ws, torch.stack(bs).contiguous()


def ln_linear(x: torch.Tensor, ln_weight: Optional[torch.Tensor], ln_bias: Optional[torch.Tensor], wt: torch.Tensor,
              bias: Optional[torch.Tensor], want_xn: bool = False, mode: Optional[int] = None):
    """qcn_ln_linear_f32: ([y_0 .. y_{n_out-1}], xn or None) with y_j = W_j LayerNorm(x) + b_j; ``wt`` is the stack
    of chain-packed weights and their scale pairs, n_out * (16384 + 2) floats (:func:`pack_chain_matrices`, made for
    the same ``mode``), ``bias`` [n_out, 128] or None."""
    outs = list(_op('ln_linear')(x, ln_weight, ln_bias, wt, bias, bool(want_xn), _mode(mode)))
    return (outs[:-1], outs[-1]) if want_xn else (outs, None)


def _ln_linear_fake(x, ln_weight, ln_bias, wt, bias, want_xn, mode):
    return [torch.empty_like(x) for _ in range(wt.numel() // (C * C + 2) + int(want_xn))]


@_register('ln_linear', '(Tensor x, Tensor? ln_weight, Tensor? ln_bias, Tensor wt, Tensor? bias, bool want_xn, int mode) -> Tensor[]',
           _ln_linear_fake)
def _ln_linear(x, ln_weight, ln_bias, wt, bias, want_xn, mode):
    dev = x.device
    n = x.size(0)
    n_out = wt.numel() // (C * C + 2)
    _require(x, 'x', torch.float32, dev, (n, C))
    _require(wt, 'wt', torch.float32, dev, (n_out * (C * C + 2),))
    if not 1 <= n_out <= 4:
        raise ValueError('n_out must be 1..4')
    if bias is not None:
        _require(bias, 'bias', torch.float32, dev, (n_out, C))
    if ln_weight is not None:
        _require(ln_weight, 'ln_weight', torch.float32, dev, (C,))
        _require(ln_bias, 'ln_bias', torch.float32, dev, (C,))
    ys = [torch.empty(n, C, dtype=torch.float32, device=dev) for _ in range(n_out)]
    xn = torch.empty(n, C, dtype=torch.float32, device=dev) if want_xn else None
    yp = [_ptr(y) for y in ys] + [None] * (4 - n_out)
    tok = _prof_begin(PROF_GEMM, float(n) * n_out * 2.0 * C * C, dev)
    _lib.check(_lib.lib().qcn_ln_linear_f32(_ptr(x), n, _ptr(ln_weight), _ptr(ln_bias), _ptr(wt), _ptr(bias), n_out,
                                            yp[0], yp[1], yp[2], yp[3], _ptr(xn), int(mode), _stream(dev)),
               'qcn_ln_linear_f32')
    _prof_end(tok, dev)
    return ys + ([xn] if want_xn else [])


def attn_post(agg: torch.Tensor, xn_dst: Optional[torch.Tensor], s_proj: torch.Tensor, x_res: torch.Tensor,
              pack: torch.Tensor, mode: Optional[int] = None) -> torch.Tensor:
    """qcn_attn_post_f32: gate/update + to_out + post-norm residual + feed-forward block of one AttentionLayer.
    ``xn_dst`` None: the kernel recomputes attn_prenorm_x_dst(x_res) from the affine stored in ``pack``."""
    return _op('attn_post')(agg, xn_dst, s_proj, x_res, pack, _mode(mode))


@_register('attn_post', '(Tensor agg, Tensor? xn_dst, Tensor s_proj, Tensor x_res, Tensor pack, int mode) -> Tensor',
           lambda agg, xn_dst, s_proj, x_res, pack, mode: torch.empty_like(agg))
def _attn_post(agg, xn_dst, s_proj, x_res, pack, mode):
    dev = agg.device
    n = agg.size(0)
    for name, t in (('agg', agg), ('xn_dst', xn_dst), ('s_proj', s_proj), ('x_res', x_res)):
        if t is None and name == 'xn_dst':
            continue
        _require(t, name, torch.float32, dev, (n, C))
    _require(pack, 'pack', torch.float32, dev, (_lib.lib().qcn_attn_post_pack_floats(),))
    out = torch.empty(n, C, dtype=torch.float32, device=dev)
    tok = _prof_begin(PROF_GEMM, float(n) * 11 * 2.0 * C * C, dev)
    _lib.check(_lib.lib().qcn_attn_post_f32(_ptr(agg), _ptr(xn_dst), _ptr(s_proj), _ptr(x_res), _ptr(pack), n, _ptr(out),
                                            int(mode), _stream(dev)), 'qcn_attn_post_f32')
    _prof_end(tok, dev)
    return out


def pack_attn_post_chain(to_g_w: torch.Tensor, to_g_b: torch.Tensor, to_s_w: torch.Tensor, to_s_b: torch.Tensor,
                         to_out_w: torch.Tensor, to_out_b: torch.Tensor, ff1_w: torch.Tensor, ff1_b: torch.Tensor,
                         ff2_w: torch.Tensor, ff2_b: torch.Tensor, postnorm, ff_prenorm, ff_postnorm, prenorm_dst,
                         mode: Optional[int] = None) -> torch.Tensor:
    """The ``pack`` of qcn_attn_post_chain_f32 (include/qcnet_hip.h) from the parameters of one AttentionLayer
    (layers/attention_layer.py:44-61): ``to_g`` [128, 256], ``to_s`` / ``to_out`` [128, 128], ``ff_mlp.0`` [512, 128], ``ff_mlp.3``
    [128, 512] and the four LayerNorms as (weight, bias) pairs.
    Split mode: the LayerNorm outputs and the hidden blocks are STATIC-scale operands of the kernel — |LN(x)| <= sqrt(127)
    max|gamma| + max|beta| and |relu(W1 y + b1)| <= max_j (sum_k |W1_jk| bound(y) + |b1_j|) hold for every input, so the powers of
    two that put those bounds at 2^15 are folded into gamma / beta / b1 and the scale table here (no row maximum, no scaling
    multiply in the kernel), and the four second-layer matrices get matrix scales that make (hidden scale x matrix scale) ONE
    value P: their products accumulate raw, the way back is one fma with the bias."""
    mode = _mode(mode)
    f = lambda t: t.detach().float()
    wg, w1, w2, b1 = f(to_g_w), f(ff1_w), f(ff2_w), f(ff1_b)
    heads = [wg[:, :C], wg[:, C:], f(to_s_w), f(to_out_w)]
    ffn = []
    for hb in range(4):
        ffn += [w1[hb * C:(hb + 1) * C, :], w2[:, hb * C:(hb + 1) * C]]
    pre_g, pre_b = f(prenorm_dst[0]), f(prenorm_dst[1])
    lf_g, lf_b = f(ff_prenorm[0]), f(ff_prenorm[1])
    extra = torch.ones(8, dtype=torch.float32, device=wg.device)
    if mode == MATMUL_SPLIT_F16:
        s_pre = static_operand_scale(layer_norm_bound(pre_g, pre_b))
        bound_ff = layer_norm_bound(lf_g, lf_b)
        s_ff = static_operand_scale(bound_ff)
        s_h = []
        for hb in range(4):
            rows = w1[hb * C:(hb + 1) * C, :]
            bound = 1.01 * float((rows.abs().sum(dim=1) * bound_ff + b1[hb * C:(hb + 1) * C].abs()).max())
            s_h.append(static_operand_scale(bound))
        p_common = min(s_h[hb] * split_matrix_scale(ffn[2 * hb + 1]) for hb in range(4))
        pairs = [pack_chain_matrix(w, mode=mode) for w in heads]
        for hb in range(4):
            m1, sc1 = pack_chain_matrix(ffn[2 * hb], mode=mode)
            sc1 = sc1.clone()
            sc1[1] = sc1[1] * s_h[hb]            # the way back from the first layer lands in the hidden block's operand scale
            pairs += [(m1, sc1), pack_chain_matrix(ffn[2 * hb + 1], mode=mode, scale=p_common / s_h[hb])]
        pre_g, pre_b, lf_g, lf_b = pre_g * s_pre, pre_b * s_pre, lf_g * s_ff, lf_b * s_ff
        b1 = torch.cat([b1[hb * C:(hb + 1) * C] * s_h[hb] for hb in range(4)])
        extra = torch.tensor([s_pre, 1.0 / s_pre, s_ff, 1.0 / s_ff, 1.0 / p_common, 1.0, 1.0, 1.0], dtype=torch.float32,
                             device=wg.device)
    else:
        pairs = [pack_chain_matrix(w, mode=mode) for w in heads + ffn]
    mats = torch.stack([p[0] for p in pairs])
    scales = torch.stack([p[1] for p in pairs]).reshape(-1)                    # 24 floats
    vecs = [to_g_b, to_out_b, postnorm[0], postnorm[1], lf_g, lf_b, b1, ff2_b, ff_postnorm[0], ff_postnorm[1], pre_g, pre_b,
            to_s_b]
    out = torch.cat([mats.reshape(-1)] + [f(v).reshape(-1) for v in vecs] + [scales, extra]).contiguous()
    assert out.numel() == int(_lib.lib().qcn_attn_post_chain_pack_floats())
    return out


def attn_post_chain(agg: torch.Tensor, x_res: torch.Tensor, pack: torch.Tensor, next_wt: Optional[torch.Tensor] = None,
                    next_bias: Optional[torch.Tensor] = None, next_ln_weight: Optional[torch.Tensor] = None,
                    next_ln_bias: Optional[torch.Tensor] = None, mode: Optional[int] = None,
                    next_operand_scale: float = 0.0) -> List[torch.Tensor]:
    """qcn_attn_post_chain_f32: ``[out, y_0 .. y_{n_next-1}]`` — the node-side block of one AttentionLayer with to_s computed
    inside (``pack``: 12 matrices, see include/qcnet_hip.h) and, with ``next_wt``, the next layer's LayerNorm + projections
    applied to the output row (``next_wt`` / ``next_bias`` as for :func:`ln_linear`, 1..3 matrices; ``next_operand_scale``:
    ``static_operand_scale(layer_norm_bound(next_ln_weight, next_ln_bias))`` or 0 for per-row maxima)."""
    return list(_op('attn_post_chain')(agg, x_res, pack, next_wt, next_bias, next_ln_weight, next_ln_bias,
                                       float(next_operand_scale), _mode(mode)))


def _attn_post_chain_fake(agg, x_res, pack, next_wt, next_bias, next_ln_weight, next_ln_bias, next_operand_scale, mode):
    n_next = 0 if next_wt is None else next_wt.numel() // (C * C + 2)
    return [torch.empty_like(agg) for _ in range(1 + n_next)]


@_register('attn_post_chain', '(Tensor agg, Tensor x_res, Tensor pack, Tensor? next_wt, Tensor? next_bias, Tensor? next_ln_weight, '
           'Tensor? next_ln_bias, float next_operand_scale, int mode) -> Tensor[]', _attn_post_chain_fake)
def _attn_post_chain(agg, x_res, pack, next_wt, next_bias, next_ln_weight, next_ln_bias, next_operand_scale, mode):
    dev = agg.device
    n = agg.size(0)
    _require(agg, 'agg', torch.float32, dev, (n, C))
    _require(x_res, 'x_res', torch.float32, dev, (n, C))
    _require(pack, 'pack', torch.float32, dev, (_lib.lib().qcn_attn_post_chain_pack_floats(),))
    n_next = 0
    if next_wt is not None:
        n_next = next_wt.numel() // (C * C + 2)
        if not 1 <= n_next <= 3:
            raise ValueError('next_wt must hold 1..3 matrices')
        _require(next_wt, 'next_wt', torch.float32, dev, (n_next * (C * C + 2),))
        if next_bias is not None:
            _require(next_bias, 'next_bias', torch.float32, dev, (n_next, C))
        if next_ln_weight is None or next_ln_bias is None:
            raise ValueError('next_wt needs next_ln_weight and next_ln_bias')
        _require(next_ln_weight, 'next_ln_weight', torch.float32, dev, (C,))
        _require(next_ln_bias, 'next_ln_bias', torch.float32, dev, (C,))
    outs = [torch.empty(n, C, dtype=torch.float32, device=dev) for _ in range(1 + n_next)]
    yp = [_ptr(y) for y in outs[1:]] + [None] * (3 - n_next)
    tok = _prof_begin(PROF_GEMM, float(n) * (12 + n_next) * 2.0 * C * C, dev)
    _lib.check(_lib.lib().qcn_attn_post_chain_f32(_ptr(agg), _ptr(x_res), _ptr(pack), n, _ptr(outs[0]), _ptr(next_wt),
                                                  _ptr(next_bias) if n_next else None, _ptr(next_ln_weight) if n_next else None,
                                                  _ptr(next_ln_bias) if n_next else None, float(next_operand_scale), n_next, yp[0],
                                                  yp[1], yp[2], int(mode), _stream(dev)), 'qcn_attn_post_chain_f32')
    _prof_end(tok, dev)
    return outs


def rel_geometry(pos_src: torch.Tensor, pos_dst: torch.Tensor, head_src: torch.Tensor, head_dst: torch.Tensor,
                 edge_index: torch.Tensor, extra: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qcn_rel_geometry_f32: [E, 3 (4 with ``extra``)] relative-geometry rows of an edge list ([2, E] int64, row 0 indexes
    the ``*_src`` tensors, row 1 the ``*_dst`` tensors; the edge lists come from the graph builders and are in range)."""
    if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
        raise ValueError('edge_index must be [2, E] int64')
    if extra is not None:
        extra = extra.to(torch.int64)
    return _op('rel_geometry')(pos_src, pos_dst, head_src, head_dst, edge_index, extra)


def _rel_geometry_fake(pos_src, pos_dst, head_src, head_dst, edge_index, extra):
    return pos_src.new_empty((edge_index.size(1), 4 if extra is not None else 3))


@_register('rel_geometry', '(Tensor pos_src, Tensor pos_dst, Tensor head_src, Tensor head_dst, Tensor edge_index, Tensor? extra) '
           '-> Tensor', _rel_geometry_fake)
def _rel_geometry(pos_src, pos_dst, head_src, head_dst, edge_index, extra):
    dev = pos_src.device
    _require(pos_src, 'pos_src', torch.float32, dev, (None, None))
    _require(pos_dst, 'pos_dst', torch.float32, dev, (None, None))
    _require(head_src, 'head_src', torch.float32, dev, (pos_src.size(0),))
    _require(head_dst, 'head_dst', torch.float32, dev, (pos_dst.size(0),))
    if pos_src.size(1) < 2 or pos_dst.size(1) < 2:
        raise ValueError('positions need x and y columns')
    if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
        raise ValueError('edge_index must be [2, E] int64')
    e = edge_index.size(1)
    src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
    if extra is not None:
        extra = extra.to(torch.int64).contiguous()
        _require(extra, 'extra', torch.int64, dev, (e,))
    out = torch.empty(e, 4 if extra is not None else 3, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().qcn_rel_geometry_f32(_ptr(pos_src), pos_src.size(1), _ptr(pos_dst), pos_dst.size(1), _ptr(head_src),
                                               _ptr(head_dst), _ptr(src), _ptr(dst), _ptr(extra), e, _ptr(out), _stream(dev)),
               'qcn_rel_geometry_f32')
    return out


CUMSUM_IDENTITY, CUMSUM_ELU1, CUMSUM_TANH_PI = 0, 1, 2


def proposal_cumsum(chunks: Sequence[torch.Tensor], dim: int, transform: int = CUMSUM_IDENTITY, offset: float = 0.0,
                    reciprocal: bool = False) -> torch.Tensor:
    """qcn_proposal_cumsum_f32: ``post(cumsum_t f(cat(chunks, -1).view(n, -1, dim)))`` -> [n, steps, dim]; chunks
    [n, steps_per_chunk * dim] each (the per-step increments one recurrent proposal step emits)."""
    return _op('proposal_cumsum')(list(chunks), int(dim), int(transform), float(offset), bool(reciprocal))


def _proposal_cumsum_fake(chunks, dim, transform, offset, reciprocal):
    n, w = chunks[0].shape
    return chunks[0].new_empty((n, len(chunks) * (w // dim), dim))


@_register('proposal_cumsum', '(Tensor[] chunks, int dim, int transform, float offset, bool reciprocal) -> Tensor', _proposal_cumsum_fake)
def _proposal_cumsum(chunks, dim, transform, offset, reciprocal):
    if not 1 <= len(chunks) <= 4:
        raise ValueError('1..4 chunks')
    dev = chunks[0].device
    n, w = chunks[0].shape
    if dim < 1 or w % dim:
        raise ValueError('chunk width %d is not a multiple of dim %d' % (w, dim))
    for i, c in enumerate(chunks):
        _require(c, 'chunks[%d]' % i, torch.float32, dev, (n, w))
    out = torch.empty(n, len(chunks) * (w // dim), dim, dtype=torch.float32, device=dev)
    ptrs = (ctypes.c_void_p * len(chunks))(*[c.data_ptr() for c in chunks])
    _lib.check(_lib.lib().qcn_proposal_cumsum_f32(ptrs, len(chunks), n, w // dim, int(dim), int(transform), float(offset),
                                                  int(bool(reciprocal)), _ptr(out), _stream(dev)), 'qcn_proposal_cumsum_f32')
    return out


def pack_mlp_heads(heads, mode: Optional[int] = None) -> torch.Tensor:
    """Pack for qcn_mlp_heads_f32 from a list of (w1 [128,128], b1, ln_w, ln_b, w2 [out,128], b2 [out]) tuples."""
    mats, vecs = [], []
    for w1, b1, ln_w, ln_b, w2, b2 in heads:
        out = w2.size(0)
        if tuple(w1.shape) != (C, C) or w2.size(1) != C or not 1 <= out <= C:
            raise ValueError('MLP head shapes: Linear(128,128) -> Linear(128,<=128)')
        w2p = w2.new_zeros(C, C)
        w2p[:out] = w2
        b2p = b2.new_zeros(C)
        b2p[:out] = b2
        mats += [w1, w2p]
        vecs += [b1, ln_w, ln_b, b2p]
    packed, scales = pack_chain_set(mats, mode=mode)
    return torch.cat([packed.reshape(-1)] + [v.detach().reshape(-1).float() for v in vecs] + [scale_table(scales)]).contiguous()


def mlp_heads(x: torch.Tensor, pack: torch.Tensor, out_dims, mode: Optional[int] = None) -> list:
    """qcn_mlp_heads_f32: [y_h [n, out_dims[h]]] for up to three MLPLayer heads over the same rows."""
    return list(_op('mlp_heads')(x, pack, [int(o) for o in out_dims], _mode(mode)))


@_register('mlp_heads', '(Tensor x, Tensor pack, int[] out_dims, int mode) -> Tensor[]',
           lambda x, pack, out_dims, mode: [x.new_empty((x.size(0), int(o))) for o in out_dims])
def _mlp_heads(x, pack, out_dims, mode):
    dev = x.device
    n, n_heads = x.size(0), len(out_dims)
    _require(x, 'x', torch.float32, dev, (n, C))
    _require(pack, 'pack', torch.float32, dev, (int(_lib.lib().qcn_mlp_heads_pack_floats(n_heads)),))
    ys = [torch.empty(n, int(o), dtype=torch.float32, device=dev) for o in out_dims]
    yp = [_ptr(y) for y in ys] + [None] * (3 - n_heads)
    od = (ctypes.c_int * n_heads)(*[int(o) for o in out_dims])
    tok = _prof_begin(PROF_GEMM, float(n) * n_heads * 2 * 2.0 * C * C, dev)
    _lib.check(_lib.lib().qcn_mlp_heads_f32(_ptr(x), n, _ptr(pack), n_heads, od, yp[0], yp[1], yp[2], int(mode), _stream(dev)),
               'qcn_mlp_heads_f32')
    _prof_end(tok, dev)
    return ys


def pack_gru_whh(weight_hh: torch.Tensor, h0: torch.Tensor, mode: Optional[int] = None) -> torch.Tensor:
    """``nn.GRU.weight_hh_l0`` ([384, 128], gates r, z, n) in the layout qcn_gru_f32 expects for the matmul mode in force
    (include/qcnet_hip.h, K10)."""
    mode = _mode(mode)       # bf16: the recurrence has no bf16 form and runs in the split mode (include/qcnet_hip.h)
    w = weight_hh.detach().float()
    if tuple(w.shape) != (3 * C, C):
        raise ValueError('weight_hh must be [384, 128]')
    if mode == MATMUL_EXACT_F32:
        return w.t().contiguous()
    dev = w.device
    planes, scales = [], []
    for g in range(3):
        wg = w[g * C:(g + 1) * C]
        mx = float(wg.abs().max())
        e = int(torch.frexp(wg.abs().max())[1]) if mx > 0 else 0
        sc = 2.0 ** (15 - e)
        ws = wg * sc
        hi = ws.half()
        planes.append(torch.stack([hi, (ws - hi.float()).half()]))          # [p, out, in]
        scales += [sc, 1.0 / sc]
    bound = max(1.0, float(h0.detach().abs().max()))
    e_h = int(torch.frexp(torch.tensor(bound))[1])                          # bound <= 2^e_h
    scales += [2.0 ** (14 - e_h), 2.0 ** (e_h - 14)]
    planes = torch.stack(planes)                                            # [g, p, out, in]
    lane = torch.arange(64, device=dev)
    out_idx = (32 * torch.arange(4, device=dev).view(4, 1, 1) + (lane & 31).view(1, 64, 1)).view(1, 4, 1, 1, 64, 1)
    in_idx = (16 * torch.arange(8, device=dev).view(8, 1, 1) + 8 * (lane >> 5).view(1, 64, 1)
              + torch.arange(8, device=dev).view(1, 1, 8)).view(1, 1, 8, 1, 64, 8)
    g_idx = torch.arange(3, device=dev).view(3, 1, 1, 1, 1, 1)
    p_idx = torch.arange(2, device=dev).view(1, 1, 1, 2, 1, 1)
    frag = planes[g_idx, p_idx, out_idx, in_idx]                            # [g, w, j, p, lane, i]
    packed = frag.contiguous().view(-1).view(torch.float32)
    out = torch.cat([packed, torch.tensor(scales, dtype=torch.float32, device=dev)])
    assert out.numel() == int(_lib.lib().qcn_gru_pack_floats())
    return out.contiguous()


def gru_last_hidden(x_seq: torch.Tensor, n_steps: int, wih_t: torch.Tensor, b_ih: torch.Tensor, whh_t: torch.Tensor,
                    b_hh: torch.Tensor, h0: torch.Tensor, mode: Optional[int] = None) -> torch.Tensor:
    """Final hidden state of a 1-layer GRU(128, 128) over ``n_steps`` steps.  ``x_seq`` [n_seq * n_steps, 128] with
    row = seq * n_steps + t; ``wih_t`` = pack_chain_matrices of the three gate matrices (r, z, n), ``b_ih`` [3, 128]; ``whh_t`` = pack_gru_whh(weight_hh_l0, h0);
    ``b_hh`` [384]; ``h0`` [128].  Input half: one qcn_ln_linear_f32 launch; recurrence: one qcn_gru_f32 launch."""
    dev = x_seq.device
    _require(x_seq, 'x_seq', torch.float32, dev, (None, C))
    if x_seq.size(0) % n_steps:
        raise ValueError('x_seq rows (%d) not a multiple of n_steps (%d)' % (x_seq.size(0), n_steps))
    (gi_r, gi_z, gi_n), _ = ln_linear(x_seq, None, None, wih_t, b_ih, mode=mode)
    return gru_recurrence(gi_r, gi_z, gi_n, n_steps, whh_t, b_hh, h0, mode=mode)


def gru_recurrence(gi_r: torch.Tensor, gi_z: torch.Tensor, gi_n: torch.Tensor, n_steps: int, whh_t: torch.Tensor,
                   b_hh: torch.Tensor, h0: torch.Tensor, mode: Optional[int] = None) -> torch.Tensor:
    """qcn_gru_f32 on precomputed input halves gi_* [n_seq * n_steps, 128]."""
    return _op('gru_recurrence')(gi_r, gi_z, gi_n, int(n_steps), whh_t, b_hh, h0, _mode(mode))


@_register('gru_recurrence', '(Tensor gi_r, Tensor gi_z, Tensor gi_n, int n_steps, Tensor whh_t, Tensor b_hh, Tensor h0, int mode) '
           '-> Tensor', lambda gi_r, gi_z, gi_n, n_steps, whh_t, b_hh, h0, mode: gi_r.new_empty((gi_r.size(0) // n_steps, C)))
def _gru_recurrence(gi_r, gi_z, gi_n, n_steps, whh_t, b_hh, h0, mode):
    dev = gi_r.device
    if n_steps < 1 or gi_r.size(0) % n_steps:
        raise ValueError('gi rows (%d) not a multiple of n_steps (%d)' % (gi_r.size(0), n_steps))
    n_seq = gi_r.size(0) // n_steps
    if mode == MATMUL_EXACT_F32:
        _require(whh_t, 'whh_t', torch.float32, dev, (C, 3 * C))
    else:
        mode = MATMUL_SPLIT_F16
        _require(whh_t, 'whh_t', torch.float32, dev, (int(_lib.lib().qcn_gru_pack_floats()),))
    _require(b_hh, 'b_hh', torch.float32, dev, (3 * C,))
    _require(h0, 'h0', torch.float32, dev, (C,))
    for name, t in (('gi_r', gi_r), ('gi_z', gi_z), ('gi_n', gi_n)):
        _require(t, name, torch.float32, dev, (n_seq * n_steps, C))
    h = torch.empty(n_seq, C, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().qcn_gru_f32(_ptr(gi_r), _ptr(gi_z), _ptr(gi_n), _ptr(whh_t), _ptr(b_hh), _ptr(h0), n_seq,
                                      int(n_steps), _ptr(h), int(mode), _stream(dev)), 'qcn_gru_f32')
    return h


LAYOUT_PLAIN, LAYOUT_MODE_MAJOR, LAYOUT_AGENT_MAJOR = 0, 1, 2


def attn_fused(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rhat: torch.Tensor, rowptr: torch.Tensor,
               col: torch.Tensor, eid: Optional[torch.Tensor], wk: torch.Tensor, wv: torch.Tensor, bv: torch.Tensor,
               rep: int = 1, layout: int = LAYOUT_PLAIN, col_shift: int = 0,
               row_index: Optional[torch.Tensor] = None, src_offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qcn_attn_fused_f32: attention propagate with the positional projections folded into the kernel.
    ``rowptr`` may be a window (view) of a larger graph's row pointer: its entries index ``col`` / ``eid`` / ``rhat``
    absolutely; ``col_shift`` re-bases the source ids of such a window.  ``row_index`` / ``src_offset`` (int32 [n_dst])
    give every destination its CSR row and a source re-basing of its own, see include/qcnet_hip.h."""
    return _op('attn_fused')(q, k, v, rhat, rowptr, col, eid, wk, wv, bv, int(rep), int(layout), int(col_shift), row_index,
                             src_offset)


@_register('attn_fused', '(Tensor q, Tensor k, Tensor v, Tensor rhat, Tensor rowptr, Tensor col, Tensor? eid, Tensor wk, Tensor wv, '
           'Tensor bv, int rep, int layout, int col_shift, Tensor? row_index, Tensor? src_offset) -> Tensor',
           lambda q, k, v, rhat, rowptr, col, eid, wk, wv, bv, rep, layout, col_shift, row_index, src_offset: torch.empty_like(q))
def _attn_fused(q, k, v, rhat, rowptr, col, eid, wk, wv, bv, rep, layout, col_shift, row_index, src_offset):
    dev = q.device
    n_dst, n_src, num_edges = q.size(0), k.size(0), col.numel()
    n_rows = rowptr.numel() - 1
    _require(q, 'q', torch.float32, dev, (n_dst, C))
    _require(k, 'k', torch.float32, dev, (n_src, C))
    _require(v, 'v', torch.float32, dev, (n_src, C))
    # with eid the feature table may be larger than the edge list (windows of an absolute-time relation)
    _require(rhat, 'rhat', torch.float32, dev, (None if eid is not None else num_edges, C))
    _require(rowptr, 'rowptr', torch.int32, dev, (n_rows + 1,))
    _require(col, 'col', torch.int32, dev, (num_edges,))
    if eid is not None:
        _require(eid, 'eid', torch.int32, dev, (num_edges,))
    _require(wk, 'wk', torch.float32, dev, (C, C))
    _require(wv, 'wv', torch.float32, dev, (C, C))
    _require(bv, 'bv', torch.float32, dev, (C,))
    if row_index is not None or src_offset is not None:
        if rep != 1 or layout != LAYOUT_PLAIN:
            raise ValueError('row_index / src_offset need layout 0, rep 1')
        if row_index is not None:
            _require(row_index, 'row_index', torch.int32, dev, (n_dst,))
        elif n_rows != n_dst:
            raise ValueError('n_rows (%d) != n_dst (%d) without row_index' % (n_rows, n_dst))
        if src_offset is not None:
            _require(src_offset, 'src_offset', torch.int32, dev, (n_dst,))
    elif n_rows * rep != n_dst or (layout == LAYOUT_PLAIN and rep != 1):
        raise ValueError('n_rows (%d) * rep (%d) != n_dst (%d)' % (n_rows, rep, n_dst))
    out = torch.empty(n_dst, C, dtype=torch.float32, device=dev)
    # algorithmic bytes (r-projection fused): this kernel visits every (edge, mode) pair
    tok = _prof_begin(PROF_ATTN_FUSED, 1552.0 * num_edges * rep + 1024.0 * n_dst, dev)
    _lib.check(_lib.lib().qcn_attn_fused_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(rhat), _ptr(rowptr), _ptr(col), _ptr(eid),
                                             _ptr(wk), _ptr(wv), _ptr(bv), n_dst, n_src, n_rows, num_edges, int(rep),
                                             int(layout), int(col_shift), _ptr(row_index), _ptr(src_offset), _ptr(out),
                                             _stream(dev)), 'qcn_attn_fused_f32')
    _prof_end(tok, dev)
    return out


# ------------------------------------------------------------------------------------------------
# profiling hooks for bench.py: per-launch event timing on the stream the kernel is launched on (torch's current stream —
# the one every op of this module launches on).  Python-side state; libqcnet_hip.so itself keeps none.
# ------------------------------------------------------------------------------------------------
PROF_ATTN, PROF_ATTN_FUSED, PROF_FOURIER, PROF_GEMM = 0, 1, 2, 3
_PROF_MASK = 0                  # process-wide switch (bench.py); the log is shared by all host threads under _PROF_LOCK
_PROF_LOG: list = []
_PROF_LOCK = threading.Lock()


def prof_enable(on, kinds=None) -> None:
    """Record event timings of the given kernel families (``kinds``: iterable of PROF_*; default all).  Each recorded
    launch costs two event records on its stream, so time only what is reported."""
    global _PROF_MASK
    _PROF_MASK = 0 if not on else (~0 if kinds is None else sum(1 << int(k) for k in kinds))


def _prof_begin(kind: int, alg_units: float, dev):
    if not (_PROF_MASK >> kind) & 1:
        return None
    start = torch.cuda.Event(enable_timing=True)
    start.record(torch.cuda.current_stream(dev))
    return (kind, alg_units, start)


def _prof_end(tok, dev) -> None:
    if tok is None:
        return
    stop = torch.cuda.Event(enable_timing=True)
    stop.record(torch.cuda.current_stream(dev))
    with _PROF_LOCK:
        _PROF_LOG.append(tok + (stop,))


def prof_collect(kind: int = -1) -> Tuple[int, float, float]:
    """(launches, summed kernel ms, summed algorithmic units) of one kernel family (-1: all), then drops those records.
    Units: bytes for the attention kernels (2064 E + 1024 n_dst materialised, 1552 E + 1024 n_dst fused,
    1040 E + 1024 n_dst without positional terms), flop for the MFMA chain kernels."""
    n, ms, units = 0, 0.0, 0.0
    with _PROF_LOCK:
        mine = [rec for rec in _PROF_LOG if kind < 0 or rec[0] == kind]
        _PROF_LOG[:] = [rec for rec in _PROF_LOG if not (kind < 0 or rec[0] == kind)]
    for rec in mine:
        rec[3].synchronize()
        n += 1
        ms += rec[2].elapsed_time(rec[3])
        units += rec[1]
    return n, ms, units


RHAT_SPLIT_SCALE_LOG2 = 11      # |rhat| <= sqrt(127): rhat * 2^11 stays inside f16


def split_rows(x: torch.Tensor, scale_log2: int = RHAT_SPLIT_SCALE_LOG2) -> torch.Tensor:
    """qcn_split_rows_f16: fp32 rows [N, 128] -> split-f16 rows [N, 256] = [hi | lo] with hi + lo = x * 2^scale_log2."""
    return _op('split_rows')(x, int(scale_log2))


@_register('split_rows', '(Tensor x, int scale_log2) -> Tensor',
           lambda x, scale_log2: x.new_empty((x.size(0), 2 * C), dtype=torch.float16))
def _split_rows(x, scale_log2):
    dev = x.device
    _require(x, 'x', torch.float32, dev, (None, C))
    y = torch.empty(x.size(0), 2 * C, dtype=torch.float16, device=dev)
    _lib.check(_lib.lib().qcn_split_rows_f16(_ptr(x), x.size(0), int(scale_log2), _ptr(y), _stream(dev)),
               'qcn_split_rows_f16')
    return y


def _split_scaled(w: torch.Tensor):
    """(hi, lo, sw): w * 2^sw = hi + lo in f16 with max |w| * 2^sw in [2^14, 2^15)."""
    w = w.detach().float()
    mx = float(w.abs().max())
    sw = 0 if mx == 0.0 else 15 - int(torch.frexp(torch.tensor(mx))[1])
    ws = torch.ldexp(w, torch.tensor(sw))
    hi = ws.half()
    lo = (ws - hi.float()).half()
    return hi, lo, sw


def pack_attn_modes(wk: torch.Tensor, wv: torch.Tensor, mode: Optional[int] = None) -> torch.Tensor:
    """Weight pack of qcn_attn_modes_f32 / qcn_attn_modes_bf16 (include/qcnet_hip.h): per head the MFMA fragments of wk /
    wv (the positional projections with the LN_r scale folded in, [128 out, 128 in]).  Split mode (also used by the exact
    mode): hi / lo f16 planes, then the two fp32 factors that undo the weight and rhat scaling.  bf16 mode: one bf16 plane
    (round to nearest even), no scaling."""
    if _mode(mode) == MATMUL_BF16:
        frags = [w.detach().float().bfloat16().view(8, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).reshape(8, 4, 64, 8)
                 for w in (wk, wv)]                                         # [head, ks, lane, 8]
        pack = torch.stack(frags, dim=1).contiguous().view(torch.uint8).reshape(-1)   # [head, mat, ks, lane, 8]
        assert pack.numel() == _lib.lib().qcn_attn_modes_bf16_pack_bytes()
        return pack
    frags, scales = [], []
    for w in (wk, wv):
        hi, lo, sw = _split_scaled(w)
        parts = [t.view(8, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).reshape(8, 4, 64, 8) for t in (hi, lo)]
        frags.append(torch.stack(parts, dim=2))                       # [head, ks, part, lane, 8]
        scales.append(2.0 ** -(sw + RHAT_SPLIT_SCALE_LOG2))
    body = torch.stack(frags, dim=1).contiguous()                       # [head, mat, ks, part, lane, 8]
    tail = torch.tensor(scales, dtype=torch.float32, device=body.device)
    pack = torch.cat([body.view(torch.uint8).reshape(-1), tail.view(torch.uint8).reshape(-1)]).contiguous()
    assert pack.numel() == _lib.lib().qcn_attn_modes_pack_bytes()
    return pack


def attn_modes(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rhat_op: torch.Tensor, rowptr: torch.Tensor,
               col: torch.Tensor, eid: Optional[torch.Tensor], wpack: torch.Tensor, bv: torch.Tensor,
               rep: int, layout: int, col_shift: int = 0) -> torch.Tensor:
    """qcn_attn_modes_f32 / qcn_attn_modes_bf16: attention propagate of a mode-replicated relation (rep <= 6 queries per CSR
    row share its edges) on the matrix cores; same result as attn_fused on the same graph (bf16: with the positional
    projections' operands rounded to bf16).  ``rhat_op``: split-f16 rows [., 256] (float16) with the split pack, or bf16
    rows [., 128] (bfloat16) with the bf16 pack."""
    return _op('attn_modes')(q, k, v, rhat_op, rowptr, col, eid, wpack, bv, int(rep), int(layout), int(col_shift))


@_register('attn_modes', '(Tensor q, Tensor k, Tensor v, Tensor rhat_op, Tensor rowptr, Tensor col, Tensor? eid, Tensor wpack, '
           'Tensor bv, int rep, int layout, int col_shift) -> Tensor',
           lambda q, k, v, rhat_op, rowptr, col, eid, wpack, bv, rep, layout, col_shift: torch.empty_like(q))
def _attn_modes(q, k, v, rhat_op, rowptr, col, eid, wpack, bv, rep, layout, col_shift):
    dev = q.device
    lib = _lib.lib()
    n_dst, n_src, num_edges = q.size(0), k.size(0), col.numel()
    n_rows = rowptr.numel() - 1
    bf16 = rhat_op.dtype == torch.bfloat16
    _require(q, 'q', torch.float32, dev, (n_dst, C))
    _require(k, 'k', torch.float32, dev, (n_src, C))
    _require(v, 'v', torch.float32, dev, (n_src, C))
    n_feat = None if eid is not None else num_edges
    if bf16:
        _require(rhat_op, 'rhat_bf16', torch.bfloat16, dev, (n_feat, C))
        _require(wpack, 'wpack', torch.uint8, dev, (lib.qcn_attn_modes_bf16_pack_bytes(),))
    else:
        _require(rhat_op, 'rhat_split', torch.float16, dev, (n_feat, 2 * C))
        _require(wpack, 'wpack', torch.uint8, dev, (lib.qcn_attn_modes_pack_bytes(),))
    _require(rowptr, 'rowptr', torch.int32, dev, (n_rows + 1,))
    _require(col, 'col', torch.int32, dev, (num_edges,))
    if eid is not None:
        _require(eid, 'eid', torch.int32, dev, (num_edges,))
    _require(bv, 'bv', torch.float32, dev, (C,))
    if n_rows * rep != n_dst or layout not in (LAYOUT_MODE_MAJOR, LAYOUT_AGENT_MAJOR):
        raise ValueError('n_rows (%d) * rep (%d) != n_dst (%d) or plain layout' % (n_rows, rep, n_dst))
    out = torch.empty(n_dst, C, dtype=torch.float32, device=dev)
    fn, name = (lib.qcn_attn_modes_bf16, 'qcn_attn_modes_bf16') if bf16 else (lib.qcn_attn_modes_f32, 'qcn_attn_modes_f32')
    # algorithmic bytes: every base edge is processed once for all rep modes (rhat row + k row + v row + indices); the bf16
    # form reads a 256-byte rhat row (1296 E + 1024 n_dst)
    tok = _prof_begin(PROF_ATTN_FUSED, (1296.0 if bf16 else 1552.0) * num_edges + 1024.0 * n_dst, dev)
    _lib.check(fn(_ptr(q), _ptr(k), _ptr(v), _ptr(rhat_op), _ptr(rowptr), _ptr(col), _ptr(eid), _ptr(wpack), _ptr(bv), n_dst,
                  n_src, n_rows, num_edges, rhat_op.size(0), int(rep), int(layout), int(col_shift), _ptr(out), _stream(dev)),
               name)
    _prof_end(tok, dev)
    return out


def attn_modes_store(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rhat_op: torch.Tensor, rowptr: torch.Tensor,
                     col: torch.Tensor, eid: Optional[torch.Tensor], wpack: torch.Tensor, bv: torch.Tensor,
                     rep: int, layout: int, col_shift: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """qcn_attn_modes_store_f32 / _bf16: ``attn_modes`` that also returns the per-edge projections
    ``(out, kk [slots / 4, 8, 4, 16], vv [slots / 4, 8, 16, 4])`` — (include/qcnet_hip.h; ``attn_modes_table_rows`` gives the
    [E, 128] rows back) — for later ``attn_modes_cached`` calls on the same sources with the same ``rowptr``."""
    return _op('attn_modes_store')(q, k, v, rhat_op, rowptr, col, eid, wpack, bv, int(rep), int(layout), int(col_shift))


def _attn_modes_store_fake(q, k, v, rhat_op, rowptr, col, eid, wpack, bv, rep, layout, col_shift):
    slots = attn_modes_table_slots(col.numel(), rowptr.numel() - 1)
    return torch.empty_like(q), q.new_empty((slots // 4, 8, 4, 16)), q.new_empty((slots // 4, 8, 16, 4))


@_register('attn_modes_store', '(Tensor q, Tensor k, Tensor v, Tensor rhat_op, Tensor rowptr, Tensor col, Tensor? eid, '
           'Tensor wpack, Tensor bv, int rep, int layout, int col_shift) -> (Tensor, Tensor, Tensor)', _attn_modes_store_fake)
def _attn_modes_store(q, k, v, rhat_op, rowptr, col, eid, wpack, bv, rep, layout, col_shift):
    dev = q.device
    lib = _lib.lib()
    n_dst, n_src, num_edges = q.size(0), k.size(0), col.numel()
    n_rows = rowptr.numel() - 1
    bf16 = rhat_op.dtype == torch.bfloat16
    _require(q, 'q', torch.float32, dev, (n_dst, C))
    _require(k, 'k', torch.float32, dev, (n_src, C))
    _require(v, 'v', torch.float32, dev, (n_src, C))
    n_feat = None if eid is not None else num_edges
    if bf16:
        _require(rhat_op, 'rhat_bf16', torch.bfloat16, dev, (n_feat, C))
        _require(wpack, 'wpack', torch.uint8, dev, (lib.qcn_attn_modes_bf16_pack_bytes(),))
    else:
        _require(rhat_op, 'rhat_split', torch.float16, dev, (n_feat, 2 * C))
        _require(wpack, 'wpack', torch.uint8, dev, (lib.qcn_attn_modes_pack_bytes(),))
    _require(rowptr, 'rowptr', torch.int32, dev, (n_rows + 1,))
    _require(col, 'col', torch.int32, dev, (num_edges,))
    if eid is not None:
        _require(eid, 'eid', torch.int32, dev, (num_edges,))
    _require(bv, 'bv', torch.float32, dev, (C,))
    if n_rows * rep != n_dst or layout not in (LAYOUT_MODE_MAJOR, LAYOUT_AGENT_MAJOR):
        raise ValueError('n_rows (%d) * rep (%d) != n_dst (%d) or plain layout' % (n_rows, rep, n_dst))
    out = torch.empty(n_dst, C, dtype=torch.float32, device=dev)
    slots = attn_modes_table_slots(num_edges, n_rows)
    if slots >= (1 << 23):
        raise ValueError('%d edges: the projection tables address slots by 32-bit byte offsets (< 2^23 slots)' % num_edges)
    kk = torch.empty(slots // 4, 8, 4, 16, dtype=torch.float32, device=dev)
    vv = torch.empty(slots // 4, 8, 16, 4, dtype=torch.float32, device=dev)
    fn, name = ((lib.qcn_attn_modes_store_bf16, 'qcn_attn_modes_store_bf16') if bf16
                else (lib.qcn_attn_modes_store_f32, 'qcn_attn_modes_store_f32'))
    # algorithmic bytes: the fused form's reads + the two projection rows written per edge
    tok = _prof_begin(PROF_ATTN_FUSED, ((1296.0 if bf16 else 1552.0) + 1024.0) * num_edges + 1024.0 * n_dst, dev)
    _lib.check(fn(_ptr(q), _ptr(k), _ptr(v), _ptr(rhat_op), _ptr(rowptr), _ptr(col), _ptr(eid), _ptr(wpack), _ptr(bv), n_dst,
                  n_src, n_rows, num_edges, rhat_op.size(0), int(rep), int(layout), int(col_shift), _ptr(out), _ptr(kk),
                  _ptr(vv), slots, _stream(dev)), name)
    _prof_end(tok, dev)
    return out, kk, vv


def attn_modes_table_slots(num_edges: int, n_rows: int) -> int:
    """qcn_attn_modes_table_slots: slots per head plane of the projection tables (rows start on multiples of four slots)."""
    return (int(num_edges) + 3 * int(n_rows) + 7) & ~3


def attn_modes_table_rows(kk: torch.Tensor, vv: torch.Tensor, rowptr: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """The tables of ``attn_modes_store`` as plain rows ``(kk_e, vv_e)`` [E, 128] in CSR order (tests / debugging: a gather
    by the slot rule of include/qcnet_hip.h, torch ops only)."""
    rp = rowptr.long()
    deg = rp[1:] - rp[:-1]
    rows = torch.repeat_interleave(torch.arange(deg.numel(), device=rp.device), deg)
    start = (rp[:-1] + 3 * torch.arange(deg.numel(), device=rp.device) + 3) & ~3
    j = torch.arange(int(rp[0]), int(rp[-1]), device=rp.device)
    slot = start[rows] + (j - rp[:-1][rows])
    kk_rows = kk[slot // 4, :, slot % 4, :].reshape(-1, C)        # [E, head, 16]
    vv_rows = vv[slot // 4, :, :, slot % 4].reshape(-1, C)
    return kk_rows, vv_rows


def attn_modes_cached(q: torch.Tensor, kk: torch.Tensor, vv: torch.Tensor, rowptr: torch.Tensor, bv: torch.Tensor,
                      rep: int, layout: int, num_edges: Optional[int] = None) -> torch.Tensor:
    """qcn_attn_modes_cached_f32: the propagate of a mode-replicated relation from the stored per-edge projections of an
    ``attn_modes_store`` call on the same sources, graph and layer (bit-identical to ``attn_modes`` for the same q)."""
    if num_edges is None:       # upper bound from the table size (the kernel reads the true count from rowptr)
        num_edges = max(0, 4 * kk.size(0) - 3 * (rowptr.numel() - 1) - 7)
    return _op('attn_modes_cached')(q, kk, vv, rowptr, bv, int(rep), int(layout), int(num_edges))


@_register('attn_modes_cached', '(Tensor q, Tensor kk, Tensor vv, Tensor rowptr, Tensor bv, int rep, int layout, int num_edges) -> Tensor',
           lambda q, kk, vv, rowptr, bv, rep, layout, num_edges: torch.empty_like(q))
def _attn_modes_cached(q, kk, vv, rowptr, bv, rep, layout, num_edges):
    dev = q.device
    n_dst, num_edges, slots = q.size(0), int(num_edges), 4 * kk.size(0)
    n_rows = rowptr.numel() - 1
    _require(q, 'q', torch.float32, dev, (n_dst, C))
    _require(kk, 'kk', torch.float32, dev, (slots // 4, 8, 4, 16))
    _require(vv, 'vv', torch.float32, dev, (slots // 4, 8, 16, 4))
    if slots < attn_modes_table_slots(num_edges, n_rows):
        raise ValueError('tables of %d slots for %d edges in %d rows' % (slots, num_edges, n_rows))
    _require(rowptr, 'rowptr', torch.int32, dev, (n_rows + 1,))
    _require(bv, 'bv', torch.float32, dev, (C,))
    if n_rows * rep != n_dst or layout not in (LAYOUT_MODE_MAJOR, LAYOUT_AGENT_MAJOR):
        raise ValueError('n_rows (%d) * rep (%d) != n_dst (%d) or plain layout' % (n_rows, rep, n_dst))
    out = torch.empty(n_dst, C, dtype=torch.float32, device=dev)
    # algorithmic bytes: the two projection rows per edge, q in and out per destination
    tok = _prof_begin(PROF_ATTN_FUSED, 1024.0 * num_edges + 1024.0 * n_dst, dev)
    _lib.check(_lib.lib().qcn_attn_modes_cached_f32(_ptr(q), _ptr(kk), _ptr(vv), _ptr(rowptr), _ptr(bv), n_dst, n_rows,
                                                   num_edges, slots, int(rep), int(layout), _ptr(out), _stream(dev)),
               'qcn_attn_modes_cached_f32')
    _prof_end(tok, dev)
    return out


def attn_modes_supported(n_dst: int, n_src: int, n_feat_rows: int) -> bool:
    """qcn_attn_modes_* address rows by 32-bit byte offsets: every row count must stay below 2^23 (beyond that
    AttentionLayer falls back to qcn_attn_fused_f32, which handles the same graph)."""
    return max(int(n_dst), int(n_src), int(n_feat_rows)) < (1 << 23)


# ------------------------------------------------------------------------------------------------
# evaluation metrics and frame transforms (N3)
# ------------------------------------------------------------------------------------------------
METRIC_RECORD = 16  # QCN_METRIC_RECORD
METRIC_COLUMNS = {'brier_fde': 0, 'brier_ade': 1, 'ade_fde': 2, 'ade_ade': 3, 'ahe_fde': 4, 'ahe_ade': 5, 'fde': 6, 'fhe': 7,
                  'miss_fde': 8, 'miss_maxde': 9, 'probmiss_fde': 10, 'probmiss_maxde': 11, 'counted': 12,
                  'best_fde_mode': 13, 'best_ade_mode': 14, 'last_valid_step': 15}
HEAD_NONE, HEAD_FROM_MOTION = -1, -2


def _as_u8(mask: Optional[torch.Tensor], name: str, device, shape) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        mask = mask.contiguous().view(torch.uint8)
    elif mask.dtype != torch.uint8:
        mask = (mask != 0).contiguous().view(torch.uint8)
    mask = mask.contiguous()
    _require(mask, name, torch.uint8, device, shape)
    return mask


def motion_metrics(pred: torch.Tensor, target: torch.Tensor, prob: Optional[torch.Tensor] = None,
                   valid_mask: Optional[torch.Tensor] = None, select: Optional[torch.Tensor] = None, pos_dim: Optional[int] = None,
                   pred_head: int = HEAD_NONE, target_head: int = -1, max_guesses: int = 6,
                   keep_invalid_final_step: bool = True, miss_threshold: float = 2.0) -> torch.Tensor:
    """qcn_motion_metrics_f32: per-sample records [n, 16] (columns: METRIC_COLUMNS) of every metric class's state update.
    pred [n, K, T, Dp], target [n, T, Dt], prob [n, K], valid_mask [n, T], select [n]."""
    if pred.dim() != 4 or target.dim() != 3:
        raise ValueError('pred must be [n, modes, steps, dim] and target [n, steps, dim]')
    dp, dt = pred.size(3), target.size(2)
    if pos_dim is None:
        pos_dim = dp if pred_head < 0 else dp - 1
    if target_head < 0:
        target_head = dt + target_head
    return _op('motion_metrics')(pred, target, prob, valid_mask, select, int(pos_dim), int(pred_head), int(target_head),
                                 int(max_guesses), bool(keep_invalid_final_step), float(miss_threshold))


@_register('motion_metrics', '(Tensor pred, Tensor target, Tensor? prob, Tensor? valid_mask, Tensor? select, int pos_dim, '
           'int pred_head, int target_head, int max_guesses, bool keep_invalid_final_step, float miss_threshold) -> Tensor',
           lambda pred, *a: pred.new_empty((pred.size(0), 16)))
def _motion_metrics(pred, target, prob, valid_mask, select, pos_dim, pred_head, target_head, max_guesses,
                    keep_invalid_final_step, miss_threshold):
    dev = pred.device
    n, k, t, dp = pred.shape
    dt = target.size(2)
    pred = pred.contiguous()
    target = target.contiguous()
    _require(pred, 'pred', torch.float32, dev, (n, k, t, dp))
    _require(target, 'target', torch.float32, dev, (n, t, dt))
    if prob is not None:
        prob = prob.contiguous()
        _require(prob, 'prob', torch.float32, dev, (n, k))
    valid_mask = _as_u8(valid_mask, 'valid_mask', dev, (n, t))
    select = _as_u8(select, 'select', dev, (n,))
    record = torch.empty(n, METRIC_RECORD, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().qcn_motion_metrics_f32(_ptr(pred), dp, _ptr(prob), _ptr(target), dt, _ptr(valid_mask), _ptr(select), n, k,
                                                 t, int(pos_dim), int(pred_head), int(target_head), int(max_guesses),
                                                 int(bool(keep_invalid_final_step)), float(miss_threshold), _ptr(record),
                                                 _stream(dev)), 'qcn_motion_metrics_f32')
    return record


def to_agent_frame(position: torch.Tensor, heading: torch.Tensor, origin_step: int, first_step: int,
                   n_steps: Optional[int] = None) -> torch.Tensor:
    """qcn_to_agent_frame_f32: [A, n_steps, 4] targets (x, y, z, heading) in the frame of each agent's pose at
    ``origin_step`` (predictors/temp_ens_learn_agg.py:104-121)."""
    if n_steps is None:
        n_steps = position.size(1) - first_step
    return _op('to_agent_frame')(position, heading, int(origin_step), int(first_step), int(n_steps))


@_register('to_agent_frame', '(Tensor position, Tensor heading, int origin_step, int first_step, int n_steps) -> Tensor',
           lambda position, heading, origin_step, first_step, n_steps: position.new_empty((position.size(0), n_steps, 4)))
def _to_agent_frame(position, heading, origin_step, first_step, n_steps):
    dev = position.device
    a, total, pd = position.shape
    _require(position, 'position', torch.float32, dev, (a, total, pd))
    _require(heading, 'heading', torch.float32, dev, (a, total))
    out = torch.empty(a, n_steps, 4, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().qcn_to_agent_frame_f32(_ptr(position), _ptr(heading), a, total, pd, int(origin_step), int(first_step),
                                                 int(n_steps), _ptr(out), _stream(dev)), 'qcn_to_agent_frame_f32')
    return out


def to_world_frame(traj: torch.Tensor, position: torch.Tensor, heading: torch.Tensor, origin_step: int,
                   rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qcn_to_world_frame_f32: [n_rows, K, T, 2] world-frame trajectories of the agents ``rows`` (all agents when None) from
    agent-frame ``traj`` [A, K, T, D] (predictors/qcnet_sliding.py:179-192)."""
    return _op('to_world_frame')(traj, position, heading, int(origin_step), rows)


def _to_world_frame_fake(traj, position, heading, origin_step, rows):
    n = traj.size(0) if rows is None else rows.numel()
    return traj.new_empty((n, traj.size(1), traj.size(2), 2))


@_register('to_world_frame', '(Tensor traj, Tensor position, Tensor heading, int origin_step, Tensor? rows) -> Tensor',
           _to_world_frame_fake)
def _to_world_frame(traj, position, heading, origin_step, rows):
    dev = traj.device
    a, k, t, d = traj.shape
    traj = traj.contiguous()
    _require(traj, 'traj', torch.float32, dev, (a, k, t, d))
    _require(position, 'position', torch.float32, dev, (a, None, None))
    _require(heading, 'heading', torch.float32, dev, (a, position.size(1)))
    if rows is not None:
        rows = rows.contiguous()
        _require(rows, 'rows', torch.int64, dev, (None,))
        if rows.numel() and (int(rows.min()) < 0 or int(rows.max()) >= a):
            raise ValueError('rows out of range')
    n_rows = a if rows is None else rows.numel()
    out = torch.empty(n_rows, k, t, 2, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().qcn_to_world_frame_f32(_ptr(traj), d, _ptr(position), _ptr(heading), _ptr(rows), n_rows, k * t,
                                                 position.size(1), position.size(2), int(origin_step), _ptr(out), _stream(dev)),
               'qcn_to_world_frame_f32')
    return out


def nms_select(traj: torch.Tensor, conf: torch.Tensor, gt: torch.Tensor, cand_per_window: int = 6, n_select: int = 6,
               radius: float = 2.5, miss_threshold: float = 2.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """qcn_nms_select_f32: traj [n, M, T, 2], conf [n, M], gt [n, T, 2] -> (selected [n, W, n_select] int32,
    score [n, W, 3] = min FDE / min ADE / miss) for the window sizes 1..W = M // cand_per_window (utils/nms.py:11-38)."""
    sel, score = _op('nms_select')(traj, conf, gt, int(cand_per_window), int(n_select), float(radius), float(miss_threshold))
    return sel, score


def _nms_select_fake(traj, conf, gt, cand_per_window, n_select, radius, miss_threshold):
    n, w = traj.size(0), traj.size(1) // cand_per_window
    return traj.new_empty((n, w, n_select), dtype=torch.int32), traj.new_empty((n, w, 3))


@_register('nms_select', '(Tensor traj, Tensor conf, Tensor gt, int cand_per_window, int n_select, float radius, '
           'float miss_threshold) -> (Tensor, Tensor)', _nms_select_fake)
def _nms_select(traj, conf, gt, cand_per_window, n_select, radius, miss_threshold):
    dev = traj.device
    n, m, t, _ = traj.shape
    traj, conf, gt = traj.contiguous(), conf.contiguous(), gt.contiguous()
    _require(traj, 'traj', torch.float32, dev, (n, m, t, 2))
    _require(conf, 'conf', torch.float32, dev, (n, m))
    _require(gt, 'gt', torch.float32, dev, (n, t, 2))
    w = m // int(cand_per_window)
    if w < 1:
        raise ValueError('fewer candidates than one window holds')
    selected = torch.empty(n, w, n_select, dtype=torch.int32, device=dev)
    score = torch.empty(n, w, 3, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().qcn_nms_select_f32(_ptr(traj), _ptr(conf), _ptr(gt), n, m, t, w, int(cand_per_window), int(n_select),
                                             float(radius), float(miss_threshold), _ptr(selected), _ptr(score), _stream(dev)),
               'qcn_nms_select_f32')
    return selected, score


# ------------------------------------------------------------------------------------------------
# mask enumerations and edge merge (G3 / G4)
# ------------------------------------------------------------------------------------------------
_NO_BAND = (-(1 << 31), (1 << 31) - 1)


def mask_pairs(mask_src: torch.Tensor, mask_dst: torch.Tensor, delta_lo: Optional[int] = None,
               delta_hi: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """qcn_mask_pairs_count / _fill: the edges (b * n + i, b * m + j) with mask_src[b, i] & mask_dst[b, j] (and
    delta_lo <= j - i <= delta_hi when a band is given) in row-major order — dense_to_sparse of the outer-product adjacency
    without materialising it.  Returns ``(edge_index [2, E] int64, offset [B * n + 1] int64)``; one host synchronisation."""
    lo = _NO_BAND[0] if delta_lo is None else int(delta_lo)
    hi = _NO_BAND[1] if delta_hi is None else int(delta_hi)
    ei, off = _op('mask_pairs')(mask_src, mask_dst, lo, hi)
    return ei, off


def _mask_pairs_fake(mask_src, mask_dst, lo, hi):
    return (mask_src.new_empty((2, _dyn()), dtype=torch.int64),
            mask_src.new_empty((mask_src.size(0) * mask_src.size(1) + 1,), dtype=torch.int64))


@_register('mask_pairs', '(Tensor mask_src, Tensor mask_dst, int delta_lo, int delta_hi) -> (Tensor, Tensor)', _mask_pairs_fake)
def _mask_pairs(mask_src, mask_dst, lo, hi):
    dev = mask_src.device
    if mask_src.dim() != 2 or mask_dst.dim() != 2 or mask_src.size(0) != mask_dst.size(0):
        raise ValueError('masks must be [B, n] and [B, m]')
    b, n = mask_src.shape
    m = mask_dst.size(1)
    ms = _as_u8(mask_src, 'mask_src', dev, (b, n))
    md = _as_u8(mask_dst, 'mask_dst', dev, (b, m))
    lib = _lib.lib()
    st = _stream(dev)
    rows = b * n
    count = torch.zeros(rows + 1, dtype=torch.int32, device=dev)
    offset = torch.empty(rows + 1, dtype=torch.int64, device=dev)
    _lib.check(lib.qcn_mask_pairs_count(_ptr(ms), _ptr(md), b, n, m, lo, hi, _ptr(count), st), 'qcn_mask_pairs_count')
    ws_bytes = lib.qcn_scan_workspace_bytes(rows)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.qcn_exclusive_scan_i32(_ptr(count), rows, _ptr(offset), _ptr(ws), ws_bytes, st), 'qcn_exclusive_scan_i32')
    num_edges = int(offset[-1].item())
    edge_index = torch.empty(2, num_edges, dtype=torch.int64, device=dev)
    if num_edges > 0:
        _lib.check(lib.qcn_mask_pairs_fill(_ptr(ms), _ptr(md), b, n, m, lo, hi, _ptr(offset), _ptr(edge_index[0]),
                                           _ptr(edge_index[1]), st), 'qcn_mask_pairs_fill')
    return edge_index, offset


def coalesce_max(edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor], num_nodes: int
                 ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """qcn_coalesce_count / _fill: edges sorted by (src, dst), duplicates merged with the largest uint8 attribute."""
    out = _op('coalesce_max')(edge_index, edge_attr, int(num_nodes))
    return out[0], (out[1] if edge_attr is not None else None)


def _coalesce_fake(edge_index, edge_attr, num_nodes):
    n = _dyn()
    return [edge_index.new_empty((2, n))] + ([edge_attr.new_empty((n,))] if edge_attr is not None else [])


@_register('coalesce_max', '(Tensor edge_index, Tensor? edge_attr, int num_nodes) -> Tensor[]', _coalesce_fake)
def _coalesce_max(edge_index, edge_attr, num_nodes):
    dev = edge_index.device
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError('edge_index must be [2, E]')
    e = edge_index.size(1)
    edge_index = edge_index.contiguous()
    _require(edge_index, 'edge_index', torch.int64, dev, (2, e))
    if edge_attr is not None:
        edge_attr = edge_attr.contiguous()
        _require(edge_attr, 'edge_attr', torch.uint8, dev, (e,))
    if e and (int(edge_index.min()) < 0 or int(edge_index.max()) >= int(num_nodes)):
        raise IndexError('edge_index out of range for num_nodes %d' % num_nodes)
    lib = _lib.lib()
    st = _stream(dev)
    ws_bytes = lib.qcn_coalesce_workspace_bytes(e)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    offset = torch.empty(e + 1, dtype=torch.int64, device=dev)
    _lib.check(lib.qcn_coalesce_count(_ptr(edge_index[0]), _ptr(edge_index[1]), _ptr(edge_attr), e, int(num_nodes), _ptr(ws), ws_bytes,
                                      _ptr(offset), st), 'qcn_coalesce_count')
    n_out = int(offset[-1].item())
    out_index = torch.empty(2, n_out, dtype=torch.int64, device=dev)
    out_attr = torch.empty(n_out, dtype=torch.uint8, device=dev) if edge_attr is not None else None
    if e:
        _lib.check(lib.qcn_coalesce_fill(_ptr(ws), _ptr(offset), e, int(num_nodes), _ptr(out_index[0]), _ptr(out_index[1]),
                                         _ptr(out_attr), st), 'qcn_coalesce_fill')
    return [out_index] + ([out_attr] if out_attr is not None else [])
