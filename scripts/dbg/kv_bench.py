"""attn_modes (recompute) vs attn_modes_store + attn_modes_cached at the cfg3 decoder's relation sizes (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd import ops
dev = torch.device('cuda:0')
K = 6
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, n_rows, n_src, per_row in (('pl2m', 16000, 4800, 127), ('t2m', 16000, 108800, 30), ('a2m', 16000, 16000, 44)):
    g = torch.Generator(device=dev).manual_seed(0)
    E = n_rows * per_row
    rowptr = (torch.arange(n_rows + 1, device=dev) * per_row).to(torch.int32)
    col = torch.randint(0, n_src, (E,), device=dev, generator=g).to(torch.int32)
    q = torch.randn(n_rows * K, 128, device=dev, generator=g)
    k = torch.randn(n_src, 128, device=dev, generator=g); v = torch.randn(n_src, 128, device=dev, generator=g)
    rhat = ops.row_normalize(torch.randn(E, 128, device=dev, generator=g))
    rs = ops.split_rows(rhat)
    wk = torch.randn(128, 128, device=dev, generator=g) * 0.05; wv = torch.randn(128, 128, device=dev, generator=g) * 0.05
    wpack = ops.pack_attn_modes(wk, wv, mode=ops.MATMUL_SPLIT_F16)
    bv = torch.zeros(128, device=dev)
    args = (k, v, rs, rowptr, col, None, wpack, bv, K, ops.LAYOUT_AGENT_MAJOR)
    t0 = timeit(lambda: ops.attn_modes(q, *args))
    t1 = timeit(lambda: ops.attn_modes_store(q, *args))
    _, kk, vv = ops.attn_modes_store(q, *args)
    t2 = timeit(lambda: ops.attn_modes_cached(q, kk, vv, rowptr, bv, K, ops.LAYOUT_AGENT_MAJOR))
    print('%-5s E %8d: recompute %7.1f us, store %7.1f us, cached %7.1f us (%.2f TB/s of tables)' % (name, E, t0, t1, t2, E * 1024 / t2 / 1e6))
