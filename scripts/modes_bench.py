"""qcn_attn_modes_f32 alone on decoder-shaped relations (cfg3: 16000 rows x 6 modes): pl2m-like (124 edges per row,
48000 source rows), t2m-like (27 edges), a2m-like (40 edges).  Prints us per launch (hip events); run under rocprofv3
--pmc for counters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)
K, n_rows = 6, 16000
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2].split(',') if len(sys.argv) > 2 else ['pl2m', 't2m', 'a2m']
shapes = {'pl2m': (124, 4800, True), 't2m': (27, 108800, True), 'a2m': (40, 16000, False)}     # pl2m: the folded polygon table
for name in which:
    deg_mean, n_src, has_eid = shapes[name]
    deg = torch.randint(max(deg_mean // 2, 1), deg_mean * 3 // 2 + 1, (n_rows,), device=dev)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = deg.cumsum(0).int()
    E = int(rowptr[-1])
    if os.environ.get('PATTERN', 'random') == 'local':       # sources near the row's own index, feature rows in CSR order
        rows = torch.repeat_interleave(torch.arange(n_rows, device=dev), deg)
        col = ((rows * n_src // n_rows + torch.randint(0, 256, (E,), device=dev)) % n_src).int()
        eid = torch.arange(E, device=dev).int() if has_eid else None
    else:
        col = torch.randint(0, n_src, (E,), dtype=torch.int32, device=dev)
        eid = torch.randperm(E, device=dev).int() if has_eid else None
    q = torch.randn(n_rows * K, 128, device=dev)
    k = torch.randn(n_src, 128, device=dev)
    v = torch.randn(n_src, 128, device=dev)
    rs = ops.split_rows(ops.row_normalize(torch.randn(E, 128, device=dev)))
    wk, wv = torch.randn(128, 128, device=dev) * 0.1, torch.randn(128, 128, device=dev) * 0.1
    pack, bv = ops.pack_attn_modes(wk, wv), torch.randn(128, device=dev)
    ops.attn_modes(q, k, v, rs, rowptr, col, eid, pack, bv, K, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.attn_modes(q, k, v, rs, rowptr, col, eid, pack, bv, K, 2)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tiles = int(((deg + 15) // 16).sum())
    print('%-5s E %8d tiles %7d  %8.1f us/launch  %.0f cycles per tile per CU (2.4 GHz, 256 CUs)  rhat %.2f TB/s'
          % (name, E, tiles, us, us * 2400 * 256 / tiles, E * 512 / us / 1e6))
