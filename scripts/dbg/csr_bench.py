"""Isolated cost of ops.csr_build at the sizes a cfg3 step uses (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _bootstrap import ensure_package
ensure_package()
from tempenslearnagg_qcnet_amd import ops
dev = torch.device('cuda:0')
for name, E, n_src, n_dst in (('pl2m', 2040000, 48000, 16000), ('g_t', 900000, 94400, 94400), ('pl2a', 2600000, 283200, 94400),
                              ('pt2pl', 144000, 144000, 4800), ('pl2pl', 300000, 4800, 4800), ('pl2m_agg', 204000, 4800, 1600)):
    g = torch.Generator(device=dev).manual_seed(0)
    src = torch.randint(0, n_src, (E,), device=dev, generator=g)
    dst = torch.randint(0, n_dst, (E,), device=dev, generator=g)
    ei = torch.stack([src.sort().values, dst])
    for _ in range(3):
        ops.csr_build(ei, n_dst)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.csr_build(ei, n_dst)
    b.record(); torch.cuda.synchronize()
    print('%-9s E %8d n_dst %7d: %7.1f us per build' % (name, E, n_dst, a.elapsed_time(b) / 20 * 1e3))
