"""Radius search at BASELINE.json configs[3] scale (dense preset, batch 4): candidate pairs per second of qcn_radius_count +
scan + qcn_radius_fill for the polygon -> agent search over the 59 absolute steps (n_y = 59 * 4 * 1000 polygon rows x 200
agents each) and the polygon graph (4000 x 1000, neighbour cap 300 hit), checked against the brute-force bound n_y * n_x."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _bootstrap
_bootstrap.ensure_package()
from tempenslearnagg_qcnet_amd import ops

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
B, A, P, S = 4, 200, 1000, 59
out = {}
for name, n_ex, nx, ny, ext_x, ext_y, r, cap, same in (('pl2a_59_steps', B * S, A, P, 100.0, 150.0, 100.0, 300, False),
                                                        ('pl2pl', B, P, P, 150.0, 150.0, 150.0, 301, True)):
    x = ((torch.rand(n_ex * nx, 2, generator=g) * 2 - 1) * ext_x).to(dev)
    y = x if same else ((torch.rand(n_ex * ny, 2, generator=g) * 2 - 1) * ext_y).to(dev)
    px = (torch.arange(n_ex + 1) * nx).to(dev)
    py = (torch.arange(n_ex + 1) * ny).to(dev)
    for _ in range(3):
        ei, _ = ops.radius_pairs(x, y, r, px, py, cap, drop_self=same)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        ei, _ = ops.radius_pairs(x, y, r, px, py, cap, drop_self=same)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    pairs = n_ex * nx * ny
    out[name] = {'candidate_pairs': pairs, 'edges': int(ei.size(1)), 'ms': dt * 1e3, 'gpairs_per_s': pairs / dt / 1e9,
                 'bytes_written': 16 * int(ei.size(1)), 'write_GBps': 16 * ei.size(1) / dt / 1e9}
print(json.dumps(out, indent=1))
