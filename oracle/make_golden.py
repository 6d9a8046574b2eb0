"""Generates tests/golden/ref_*.npz from the reference's own Python sources (build container only).

CPU ORACLE tooling — test infrastructure, never the product.  /root/reference does not exist on the
GPU box, so this script is run here once and its small outputs are committed as data fixtures.

What can be imported from /root/reference (ordinary pure-torch files, loaded by file path):
  utils/weight_init.py, utils/geometry.py, layers/fourier_embedding.py, layers/mlp_layer.py
They do ``from utils import weight_init``; the package ``utils/__init__.py`` would drag in
``utils/graph.py`` → ``torch_geometric`` (not installed), so a namespace module named ``utils`` is
populated with the reference's *own* functions from those two files.  Nothing of torch_geometric /
torch_cluster is imitated: ``layers/attention_layer.py`` and ``modules/*`` therefore stay un-importable
and their rows stay "parity unpinned" (SURVEY.md §8(c)).

Usage:  python oracle/make_golden.py
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def _load(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    wi = _load('_ref_weight_init', os.path.join(REF, 'utils', 'weight_init.py'))
    geo = _load('_ref_geometry', os.path.join(REF, 'utils', 'geometry.py'))
    ns = types.ModuleType('utils')
    ns.weight_init = wi.weight_init
    ns.angle_between_2d_vectors = geo.angle_between_2d_vectors
    ns.wrap_angle = geo.wrap_angle
    saved = sys.modules.get('utils')
    sys.modules['utils'] = ns
    try:
        fe = _load('_ref_fourier_embedding', os.path.join(REF, 'layers', 'fourier_embedding.py'))
        mlp = _load('_ref_mlp_layer', os.path.join(REF, 'layers', 'mlp_layer.py'))
    finally:
        if saved is not None:
            sys.modules['utils'] = saved
        else:
            del sys.modules['utils']
    return wi, geo, fe, mlp


def _perturb(module: torch.nn.Module, gen: torch.Generator) -> None:
    """weight_init leaves every bias at 0 and every LayerNorm at identity, which would hide bias /
    affine bugs — move all of them off their trivial values (deterministically)."""
    with torch.no_grad():
        for name, prm in module.named_parameters():
            if prm.dim() == 1:
                prm.add_(torch.randn(prm.shape, generator=gen) * 0.1)
            elif 'freqs' in name:
                prm.copy_(torch.randn(prm.shape, generator=gen) * 0.05)


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    wi, geo, fe, mlp = load_reference()
    gen = torch.Generator().manual_seed(1234)
    torch.manual_seed(1234)
    blobs = {}

    # ---- geometry (utils/geometry.py:19-23, 47-51)
    ctr = torch.randn(257, 2, generator=gen)
    nbr = torch.randn(257, 2, generator=gen) * 30.0
    nbr[:5] = 0.0                                            # atan2(0, 0) rows
    ang = (torch.rand(513, generator=gen) - 0.5) * 40.0
    ang[:4] = torch.tensor([math.pi, -math.pi, 0.0, 3 * math.pi])
    blobs['geo_ctr'], blobs['geo_nbr'], blobs['geo_angle_in'] = ctr, nbr, ang
    blobs['geo_angle_between'] = geo.angle_between_2d_vectors(ctr, nbr)
    blobs['geo_wrap'] = geo.wrap_angle(ang)

    # ---- FourierEmbedding (layers/fourier_embedding.py) for every input_dim the path uses
    for in_dim, n_rows, n_cat in ((0, 19, 2), (1, 33, 2), (2, 40, 0), (3, 65, 0), (3, 21, 1), (4, 50, 1), (4, 37, 0)):
        m = fe.FourierEmbedding(input_dim=in_dim, hidden_dim=128, num_freq_bands=64).eval()
        _perturb(m, gen)
        tag = 'fe%d_c%d' % (in_dim, n_cat)
        cont = None
        if in_dim:
            cont = torch.randn(n_rows, in_dim, generator=gen) * torch.tensor([40.0, 1.5, 1.5, 5.0][:in_dim])
        cats = [torch.randn(n_rows, 128, generator=gen) * 0.05 for _ in range(n_cat)] or None
        with torch.no_grad():
            out = m(continuous_inputs=cont, categorical_embs=cats)
        for k, v in m.state_dict().items():
            blobs['%s.sd.%s' % (tag, k)] = v
        if cont is not None:
            blobs[tag + '.cont'] = cont
        for i, c in enumerate(cats or []):
            blobs['%s.cat%d' % (tag, i)] = c
        blobs[tag + '.out'] = out

    # ---- MLPLayer (layers/mlp_layer.py) for the decoder head widths (modules/qcnet_decoder.py:117-138)
    for out_dim in (1, 40, 120):
        m = mlp.MLPLayer(input_dim=128, hidden_dim=128, output_dim=out_dim).eval()
        _perturb(m, gen)
        x = torch.randn(23, 6, 128, generator=gen)
        with torch.no_grad():
            y = m(x)
        tag = 'mlp%d' % out_dim
        for k, v in m.state_dict().items():
            blobs['%s.sd.%s' % (tag, k)] = v
        blobs[tag + '.x'], blobs[tag + '.out'] = x, y

    # ---- weight_init statistics (utils/weight_init.py:17-82): bounds / constants only, not the RNG stream
    lin = torch.nn.Linear(128, 384)
    wi.weight_init(lin)
    blobs['wi_linear_absmax'] = lin.weight.detach().abs().max()
    blobs['wi_linear_bias_absmax'] = lin.bias.detach().abs().max()
    gru = torch.nn.GRU(128, 128)
    wi.weight_init(gru)
    whh = gru.weight_hh_l0.detach().chunk(3, 0)
    blobs['wi_gru_hh_orth_err'] = max((w @ w.t() - torch.eye(128)).abs().max() for w in whh)
    blobs['wi_gru_bias_absmax'] = max(gru.bias_ih_l0.detach().abs().max(), gru.bias_hh_l0.detach().abs().max())

    np.savez_compressed(os.path.join(OUT, 'ref_layers.npz'),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in blobs.items()})
    print('wrote %d arrays to %s' % (len(blobs), os.path.join(OUT, 'ref_layers.npz')))


def main_nms() -> None:
    """tests/golden/ref_nms.npz from the reference's own utils/nms.py (pure numpy, imported by file path): clustered
    synthetic candidate sets of every window size 6 .. 60, with the selections the reference makes."""
    ref = _load('_ref_nms', os.path.join(REF, 'utils', 'nms.py'))
    rng = np.random.default_rng(77)
    n_agents, n_cand, n_steps = 24, 60, 50
    trajs = np.empty((n_agents, n_cand, n_steps, 2), dtype=np.float32)
    conf = np.empty((n_agents, n_cand), dtype=np.float32)
    sel = np.empty((n_agents, 10, 6), dtype=np.int64)
    sel_conf = np.empty((n_agents, 10, 6), dtype=np.float32)
    for a in range(n_agents):
        # end points in a few clusters (tight, so suppression and the top-up branch both happen) plus spread ones
        n_clusters = 2 + a % 5
        centres = rng.uniform(-30, 30, size=(n_clusters, 2))
        which = rng.integers(0, n_clusters, size=n_cand)
        spread = np.where(rng.random(n_cand) < (0.2 if a % 3 else 0.0), 8.0, 0.8)[:, None]
        ends = centres[which] + rng.normal(size=(n_cand, 2)) * spread
        ramp = np.linspace(0.02, 1.0, n_steps)[None, :, None]
        trajs[a] = (ends[:, None, :] * ramp + rng.normal(size=(n_cand, n_steps, 2)) * 0.05).astype(np.float32)
        logits = rng.normal(size=(10, 6))
        conf[a] = (np.exp(logits) / np.exp(logits).sum(axis=1, keepdims=True)).reshape(-1).astype(np.float32)
        for w in range(1, 11):
            out = trajs[a, :6 * w].copy()
            clfs = conf[a, :6 * w].copy()
            t_sel, c_sel = ref.nms(out, clfs)
            idx = [int(np.flatnonzero((out == t).all(axis=(1, 2)))[0]) for t in t_sel]
            sel[a, w - 1], sel_conf[a, w - 1] = idx, c_sel
    gt = (trajs[:, 0] + rng.normal(size=(n_agents, n_steps, 2)) * 0.7).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'ref_nms.npz'), trajs=trajs, conf=conf, gt=gt, selected=sel, selected_conf=sel_conf)
    print('wrote ref_nms.npz: %d agents, selections %s' % (n_agents, sel.shape))


def main_geometry_edges() -> None:
    """tests/golden/ref_geometry_edges.npz: the relative-geometry rows [|dp|, angle(heading_dst, dp), wrap(dheading)] of an
    edge list, computed with the reference's own utils/geometry.py functions in the reference's call pattern
    (modules/qcnet_agent_encoder.py:161-166, modules/qcnet_decoder.py:183-188) — the vectors the qcn_rel_geometry_f32
    KERNEL is held to on the GPU (tests/test_gpu_graph.py), incl. zero displacements, equal headings and headings on the
    wrap boundary."""
    geo = _load('_ref_geometry', os.path.join(REF, 'utils', 'geometry.py'))
    gen = torch.Generator().manual_seed(4321)
    ns, nd, e = 400, 150, 6000
    pos_s = torch.randn(ns, 2, generator=gen) * 40.0
    pos_d = torch.randn(nd, 2, generator=gen) * 40.0
    head_s = (torch.rand(ns, generator=gen) * 2 - 1) * math.pi
    head_d = (torch.rand(nd, generator=gen) * 2 - 1) * math.pi
    head_d[:4] = torch.tensor([0.0, math.pi, -math.pi, math.pi / 2])
    ei = torch.stack([torch.randint(0, ns, (e,), generator=gen), torch.randint(0, nd, (e,), generator=gen)])
    ei[0, :40] = torch.arange(40)
    pos_s[ei[0, :40]] = pos_d[ei[1, :40]]                      # zero displacement
    head_s[ei[0, 40:70]] = head_d[ei[1, 40:70]]                # zero heading difference
    src, dst = ei[0], ei[1]
    head_vec_d = torch.stack([head_d.cos(), head_d.sin()], dim=-1)
    rel_pos = pos_s[src] - pos_d[dst]
    rows = torch.stack([torch.norm(rel_pos[:, :2], p=2, dim=-1),
                        geo.angle_between_2d_vectors(ctr_vector=head_vec_d[dst], nbr_vector=rel_pos[:, :2]),
                        geo.wrap_angle(head_s[src] - head_d[dst])], dim=-1)
    path = os.path.join(OUT, 'ref_geometry_edges.npz')
    np.savez_compressed(path, pos_s=pos_s.numpy(), pos_d=pos_d.numpy(), head_s=head_s.numpy(), head_d=head_d.numpy(),
                        edge_index=ei.numpy(), rows=rows.numpy())
    print('wrote', path)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'geometry_edges':
        main_geometry_edges()
    else:
        main()
        main_nms()
        main_geometry_edges()
