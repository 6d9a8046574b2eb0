"""Shared by tests/test_semi_golden.py (CPU: oracle) and tests/test_gpu_semi_golden.py (HIP): the cases of
tests/golden/ref_modules_semi.npz — SEMI-PINNED goldens written by oracle/make_semi_golden.py, which executed the
reference's own layers/attention_layer.py and modules/*.py class bodies in the build container over stand-ins of the
absent torch_geometric / torch_cluster (oracle/ref_standins.py: bindings to the oracle's restated primitives).
What a pass means: the reference's index arithmetic, masks, reshapes, transposes, wiring and heads agree with the side
under test; the third-party primitives themselves stay "parity unpinned" (DESIGN.md §2)."""
import numpy as np
import torch

from helpers import load_golden

from oracle.seeded_weights import seeded_state_dict, weights_digest
from tempenslearnagg_qcnet_amd.synthetic import HPARAMS_AV2, SCENE_AV2, config_scene, make_batch, make_scenario

EDGE_KEYS = ('edge_index_pt2pl', 'edge_index_pl2pl', 'edge_index_t', 'edge_index_pl2a', 'edge_index_a2a',
             'edge_index_t2m', 'edge_index_pl2m', 'edge_index_a2m', 'edge_index_m2m', 'edge_index_pl2m_clone')
DEC_KEYS = ('loc_propose_pos', 'scale_propose_pos', 'loc_propose_head', 'conc_propose_head', 'loc_refine_pos',
            'scale_refine_pos', 'loc_refine_head', 'conc_refine_head', 'pi')

_G = None


def golden():
    global _G
    if _G is None:
        _G = load_golden('ref_modules_semi.npz')
    return _G


def relief(batch, seed=5):
    gen = torch.Generator().manual_seed(seed)
    for store in ('map_point', 'map_polygon'):
        z = torch.randn(batch[store]['position'].size(0), generator=gen) * 2.0
        batch[store]['position'][:, 2] = z
        batch[store]['height'] = z.abs() * 0.1
    return batch


def _tiny():
    return config_scene('tiny')


# case -> (weights tag, hyper-parameters, data factory, time shift or None for the ten-frame loop)
def cases():
    tiny = _tiny()
    hp3 = dict(tiny['hparams'], output_head=True, input_dim=3)
    return {
        'tiny_b3_ts10': ('tiny', tiny['hparams'], lambda: make_batch([0, 1, 2], **tiny['scene']), 10),
        'tiny_b1_ts1': ('tiny', tiny['hparams'], lambda: make_batch([5], **tiny['scene']), 1),
        'tiny_single_ts4': ('tiny', tiny['hparams'], lambda: make_scenario(3, **tiny['scene']), 4),
        'tiny_loop': ('tiny', tiny['hparams'], lambda: make_batch([0, 1], **tiny['scene']), None),
        'cfg1_ts10': ('cfg1', HPARAMS_AV2, lambda: make_batch([0], **SCENE_AV2), 10),
        'cfg1_ts1': ('cfg1', HPARAMS_AV2, lambda: make_batch([0], **SCENE_AV2), 1),
        'cfg1_loop': ('cfg1', HPARAMS_AV2, lambda: make_batch([0], **SCENE_AV2), None),
        'branch_ts6': ('branch', hp3, lambda: relief(make_batch([0, 1, 2], **tiny['scene'])), 6),
        'branch_loop': ('branch', hp3, lambda: relief(make_batch([0, 1], **tiny['scene'])), None),
    }


def inputs_digest(data) -> np.ndarray:
    tot = np.zeros(2, dtype=np.float64)
    for key in ('agent', 'map_polygon', 'map_point'):
        for name in sorted(data[key]):
            v = data[key][name]
            if torch.is_tensor(v):
                tot += [float(v.double().sum()), float(v.double().abs().sum())]
    return tot


def seeded_model_weights(module: torch.nn.Module, tag: str):
    """Loads the fixture's weights (named by seed) into ``module`` and returns the CPU state_dict; asserts the digest the
    fixture was made with, i.e. that both sides really hold the same numbers under the same names."""
    g = golden()
    module.load_state_dict(seeded_state_dict(module.state_dict(), int(g['seed_weights'])))
    sd = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    np.testing.assert_allclose(weights_digest(sd), g[tag + '.weights_digest'], rtol=1e-12, atol=1e-9)
    return sd


def case_data(case: str):
    tag, hp, make, ts = cases()[case]
    data = make()
    np.testing.assert_allclose(inputs_digest(data), golden()[case + '.inputs_digest'], rtol=1e-12, atol=1e-9)
    return tag, hp, data, ts


def want(case: str, key: str) -> torch.Tensor:
    v = torch.from_numpy(golden()['%s.%s' % (case, key)])
    return v.long() if v.dtype == torch.int32 else v


def has(case: str, key: str) -> bool:
    return '%s.%s' % (case, key) in golden().files
