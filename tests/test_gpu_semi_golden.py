"""GPU: the HIP path against the SEMI-PINNED goldens (tests/semi_golden.py: outputs of the reference's own class bodies
executed over stand-ins of the absent third-party primitives).  Edge lists bit-exact, fp32 outputs at the per-element
1e-4 bar of helpers.assert_close — the same bar as the oracle tests, but here the expected values come from the
reference's code, not from this project's restatement of it."""
import pytest
import torch

import semi_golden as S
from helpers import assert_close, rel_err

from tempenslearnagg_qcnet_amd.layers import AttentionLayer
from tempenslearnagg_qcnet_amd.predictors import TempEnsLearnAgg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4
_MODELS = {}


def _model(tag, hp):
    if tag not in _MODELS:
        model = TempEnsLearnAgg(**hp).eval()
        S.seeded_model_weights(model, tag)
        _MODELS[tag] = model.to(DEV)
    return _MODELS[tag]


@pytest.mark.parametrize('case', ['tiny_b3_ts10', 'tiny_b1_ts1', 'tiny_single_ts4', 'cfg1_ts10', 'cfg1_ts1', 'branch_ts6'])
def test_hip_frame_vs_reference_class_bodies(case):
    tag, hp, data, ts = S.case_data(case)
    model = _model(tag, hp)
    with torch.no_grad():
        pred, enc = model(data.to(DEV), ts)
    got = {**enc, **pred}
    n = 0
    for k in S.EDGE_KEYS:
        if k in got and S.has(case, k):
            assert got[k].dtype == torch.int64
            assert torch.equal(got[k].cpu(), S.want(case, k)), (case, k)
            n += 1
    assert n >= 9, n
    if S.has(case, 'x_pt') and 'x_pt' in enc:
        assert_close(enc['x_pt'], S.want(case, 'x_pt'), 'x_pt', atol=5e-6 * float(S.want(case, 'x_pt').abs().max()))
    assert_close(enc['x_pl'][:, 0], S.want(case, 'x_pl'), 'x_pl')
    x_a = S.want(case, 'x_a')
    assert_close(enc['x_a'] if x_a.dim() == 3 else enc['x_a'][:, -1], x_a, 'x_a')
    keys = S.DEC_KEYS if hp['output_head'] else tuple(k for k in S.DEC_KEYS if not k.endswith('_head'))
    for k in keys + ('m',):
        assert_close(pred[k], S.want(case, k), k)
        assert rel_err(pred[k], S.want(case, k)) < TOL, k
    if not hp['output_head']:
        for k in ('loc_propose_head', 'conc_propose_head', 'loc_refine_head', 'conc_refine_head'):
            assert pred[k].shape == S.want(case, k).shape and not bool(pred[k].any()), k


@pytest.mark.parametrize('case', ['tiny_loop', 'cfg1_loop', 'branch_loop'])
def test_hip_ten_frame_pass_vs_reference_class_bodies(case):
    """predict() (batched frames + streaming encoder: the bench path) and the frame-by-frame loop, both against the
    reference's decoder / TemporalAggregate bodies driven by the loop of predictors/temp_ens_learn_agg.py:196-202."""
    tag, hp, data, _ = S.case_data(case)
    model = _model(tag, hp)
    data = data.to(DEV)
    keys = S.DEC_KEYS if hp['output_head'] else tuple(k for k in S.DEC_KEYS if not k.endswith('_head'))
    with torch.no_grad():
        fast = model.predict(data)
        model.batch_frames = False
        try:
            loop = model.predict(data)
        finally:
            model.batch_frames = True
    for got in (fast, loop):
        for k in keys + ('final_m',):
            assert_close(got[k], S.want(case, k), k)
            assert rel_err(got[k], S.want(case, k)) < TOL, k
        for stage in ('propose', 'refine'):                  # predictors/temp_ens_learn_agg.py:204-220, both branches
            cols = [S.want(case, 'loc_%s_pos' % stage), S.want(case, 'scale_%s_pos' % stage)]
            if hp['output_head']:
                cols = [cols[0], S.want(case, 'loc_%s_head' % stage), cols[1], S.want(case, 'conc_%s_head' % stage)]
            want_traj = torch.cat(cols, dim=-1)[..., :50, :]
            assert got['traj_' + stage].shape == want_traj.shape == (want_traj.size(0), 6, 50, 6 if hp['output_head'] else 4)
            assert_close(got['traj_' + stage], want_traj, 'traj_' + stage)
        assert_close(got['pi_softmax'], torch.softmax(S.want(case, 'pi'), dim=-1), 'pi_softmax')


@pytest.mark.parametrize('tag,bipartite,pos', [('bip_pos', True, True), ('self_pos', False, True),
                                               ('self_nopos', False, False), ('bip_long', True, True)])
def test_hip_attention_layer_vs_reference_class_body(tag, bipartite, pos):
    from oracle.seeded_weights import seeded_state_dict, weights_digest
    case = 'layer_' + tag
    g = S.golden()
    layer = AttentionLayer(hidden_dim=128, num_heads=8, head_dim=16, dropout=0.1, bipartite=bipartite, has_pos_emb=pos).eval()
    layer.load_state_dict(seeded_state_dict(layer.state_dict(), int(g['seed_weights']) + 1))
    assert abs(weights_digest(layer.state_dict()) - g[case + '.weights_digest']).max() < 1e-9
    layer = layer.to(DEV)
    x_src = S.want(case, 'x_src').to(DEV)
    x = (x_src, S.want(case, 'x_dst').to(DEV)) if bipartite else x_src
    r = S.want(case, 'r').to(DEV) if pos else None
    with torch.no_grad():
        got = layer(x, r, S.want(case, 'edge_index').to(DEV))
    assert_close(got, S.want(case, 'out'), 'x_a')
    assert rel_err(got, S.want(case, 'out')) < 1e-5
