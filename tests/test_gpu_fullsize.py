"""GPU parity at BASELINE.json's FULL sizes (configs 2 / 3 / 4 were only ever checked at reduced size).

* cfg2 (QCNet encoder + decoder, batch 32): the CPU oracle is affordable for ONE frame of the whole batch — edge lists
  bit-exact, outputs to the per-element bar.
* cfg3 (ten frames + TemporalAggregate, batch 32): the oracle on one full-size scenario (M5 / P2 at full size), plus the
  size-independent properties on the whole batch: every scenario of the batch equals its prediction alone, the batched ten
  frames equal the frame loop.
* cfg4 (dense: 200 agents x 1000 polygons, batch 4) in BOTH arithmetic modes it is benched in: scenario independence,
  batched = frame loop, finiteness; one frame of one dense scenario against the oracle (edge lists bit-exact incl. the
  neighbour-cap truncation of the polygon graph, outputs to the mode's bar).
"""
import time

import pytest
import torch

from helpers import assert_close, cpu_state_dict, oracle, randomize_, rel_err

from tempenslearnagg_qcnet_amd import ops
from tempenslearnagg_qcnet_amd.predictors import TempEnsLearnAgg
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch

pytestmark = pytest.mark.gpu
BF16_TOL_MODEL = 3e-2   # as tests/test_gpu_bf16.py
DEV = 'cuda:0'

EDGE_KEYS_ENC = ('edge_index_pl2pl', 'edge_index_t', 'edge_index_pl2a', 'edge_index_a2a')
EDGE_KEYS_DEC = ('edge_index_t2m', 'edge_index_pl2m', 'edge_index_a2m', 'edge_index_m2m', 'edge_index_pl2m_clone')
OUT_KEYS = ('loc_propose_pos', 'scale_propose_pos', 'loc_refine_pos', 'scale_refine_pos', 'pi')
PRED_KEYS = ('traj_refine', 'pi_softmax', 'final_m')


def _model(hparams, seed=0):
    torch.manual_seed(seed)
    model = randomize_(TempEnsLearnAgg(**hparams), seed=seed + 1).eval()
    randomize_(model.temporal_aggregate_layer, seed=seed + 2)
    return model


@pytest.fixture
def restore_matmul_mode():
    mode = ops.matmul_mode()
    yield
    ops.set_matmul_mode(mode)


def test_cfg2_batch32_one_frame_vs_oracle():
    """BASELINE.json configs[1] at its stated size: 32 AV2-shaped scenarios through encoder + decoder."""
    cfg = config_scene('cfg2')
    model = _model(cfg['hparams'])
    sd = cpu_state_dict(model)
    batch = make_batch(list(range(32)), **cfg['scene'])
    t0 = time.time()
    with torch.no_grad():
        want_pred, want_enc = oracle.qcnet_forward(sd, batch, 10, cfg['hparams'], p='base_model.')
    print('oracle, one frame of 32 scenarios: %.1f s' % (time.time() - t0))
    with torch.no_grad():
        got_pred, got_enc = model.to(DEV)(batch.to(DEV), 10)
    for k in EDGE_KEYS_ENC:
        assert torch.equal(got_enc[k].cpu(), want_enc[k]), k
    for k in EDGE_KEYS_DEC:
        assert torch.equal(got_pred[k].cpu(), want_pred[k]), k
    assert_close(got_enc['x_a'], want_enc['x_a'], 'x_a')
    for k in OUT_KEYS + ('m',):
        worst = assert_close(got_pred[k], want_pred[k], k)
        print('cfg2 %s: worst element at %.2f of the bar, max-norm rel err %.2e' % (k, worst, rel_err(got_pred[k], want_pred[k])))


def test_cfg3_one_full_size_scenario_vs_oracle():
    """M5 / P2 at full size: ten frames + TemporalAggregate of one 50-agent / 150-polygon scenario against the oracle."""
    cfg = config_scene('cfg3')
    model = _model(cfg['hparams'], seed=4)
    sd = cpu_state_dict(model)
    batch = make_batch([17], **cfg['scene'])
    with torch.no_grad():
        want = oracle.temp_ens_learn_agg_forward(sd, batch, cfg['hparams'])
        got = model.to(DEV).predict(batch.to(DEV))
    for k in ('traj_refine', 'pi_softmax', 'final_m', 'loc_refine_pos', 'scale_refine_pos', 'pi'):
        worst = assert_close(got[k], want[k], k)
        print('cfg3 scenario %s: worst element at %.2f of the bar, max-norm rel err %.2e' % (k, worst, rel_err(got[k], want[k])))


def _properties(model, cfg, ids, probe, tol):
    """predict(batch)[scenario] == predict([scenario]) and batched frames == frame loop, all finite."""
    A = cfg['scene']['num_agents']
    batch = make_batch(ids, **cfg['scene']).to(DEV)
    with torch.no_grad():
        full = model.predict(batch)
        for k in PRED_KEYS:
            assert torch.isfinite(full[k]).all(), k
        for i in probe:
            alone = model.predict(make_batch([ids[i]], **cfg['scene']).to(DEV))
            for k in PRED_KEYS:
                assert rel_err(full[k][i * A:(i + 1) * A], alone[k]) < tol, (k, i)
        model.batch_frames = False
        try:
            loop = model.predict(batch)
        finally:
            model.batch_frames = True
    for k in PRED_KEYS:
        assert rel_err(full[k], loop[k]) < tol, k
    return full


def test_cfg3_batch32_properties():
    """BASELINE.json configs[2] at its stated size (the bench workload)."""
    cfg = config_scene('cfg3')
    model = _model(cfg['hparams'], seed=4).to(DEV)
    full = _properties(model, cfg, list(range(32)), probe=(0, 17, 31), tol=1e-5)
    assert full['traj_refine'].shape == (32 * 50, 6, 50, 4)
    # scenario 17 of the batch is the scenario test_cfg3_one_full_size_scenario_vs_oracle checks against the oracle


@pytest.mark.parametrize('mode', [ops.MATMUL_SPLIT_F16, ops.MATMUL_BF16])
def test_cfg4_dense_batch4_properties(mode, restore_matmul_mode):
    """BASELINE.json configs[3] at its stated size — 4 scenarios x 200 agents x 1000 polygons, radii 100 / 150 m — in the
    fp32-equivalent mode and in the bf16 mode the config names.  In the bf16 mode a scenario alone and the same scenario
    inside a batch run the same kernels on the same rows, so independence holds to fp32 rounding there as well."""
    ops.set_matmul_mode(mode)
    cfg = config_scene('cfg4')
    model = _model(cfg['hparams'], seed=6).to(DEV)
    full = _properties(model, cfg, list(range(4)), probe=(1, 3), tol=1e-5 if mode != ops.MATMUL_BF16 else 2e-3)
    assert full['traj_refine'].shape == (4 * 200, 6, 50, 4)


@pytest.mark.parametrize('mode', [ops.MATMUL_SPLIT_F16, ops.MATMUL_BF16])
def test_cfg4_one_dense_scenario_one_frame_vs_oracle(mode, restore_matmul_mode):
    """One frame of one full-size dense scenario (200 agents, 1000 polygons: 0.5 M polygon pairs inside the radius before the
    neighbour cap, ~3 M polygon -> agent edges) against the oracle in the same arithmetic."""
    from contextlib import nullcontext
    ops.set_matmul_mode(mode)
    cfg = config_scene('cfg4')
    model = _model(cfg['hparams'], seed=6)
    sd = cpu_state_dict(model)
    batch = make_batch([2], **cfg['scene'])
    t0 = time.time()
    with torch.no_grad(), (oracle.bf16_operands() if mode == ops.MATMUL_BF16 else nullcontext()):
        want_pred, want_enc = oracle.qcnet_forward(sd, batch, 10, cfg['hparams'], p='base_model.')
    print('oracle, one frame of one dense scenario: %.1f s' % (time.time() - t0))
    with torch.no_grad():
        got_pred, got_enc = model.to(DEV)(batch.to(DEV), 10)
    per_target = torch.bincount(want_enc['edge_index_pl2pl'][1], minlength=1000)
    assert int(per_target.max()) >= 300                          # the cap of radius_graph is hit at this density
    for k in EDGE_KEYS_ENC:
        assert torch.equal(got_enc[k].cpu(), want_enc[k]), k
    for k in EDGE_KEYS_DEC:
        assert torch.equal(got_pred[k].cpu(), want_pred[k]), k
    fp32_pred = None
    if mode == ops.MATMUL_BF16:                                   # the mode's own cost at this size: bf16 oracle vs fp32 oracle
        with torch.no_grad():
            fp32_pred, fp32_enc = oracle.qcnet_forward(sd, batch, 10, cfg['hparams'], p='base_model.')
        # Three-way guard (a max-norm bar alone would pass a path that ignores the mode), applied where
        # it discriminates.  SHALLOW outputs (x_pl: 2 layers, x_a: 6 layers behind the inputs): the HIP result must be
        # CLOSER to the bf16-operand oracle than the fp32 oracle is, as in tests/test_gpu_bf16.py:173-177.
        for k, g, w, f in (('x_pl', got_enc['x_pl'][:, 0], want_enc['x_pl'][:, 0], fp32_enc['x_pl'][:, 0]),
                           ('x_a', got_enc['x_a'], want_enc['x_a'], fp32_enc['x_a'])):
            err, far, ofar = rel_err(g, w), rel_err(g, f), rel_err(w, f)
            print('cfg4 bf16 %s: HIP vs bf16 oracle %.2e | HIP vs fp32 oracle %.2e | bf16 oracle vs fp32 oracle %.2e' % (k, err, far, ofar))
            assert err < BF16_TOL_MODEL and err < ofar, k
            assert far < 3 * max(ofar, 1e-3) + err, k
    for k in OUT_KEYS + ('m',):
        if mode == ops.MATMUL_BF16:
            err, far, ofar = rel_err(got_pred[k], want_pred[k]), rel_err(got_pred[k], fp32_pred[k]), rel_err(want_pred[k], fp32_pred[k])
            print('cfg4 bf16 %s: HIP vs bf16 oracle %.2e | HIP vs fp32 oracle %.2e | bf16 oracle vs fp32 oracle %.2e' % (k, err, far, ofar))
            # DEEP outputs (28 more layers, softmax rows of up to 1000 polygon edges): measured on this scene, HIP, the bf16
            # oracle and the fp32 oracle are MUTUALLY 1.1e-2 .. 1.5e-2 apart (loc 1.07e-2 / 1.11e-2 / 1.28e-2, scale_refine
            # 1.53e-2 / 1.50e-2 / 1.40e-2) — HIP and oracle round activations that already differ by fp32 summation order, a
            # value on a bf16 rounding boundary lands on different sides, and at this depth the two bf16 realisations have
            # decorrelated: "closer to the bf16 oracle than fp32 is" no longer holds or fails for a reason (that is why the
            # round-2 bar of 3e-2 broke at 3.07e-2 for `m` after an unrelated kernel change).  What the full-size run CAN tell
            # at this depth: the result is AS FAR from the fp32 oracle as the mode itself (a path that ignored the mode
            # would sit at far ~ 1e-6), not further (far <= 3 ofar + err), and within 5e-2 of the bf16 oracle.  The
            # arithmetic contract itself is held at 1e-5 by the kernel-level tests (tests/test_gpu_bf16.py).
            assert err < 5e-2, k
            assert far > 0.3 * ofar, k
            assert far < 3 * max(ofar, 1e-3) + err, k
        else:
            worst = assert_close(got_pred[k], want_pred[k], k)
            print('cfg4 %s: worst element at %.2f of the bar' % (k, worst))
