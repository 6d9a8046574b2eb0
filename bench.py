"""Benchmark of the TempEnsLearnAgg hot path on MI355X (contract: see the round prompt / DESIGN.md "Measurement").

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one full TempEnsLearnAgg inference pass (ten sliding QCNet frames + TemporalAggregate, edge construction
included) over one batch of 32 synthetic AV2-shaped scenarios already resident in HBM (BASELINE.json configs[2]).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import _bootstrap

_bootstrap.ensure_package()

from tempenslearnagg_qcnet_amd import ops, sharding  # noqa: E402
from tempenslearnagg_qcnet_amd.predictors import TempEnsLearnAgg  # noqa: E402
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def build_model(hparams, device):
    torch.manual_seed(0)
    model = TempEnsLearnAgg(**hparams).eval()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for prm in model.parameters():          # random-init weights; move biases / LayerNorms off identity
            if prm.dim() == 1:
                prm.add_(torch.randn(prm.shape, generator=gen) * 0.1)
    return model.to(device)


def cpu_baseline(hparams, scene, state_dict, sample_scenarios: int):
    """The oracle (PyG-formulation restatement, oracle/qcnet_oracle.py) timed on the host cores: kind 'port'."""
    from oracle import qcnet_oracle as oracle
    threads = min(os.cpu_count() or 1, int(os.environ.get('OMP_NUM_THREADS', '16')))
    torch.set_num_threads(threads)
    batch = make_batch(list(range(sample_scenarios)), **scene)
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        oracle.temp_ens_learn_agg_forward(sd, batch, hparams)
    dt = time.perf_counter() - t0
    return {'value': sample_scenarios / dt, 'unit': 'scenarios/sec', 'cores': threads, 'kind': 'port',
            'sample': '%d scenario(s) of the same workload, full 10-frame loop + aggregate, %.1f s' % (sample_scenarios, dt)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='cfg3', choices=['cfg3', 'cfg4', 'tiny'])
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-isolated-pass', action='store_true',
                    help='skip the extra attention timing with the preparation stream off (roofline.isolated)')
    ap.add_argument('--no-exact-pass', action='store_true', help='skip the extra timing in the exact-fp32-MFMA mode')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    cfg = config_scene(args.config)
    batch_size = args.batch or cfg['batch']
    model = build_model(cfg['hparams'], device)
    # weak scaling: every rank owns its own batch of scenarios (ids disjoint across ranks)
    ids = [rank * batch_size + i for i in range(batch_size)]
    data = make_batch(ids, **cfg['scene']).to(device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The path runs on a high-priority stream: the graph preparation of the next batch (utils/prep.py: a normal-priority
    # side stream) overlaps this batch's attention / MLP kernels and should only take the CUs they leave idle.
    main_stream = torch.cuda.Stream(device=device, priority=-1) if os.environ.get('QCN_MAIN_PRIORITY', '1') != '0' \
        else torch.cuda.current_stream(device)
    main_stream.wait_stream(torch.cuda.current_stream(device))
    with torch.no_grad(), torch.cuda.stream(main_stream):
        data.mark_ready()
        for _ in range(args.warmup):   # same work as a timed step, so first-use kernel loads stay out of the timing
            pred = model.predict(data)
            sharding.gather_records(sharding.focal_records(ids, data, pred), world)
            sharding.reduce_metric_sums(sharding.validation_metric_sums(data, pred), world)
        barrier()
        ops.prof_enable(True, kinds=[ops.PROF_ATTN_FUSED])
        t0 = time.perf_counter()
        records = []
        for _ in range(args.steps):
            pred = model.predict(data)
            records.append(sharding.focal_records(ids, data, pred))
        # the path's only exchange: gather the per-scenario prediction records and the metric sums
        gathered = sharding.gather_records(torch.cat(records), world)
        sums = sharding.reduce_metric_sums(sharding.validation_metric_sums(data, pred), world)
        barrier()
        elapsed = time.perf_counter() - t0
        ops.prof_enable(False)
    launches, attn_ms, attn_bytes = ops.prof_collect(ops.PROF_ATTN_FUSED)
    ops.prof_collect(-1)

    # The same launches with the preparation stream switched off (everything inline on one stream): in the timed region
    # above the attention kernels share the CUs with the next batch's preparation kernels, which lengthens their
    # individual durations while shortening the step; this pass times them alone on the GPU (kernel quality).
    from tempenslearnagg_qcnet_amd.utils import prep as _prep
    iso = None
    if _prep.ENABLED and not args.no_isolated_pass:
        _prep.ENABLED = False
        with torch.no_grad(), torch.cuda.stream(main_stream):
            model.predict(data)
            torch.cuda.synchronize()
            ops.prof_enable(True, kinds=[ops.PROF_ATTN_FUSED])
            for _ in range(min(args.steps, 3)):
                model.predict(data)
            torch.cuda.synchronize()
            ops.prof_enable(False)
        iso = ops.prof_collect(ops.PROF_ATTN_FUSED)
        ops.prof_collect(-1)
        _prep.ENABLED = True

    # The same steps with the chain kernels in their exact-fp32-MFMA mode (include/qcnet_hip.h: QCN_MATMUL_EXACT_F32),
    # reported beside `value` (which is measured in the default split-f16 mode, whose matmul error is not larger).
    exact_ms = None
    if world == 1 and not args.no_exact_pass:
        mode = ops.matmul_mode()
        ops.set_matmul_mode(ops.MATMUL_EXACT_F32)
        with torch.no_grad():
            model.predict(data)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pred_exact = model.predict(data)
                sharding.focal_records(ids, data, pred_exact)
            torch.cuda.synchronize()
            exact_ms = (time.perf_counter() - t1) / args.steps * 1e3
        ops.set_matmul_mode(mode)

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # HBM bytes per launch of the roofline kernel from the rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, calibrated
    # on a known byte count in the kernel's own access width: scripts/pmc_workload.py, scripts/pmc_summary.py); the
    # counters cannot be read from inside this process, so the committed summary of the same workload is quoted.
    traffic = None
    pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r1_x_pmc_attention.json')
    if args.config == 'cfg3' and batch_size == cfg['batch'] and os.path.exists(pmc_file):
        with open(pmc_file) as fh:
            traffic = json.load(fh)['attn_fused']['traffic_bytes_per_launch']

    if rank == 0:
        total = world * batch_size * args.steps
        achieved = attn_bytes / (attn_ms * 1e-3) / 1e9 if attn_ms > 0 else 0.0
        out = {
            'metric': 'scenarios/sec (TempEnsLearnAgg fwd, AV2-shape)', 'value': total / elapsed,
            'unit': 'scenarios/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: TempEnsLearnAgg (10 sliding QCNet frames + TemporalAggregate), batch %d '
                                   'synthetic AV2-shaped scenarios per GPU (%d agents x 110 steps, %d polygons), random-init '
                                   'fp32 weights' % (args.config, batch_size, cfg['scene']['num_agents'],
                                                     cfg['scene']['num_polygons']),
                       'global_batch': world * batch_size, 'parallelism': 'scenario-sharded x%d' % world,
                       'matmul': 'split-f16 x3 MFMA, 22-bit operands, fp32 accumulate (backward error 8.7e-7 vs 1.4e-6 '
                                 'of the fp32 fma chain); attention / softmax / LayerNorm / geometry in fp32'},
            'roofline': {'bound': 'hbm', 'kernel': 'attn_modes_kernel + attn_fused_kernel (qcn_attn_modes_f32, qcn_attn_fused_f32: every positional attention launch)', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'launches': launches, 'avg_launch_us': attn_ms * 1e3 / max(launches, 1),
                         'alg_bytes_per_launch': attn_bytes / max(launches, 1),
                         'formula': '1552*E + 1024*n_dst (SURVEY 8(d), r-projection fused) with E = the edges a launch processes: a mode-replicated '
                                    'relation is processed once per base edge for all K modes; traffic = HBM bytes per launch (PMC); '
                                    'isolated = the same launches timed with the preparation stream off (in the timed region they share '
                                    'the CUs with the next batch\'s preparation kernels)', 'traffic': traffic,
                         'isolated': None if iso is None or iso[1] <= 0 else
                         {'achieved': iso[2] / (iso[1] * 1e-3) / 1e9, 'frac': iso[2] / (iso[1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          'launches': iso[0], 'avg_launch_us': iso[1] * 1e3 / max(iso[0], 1)}},
            'exact_f32_mfma': None if exact_ms is None else
            {'ms_per_step': exact_ms, 'value': batch_size / (exact_ms * 1e-3), 'unit': 'scenarios/sec',
             'max_rel_diff_loc_refine_pos': float((pred_exact['loc_refine_pos'] - pred['loc_refine_pos']).abs().max()
                                                  / pred['loc_refine_pos'].abs().max())},
            'gathered_records': int(gathered.size(0)), 'metric_sums': dict(zip(sharding.VALIDATION_METRICS, [[float(a), float(b)] for a, b in sums.view(-1, 2).tolist()])),
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg['hparams'], cfg['scene'], model.state_dict(),
                                               1 if args.config != 'tiny' else 2)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
