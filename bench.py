"""Benchmark of the TempEnsLearnAgg hot path on MI355X (contract: see the round prompt / DESIGN.md "Measurement").

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one full TempEnsLearnAgg inference pass (ten sliding QCNet frames + TemporalAggregate, edge construction
included) over one batch of 32 synthetic AV2-shaped scenarios already resident in HBM (BASELINE.json configs[2]); every rank
owns its own scenarios and every step of a run sees a different batch (scenario ids disjoint across ranks and steps —
BASELINE.json configs[4]'s "batch 32 per step", sharded by scenario).  ``--config cfg5 --scenarios S`` runs a fixed pool of S
scenarios through ``sharding.run_sharded`` instead (strong scaling; the batches are built by loader worker processes — the
reference's DataLoader(num_workers, pin_memory, persistent_workers), evaluate.py:42-43 — and the host-to-device copies sit
inside the timed region, as they would for a dataset).  ``--config cfg4`` is the dense preset in the bf16 mode it names.
Rank 0 prints ONE JSON line.
"""
import argparse
import functools
import json
import os
import statistics
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
from tempenslearnagg_qcnet_amd.synthetic import config_scene, make_batch, make_scenario  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8 TB/s
# MFMA ceilings of the three arithmetic modes, TFLOP/s of fp32-equivalent work (MI355X_MICROARCH.md dense peaks: fp32 matrix
# 157, f16 / bf16 2517; the split mode spends three f16 MFMAs per product)
MFMA_PEAK_TFLOPS = {ops.MATMUL_EXACT_F32: 157.0, ops.MATMUL_SPLIT_F16: 2517.0 / 3.0, ops.MATMUL_BF16: 2517.0}
MODES = {'exact': ops.MATMUL_EXACT_F32, 'split': ops.MATMUL_SPLIT_F16, 'bf16': ops.MATMUL_BF16}
DTYPE_LABEL = {ops.MATMUL_EXACT_F32: 'f32', ops.MATMUL_SPLIT_F16: 'f32 (split-f16x3)', ops.MATMUL_BF16: 'bf16'}
MATMUL_NOTE = {
    ops.MATMUL_EXACT_F32: 'exact fp32 MFMA (v_mfma_f32_32x32x2_f32)',
    ops.MATMUL_SPLIT_F16: 'split-f16 x3 MFMA, 22-bit operands, fp32 accumulate (backward error 8.7e-7 vs 1.4e-6 of the fp32 fma '
                          'chain); attention / softmax / LayerNorm / geometry in fp32',
    ops.MATMUL_BF16: 'bf16 operands (RNE), fp32 accumulate, in every Linear of the path incl. the per-edge positional '
                     'projections of the decoder relations; GRU recurrence, softmax, LayerNorm, geometry in fp32'}


def build_model(hparams, device):
    torch.manual_seed(0)
    model = TempEnsLearnAgg(**hparams).eval()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for prm in model.parameters():          # random-init weights; move biases / LayerNorms off identity
            if prm.dim() == 1:
                prm.add_(torch.randn(prm.shape, generator=gen) * 0.1)
    return model.to(device)


def cpu_baseline(hparams, scene, state_dict, n_many: int, bf16: bool, one_frame_only: bool = False):
    """The oracle (PyG-formulation restatement, oracle/qcnet_oracle.py) timed on the host cores: kind 'port'.  Two figures on
    bounded samples of the same workload: all cores the box gives this job on a small batch, and one thread on one scenario.
    ``one_frame_only`` (dense preset: a full pass of ONE scenario takes minutes on the host): one of the ten frames of one
    scenario is timed and the pass is priced at 10.8 frames (ten frames + the aggregate, which is a decoder pass)."""
    from contextlib import nullcontext
    from oracle import qcnet_oracle as oracle
    threads = min(os.cpu_count() or 1, int(os.environ.get('OMP_NUM_THREADS', '16')))
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}

    def run(n_threads, n_scen):
        torch.set_num_threads(n_threads)
        batch = make_batch(list(range(n_scen)), **scene)
        t0 = time.perf_counter()
        with torch.no_grad(), (oracle.bf16_operands() if bf16 else nullcontext()):
            if one_frame_only:
                oracle.qcnet_forward(sd, batch, 10, hparams, p='base_model.')
            else:
                oracle.temp_ens_learn_agg_forward(sd, batch, hparams)
        return (time.perf_counter() - t0) * (10.8 if one_frame_only else 1.0)

    what = 'ONE of its ten frames timed, pass priced at 10.8 frames' if one_frame_only else 'full 10-frame loop + aggregate'
    dt_many = run(threads, n_many)
    out = {'value': n_many / dt_many, 'unit': 'scenarios/sec', 'cores': threads, 'kind': 'port',
           'sample': 'one batch of %d scenario(s) of the same workload, %s, %.1f s on %d threads' % (n_many, what, dt_many, threads)}
    if not one_frame_only:
        dt_one = run(1, 1)
        out['single_thread'] = {'value': 1.0 / dt_one, 'unit': 'scenarios/sec', 'cores': 1,
                                'sample': '1 scenario, %s, %.1f s' % (what, dt_one)}
    torch.set_num_threads(threads)
    return out


def stream_copy_peak(device, n_bytes: int = 1 << 30, reps: int = 10) -> float:
    """Achievable HBM bandwidth of this box: GB/s (read + write) of the library's float4 streaming copy
    (qcn_stream_copy, csrc/api.hip) on a 1 GiB buffer — the guide's microbenchmark, not a library memcpy on bytes
    (``torch.copy_`` on uint8 read 20 % low and flattered ``frac_of_achievable``)."""
    from tempenslearnagg_qcnet_amd import _lib
    x = torch.zeros(n_bytes // 4, dtype=torch.float32, device=device)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream(device).cuda_stream
    copy = lambda: _lib.check(_lib.lib().qcn_stream_copy(x.data_ptr(), y.data_ptr(), n_bytes, stream), 'qcn_stream_copy')
    copy()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        copy()
    stop.record()
    stop.synchronize()
    return 2.0 * n_bytes * reps / (start.elapsed_time(stop) * 1e-3) / 1e9


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='cfg3', choices=['cfg3', 'cfg4', 'cfg5', 'tiny'])
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--scenarios', type=int, default=None, help='cfg5: size of the scenario pool (default 10000)')
    ap.add_argument('--workers', type=int, default=6, help='cfg5: loader worker processes per rank (0: synchronous)')
    ap.add_argument('--matmul', default=None, choices=sorted(MODES), help='default: bf16 for cfg4 (as BASELINE.json names it), '
                    'split (fp32-equivalent) otherwise')
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
    mode = MODES[args.matmul] if args.matmul else (ops.MATMUL_BF16 if args.config == 'cfg4' else ops.MATMUL_SPLIT_F16)
    ops.set_matmul_mode(mode)
    model = build_model(cfg['hparams'], device)
    pooled = args.config == 'cfg5'

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The path runs on a high-priority stream: the graph preparation of the next batch (utils/prep.py: a normal-priority
    # side stream) overlaps this batch's attention / MLP kernels and should only take the CUs they leave idle.
    main_stream = torch.cuda.Stream(device=device, priority=-1) if os.environ.get('QCN_MAIN_PRIORITY', '1') != '0' \
        else torch.cuda.current_stream(device)
    main_stream.wait_stream(torch.cuda.current_stream(device))

    # Weak scaling: rank r, step s owns scenario ids (s * world + r) * batch .. + batch (disjoint across ranks and steps);
    # the batches of up to 8 different steps are resident in HBM before the timed region and cycled.
    n_pool = max(1, min(args.steps, 8, args.warmup if args.warmup > 0 else 8))   # every pooled batch is seen by a warm-up step
    step_ids = [[(s * world + rank) * batch_size + i for i in range(batch_size)] for s in range(n_pool)]
    batches = [make_batch(ids, **cfg['scene']).to(device) for ids in step_ids]
    step_ms = []
    host_loop = None
    with torch.no_grad(), torch.cuda.stream(main_stream):
        for b in batches:
            b.mark_ready()
        for w in range(args.warmup):   # same work as a timed step, so first-use kernel loads stay out of the timing
            ids, data = step_ids[w % n_pool], batches[w % n_pool]
            pred = model.predict(data)
            sharding.gather_records(sharding.focal_records(ids, data, pred), world)
            sharding.reduce_metric_sums(sharding.validation_metric_sums(data, pred), world)
        barrier()
        # timed region: only the roofline kernels (the positional attention launches) are bracketed by events — each recorded
        # launch costs two event records, and bracketing the ~300 chain launches of a step as well costs 2.5 % of `value`
        ops.prof_enable(True, kinds=[ops.PROF_ATTN_FUSED])
        if pooled:
            n_scen = args.scenarios or 10000
            # persistent workers (evaluate.py:43): forked and building their first batches before the timed region, as the
            # resident batches of the other configs are made before it (at most workers x 2 batches are ready at t0)
            loader = (sharding.start_loader(functools.partial(make_scenario, **cfg['scene']), n_scen, batch_size, rank, world,
                                            args.workers) if args.workers > 0 else None)
            barrier()
            t0 = time.perf_counter()
            # batches are built by worker processes ahead of the GPU and arrive in pinned memory (datasets/loader.py: the
            # reference's DataLoader(num_workers, pin_memory), evaluate.py:42-43); --workers 0 builds them inside the loop
            gathered, sums = sharding.run_sharded(model.predict, lambda ids: make_batch(ids, **cfg['scene']), n_scen, batch_size,
                                                  rank, world, device, metric_fn=sharding.validation_metric_sums,
                                                  loader=loader)
            barrier()
            elapsed = time.perf_counter() - t0
            steps = -(-len(sharding.shard_scenario_ids(n_scen, 0, world)) // batch_size)
            total = n_scen
            pred = None
        else:
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            t0 = time.perf_counter()
            records, sums = [], None
            marks[0].record(main_stream)
            for s in range(args.steps):
                ids, data = step_ids[s % n_pool], batches[s % n_pool]
                pred = model.predict(data)
                records.append(sharding.focal_records(ids, data, pred))
                part = sharding.validation_metric_sums(data, pred)
                sums = part if sums is None else sums + part
                marks[s + 1].record(main_stream)
            host_loop = time.perf_counter() - t0      # the host has enqueued every timed step (it runs ahead of the GPU)
            # the path's only exchange: gather the per-scenario prediction records and the metric sums
            gathered = sharding.gather_records(torch.cat(records), world)
            sums = sharding.reduce_metric_sums(sums, world)
            barrier()
            elapsed = time.perf_counter() - t0
            steps = args.steps
            total = world * batch_size * args.steps
            step_ms = [marks[s].elapsed_time(marks[s + 1]) for s in range(args.steps)]
        ops.prof_enable(False)
    # the same records per step (the host and the preparation stream run ahead of the main stream, so the last steps of a
    # short run have the GPU to themselves: the per-step fractions show both regimes)
    frac_by_step = []
    attn_log = [r for r in ops._PROF_LOG if r[0] == ops.PROF_ATTN_FUSED]
    if not pooled and attn_log and len(attn_log) % args.steps == 0:
        per = len(attn_log) // args.steps
        for s_ in range(args.steps):
            recs = attn_log[s_ * per:(s_ + 1) * per]
            ms_ = sum(r[2].elapsed_time(r[3]) for r in recs)
            frac_by_step.append(round(sum(r[1] for r in recs) / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) if ms_ > 0 else None)
    launches, attn_ms, attn_bytes = ops.prof_collect(ops.PROF_ATTN_FUSED)
    ops.prof_collect(-1)
    # the MFMA chain kernels of the same steps (same regime: preparation stream on, batches cycled), bracketed in a pass of
    # their own right after the timed region
    n_chain = 0 if pooled else min(args.steps, 5)
    with torch.no_grad(), torch.cuda.stream(main_stream):
        ops.prof_enable(True, kinds=[ops.PROF_FOURIER, ops.PROF_GEMM])
        for s_ in range(n_chain):
            model.predict(batches[s_ % n_pool])
        torch.cuda.synchronize()
        ops.prof_enable(False)
    f_launches, f_ms, f_flop = ops.prof_collect(ops.PROF_FOURIER)
    g_launches, g_ms, g_flop = ops.prof_collect(ops.PROF_GEMM)
    ops.prof_collect(-1)
    if pred is not None:    # the bench asserts what it measured: finite predictions, probabilities that sum to one
        assert bool(torch.isfinite(pred['traj_refine']).all()) and bool(torch.isfinite(pred['pi_softmax']).all())
        assert float((pred['pi_softmax'].sum(-1) - 1).abs().max()) < 1e-4
    assert gathered.size(0) == (total if not pooled else args.scenarios or 10000), 'a scenario record went missing'

    # The same launches with the preparation stream switched off (everything inline on one stream): in the timed region
    # above the attention kernels share the CUs with the next batch's preparation kernels, which lengthens their
    # individual durations while shortening the step; this pass times them alone on the GPU (kernel quality).
    from tempenslearnagg_qcnet_amd.utils import prep as _prep
    iso = iso_f = iso_g = None
    data, ids = batches[0], step_ids[0]
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
        # ... and the MFMA chain kernels the same way (alone on the GPU, one stream)
        with torch.no_grad(), torch.cuda.stream(main_stream):
            ops.prof_enable(True, kinds=[ops.PROF_FOURIER, ops.PROF_GEMM])
            for _ in range(min(args.steps, 3)):
                model.predict(data)
            torch.cuda.synchronize()
            ops.prof_enable(False)
        iso_f, iso_g = ops.prof_collect(ops.PROF_FOURIER), ops.prof_collect(ops.PROF_GEMM)
        ops.prof_collect(-1)
        _prep.ENABLED = True

    # The same steps with the chain kernels in their exact-fp32-MFMA mode (include/qcnet_hip.h: QCN_MATMUL_EXACT_F32),
    # reported beside `value` (split mode: its matmul error is not larger; bf16 mode: the distance is the mode's cost).
    exact = None
    if world == 1 and not args.no_exact_pass and not pooled:
        ops.set_matmul_mode(ops.MATMUL_EXACT_F32)
        with torch.no_grad():
            model.predict(data)
            torch.cuda.synchronize()
            ops.prof_enable(True, kinds=[ops.PROF_FOURIER, ops.PROF_GEMM])
            t1 = time.perf_counter()
            n_exact = min(args.steps, 10)
            for _ in range(n_exact):
                pred_exact = model.predict(data)
                sharding.focal_records(ids, data, pred_exact)
            torch.cuda.synchronize()
            exact_ms = (time.perf_counter() - t1) / n_exact * 1e3
            ops.prof_enable(False)
            ops.set_matmul_mode(mode)
            pred_mode = model.predict(data)
        ef, eg = ops.prof_collect(ops.PROF_FOURIER), ops.prof_collect(ops.PROF_GEMM)
        e_tf = (ef[2] + eg[2]) / ((ef[1] + eg[1]) * 1e-3) / 1e12 if ef[1] + eg[1] > 0 else 0.0
        exact = {'ms_per_step': exact_ms, 'value': batch_size / (exact_ms * 1e-3), 'unit': 'scenarios/sec',
                 'mfma_achieved_tflops': e_tf, 'mfma_frac': e_tf / MFMA_PEAK_TFLOPS[ops.MATMUL_EXACT_F32],
                 'max_rel_diff_loc_refine_pos': float((pred_exact['loc_refine_pos'] - pred_mode['loc_refine_pos']).abs().max()
                                                      / pred_mode['loc_refine_pos'].abs().max())}
    ops.set_matmul_mode(mode)

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # HBM bytes per launch of the roofline kernel from the rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, calibrated
    # on a known byte count in the kernel's own access width: scripts/pmc_workload.py, scripts/pmc_summary.py); the
    # counters cannot be read from inside this process, so the committed summary of the same workload is quoted.
    traffic = None
    for name in ('r3_pmc_attention.json', 'r2_final_pmc_attention.json', 'r2_pmc_attention.json', 'r1_x_pmc_attention.json'):
        pmc_file = os.path.join(ROOT, 'profiles', name)
        if args.config == 'cfg3' and batch_size == cfg['batch'] and mode == ops.MATMUL_SPLIT_F16 and os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                traffic = json.load(fh)['attn_fused']['traffic_bytes_per_launch']
            break

    if rank == 0:
        copy_peak = stream_copy_peak(device)
        achieved = attn_bytes / (attn_ms * 1e-3) / 1e9 if attn_ms > 0 else 0.0
        chain_ms, chain_flop = f_ms + g_ms, f_flop + g_flop
        chain_tf = chain_flop / (chain_ms * 1e-3) / 1e12 if chain_ms > 0 else 0.0
        out = {
            'metric': 'scenarios/sec (TempEnsLearnAgg fwd, AV2-shape)', 'value': total / elapsed,
            'unit': 'scenarios/sec', 'n_gpus': world, 'steps': steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if pooled else 'weak',
            'vs_baseline': None, 'dtype': DTYPE_LABEL[mode], 'data': 'synthetic',
            'config': {'workload': '%s: TempEnsLearnAgg (10 sliding QCNet frames + TemporalAggregate), batch %d '
                                   'synthetic AV2-shaped scenarios per GPU per step (%d agents x 110 steps, %d polygons), random-init '
                                   'fp32 weights; %s' % (args.config, batch_size, cfg['scene']['num_agents'],
                                                         cfg['scene']['num_polygons'],
                                                         ('a pool of %d scenarios sharded by id over the ranks; batches built by %d loader worker '
                                                          'processes per rank (started before the timed region, as persistent workers '
                                                          'are), pinned and copied to the device inside it' % (total, args.workers)) if pooled else
                                                         'a different batch every step, scenario ids disjoint across ranks and steps'),
                       'global_batch': world * batch_size, 'parallelism': 'scenario-sharded x%d' % world,
                       'matmul': MATMUL_NOTE[mode]},
            'host_enqueue_ms_per_step': None if host_loop is None else host_loop * 1e3 / steps,   # < ms_per_step: the GPU is the bound
            'step_ms': None if not step_ms else {'median': statistics.median(step_ms), 'min': min(step_ms), 'max': max(step_ms), 'all': [round(v, 2) for v in step_ms],
                                                 'note': 'per-step GPU time between events on the main stream (rank 0); ms_per_step is '
                                                         'wall / steps of the pipelined loop'},
            'roofline': {'bound': 'hbm', 'kernel': 'attn_modes_kernel + attn_fused_kernel (qcn_attn_modes_*, qcn_attn_fused_f32: every positional attention launch)', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'achievable_peak': copy_peak, 'frac_of_achievable': achieved / copy_peak if copy_peak > 0 else None,
                         'launches': launches, 'avg_launch_us': attn_ms * 1e3 / max(launches, 1), 'frac_by_step': frac_by_step,
                         'alg_bytes_per_launch': attn_bytes / max(launches, 1),
                         'formula': '1552*E + 1024*n_dst (SURVEY 8(d), r-projection fused; 1296*E in the bf16 mode: 256-byte rhat rows) with E = the edges a launch '
                                    'processes: a mode-replicated relation is processed once per base edge for all K modes; traffic = HBM '
                                    'bytes per launch (PMC); achievable_peak = device-to-device copy of 1 GiB on this box; isolated = the same '
                                    'launches timed with the preparation stream off (in the timed region they share the CUs with the next '
                                    'batch\'s preparation kernels)', 'traffic': traffic,
                         'isolated': None if iso is None or iso[1] <= 0 else
                         {'achieved': iso[2] / (iso[1] * 1e-3) / 1e9, 'frac': iso[2] / (iso[1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          'launches': iso[0], 'avg_launch_us': iso[1] * 1e3 / max(iso[0], 1)}},
            'mfma': {'bound': 'mfma', 'kernel': 'fourier_kernel + ln_linear_kernel + attn_post_kernel + mlp_heads_kernel (the MFMA chain kernels)',
                     'achieved': chain_tf, 'peak': MFMA_PEAK_TFLOPS[mode], 'unit': 'TFLOP/s', 'frac': chain_tf / MFMA_PEAK_TFLOPS[mode],
                     'launches': f_launches + g_launches, 'steps': n_chain, 'ms_per_step': chain_ms / max(n_chain, 1),
                     'fourier': {'ms_per_step': f_ms / max(n_chain, 1), 'tflops': f_flop / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0},
                     'node': {'ms_per_step': g_ms / max(n_chain, 1), 'tflops': g_flop / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0},
                     'isolated': None if iso_f is None or iso_f[1] + iso_g[1] <= 0 else
                     {'achieved': (iso_f[2] + iso_g[2]) / ((iso_f[1] + iso_g[1]) * 1e-3) / 1e12,
                      'frac': (iso_f[2] + iso_g[2]) / ((iso_f[1] + iso_g[1]) * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[mode],
                      'launches': iso_f[0] + iso_g[0],
                      'fourier_tflops': iso_f[2] / (iso_f[1] * 1e-3) / 1e12 if iso_f[1] > 0 else 0.0,
                      'node_tflops': iso_g[2] / (iso_g[1] * 1e-3) / 1e12 if iso_g[1] > 0 else 0.0,
                      'note': 'the same launches with the preparation stream off: in the timed regime the Fourier launches of the '
                              'next batch run on the side stream beside the attention kernels, which lengthens both'},
                     'formula': 'fp32-equivalent flop 2*128*128 per row and Linear (SURVEY 8(d)) / summed launch durations (events on '
                                'the launch stream, in a pass of `steps` steps right after the timed region, same regime); peak = dense MFMA peak of the mode: 157 exact fp32, 2517 / 3 split-f16 '
                                '(three MFMAs per product), 2517 bf16'},
            'exact_f32_mfma': exact,
            'gathered_records': int(gathered.size(0)), 'metric_sums': dict(zip(sharding.VALIDATION_METRICS, [[float(a), float(b)] for a, b in sums.view(-1, 2).tolist()])),
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg['hparams'], cfg['scene'], model.state_dict(),
                                               {'tiny': 4, 'cfg4': 1}.get(args.config, 2), mode == ops.MATMUL_BF16,
                                               one_frame_only=args.config == 'cfg4')
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
