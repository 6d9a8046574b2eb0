"""CPU ORACLE — test infrastructure, never the product.

Plain-PyTorch / numpy restatement of the evaluation side of the path: the metric classes' state
updates (metrics/*.py), valid_filter / topk (metrics/utils.py), the ground-truth and post-processing frame transforms
(predictors/temp_ens_learn_agg.py:104-121, predictors/qcnet_sliding.py:179-197), the NMS ensembling baseline
(utils/nms.py:11-38) and its scoring (predictors/qcnet_sliding.py:48-54).  Only ``tests/`` imports this file.

Pinning status:
* ``nms``: PINNED — ``oracle/make_golden.py`` imports the reference's own ``utils/nms.py`` (pure numpy) and writes
  ``tests/golden/ref_nms.npz``; ``tests/test_oracle_pinned.py`` checks this restatement against it.
* metric classes: PARITY UNPINNED — the reference's classes derive from ``torchmetrics.Metric`` and import
  ``torch_scatter`` (neither installed here or on the GPU box) and the reference holds no tests or fixtures for them.  The
  functions below follow the ``update`` bodies line by line on plain tensors; ``sum`` / ``count`` are returned.
* ``compute_ade`` / ``compute_fde`` are from the absent ``av2`` package (environment.yml: av2==0.2.1,
  ``av2/datasets/motion_forecasting/eval/metrics.py``): published algorithm restated (per-mode mean / final L2 displacement).
"""
import math
from typing import Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


def wrap_angle(angle: Tensor, min_val: float = -math.pi, max_val: float = math.pi) -> Tensor:
    """utils/geometry.py:47-51."""
    return min_val + (angle + max_val) % (max_val - min_val)


def topk(max_guesses: int, pred: Tensor, prob: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """metrics/utils.py:22-56 (per-sample branch: joint=False, ptr=None)."""
    max_guesses = min(max_guesses, pred.size(1))
    if max_guesses == pred.size(1):
        if prob is not None:
            prob = prob / prob.sum(dim=-1, keepdim=True)
        else:
            prob = pred.new_ones((pred.size(0), max_guesses)) / max_guesses
        return pred, prob
    if prob is not None:
        inds_topk = torch.topk(prob, k=max_guesses, dim=-1, largest=True, sorted=True)[1]
        rows = torch.arange(pred.size(0)).unsqueeze(-1).expand(-1, max_guesses)
        pred_topk = pred[rows, inds_topk]
        prob_topk = prob[rows, inds_topk]
        prob_topk = prob_topk / prob_topk.sum(dim=-1, keepdim=True)
    else:
        pred_topk = pred[:, :max_guesses]
        prob_topk = pred.new_ones((pred.size(0), max_guesses)) / max_guesses
    return pred_topk, prob_topk


def valid_filter(pred: Tensor, target: Tensor, prob: Optional[Tensor] = None, valid_mask: Optional[Tensor] = None,
                 keep_invalid_final_step: bool = True):
    """metrics/utils.py:59-82 without the ptr bookkeeping."""
    if valid_mask is None:
        valid_mask = target.new_ones(target.size()[:-1], dtype=torch.bool)
    filter_mask = valid_mask.any(dim=-1) if keep_invalid_final_step else valid_mask[:, -1]
    return (pred[filter_mask], target[filter_mask], prob[filter_mask] if prob is not None else None,
            valid_mask[filter_mask])


def _inds_last(valid_mask: Tensor) -> Tensor:
    return (valid_mask * torch.arange(1, valid_mask.size(-1) + 1)).argmax(dim=-1)


def metric_update(name: str, pred: Tensor, target: Tensor, prob: Optional[Tensor] = None,
                  valid_mask: Optional[Tensor] = None, keep_invalid_final_step: bool = True, criterion: str = 'FDE',
                  max_guesses: int = 6, miss_threshold: float = 2.0) -> Tuple[Tensor, int]:
    """(sum addend, count addend) of one ``update`` call of the reference class ``name``."""
    pred, target, prob, valid_mask = valid_filter(pred, target, prob, valid_mask, keep_invalid_final_step)
    pred_topk, prob_topk = topk(max_guesses, pred, prob)
    n = pred.size(0)
    rows = torch.arange(n)
    inds_last = _inds_last(valid_mask)
    heading = name in ('minAHE', 'minFHE')                     # position = all but the last component (min_ahe.py:47-49)
    pp = pred_topk[..., :-1] if heading else pred_topk
    tp = target[..., :-1] if heading else target
    dist = torch.norm(pp - tp.unsqueeze(1), p=2, dim=-1)       # [n, G, T]
    fde = torch.norm(pp[rows, :, inds_last] - tp[rows, inds_last].unsqueeze(-2), p=2, dim=-1)   # [n, G]
    if name == 'Brier':                                        # metrics/brier.py:37-52
        best = fde.argmin(dim=-1) if criterion == 'FDE' else (dist * valid_mask.unsqueeze(1)).sum(dim=-1).argmin(dim=-1)
        s = (1.0 - prob_topk[rows, best]).pow(2).sum()
    elif name == 'minADE':                                     # metrics/min_ade.py:36-54
        if criterion == 'FDE':
            best = fde.argmin(dim=-1)
            s = ((torch.norm(pred_topk[rows, best] - target, p=2, dim=-1) * valid_mask).sum(dim=-1) /
                 valid_mask.sum(dim=-1)).sum()
        else:
            s = ((dist * valid_mask.unsqueeze(1)).sum(dim=-1).min(dim=-1)[0] / valid_mask.sum(dim=-1)).sum()
    elif name == 'minAHE':                                     # metrics/min_ahe.py:37-53
        best = fde.argmin(dim=-1) if criterion == 'FDE' else (dist * valid_mask.unsqueeze(1)).sum(dim=-1).argmin(dim=-1)
        s = ((wrap_angle(pred_topk[rows, best, :, -1] - target[..., -1]).abs() * valid_mask).sum(dim=-1) /
             valid_mask.sum(dim=-1)).sum()
    elif name == 'minFDE':                                     # metrics/min_fde.py:36-45
        s = fde.min(dim=-1)[0].sum()
    elif name == 'minFHE':                                     # metrics/min_fhe.py:36-48
        best = fde.argmin(dim=-1)
        s = wrap_angle(pred_topk[rows, best, inds_last, -1] - target[rows, inds_last, -1]).abs().sum()
    elif name == 'MR':                                         # metrics/mr.py:38-54
        if criterion == 'FDE':
            s = (fde.min(dim=-1)[0] > miss_threshold).sum()
        else:
            s = ((dist * valid_mask.unsqueeze(1)).max(dim=-1)[0].min(dim=-1)[0] > miss_threshold).sum()
    elif name == 'ProbMR':                                     # metrics/prob_mr.py:38-59
        if criterion == 'FDE':
            m, best = fde.min(dim=-1)
        else:
            m, best = (dist * valid_mask.unsqueeze(1)).max(dim=-1)[0].min(dim=-1)
        s = torch.where(m > miss_threshold, torch.ones_like(m), 1.0 - prob_topk[rows, best]).sum()
    else:
        raise ValueError(name)
    return s.double(), n


def get_ground_truth(position: Tensor, heading: Tensor, num_historical_steps: int, num_future_steps: int,
                     time_shift: int = 10) -> Tensor:
    """predictors/temp_ens_learn_agg.py:104-121."""
    origin = position[:, num_historical_steps - 1 + time_shift]
    theta = heading[:, num_historical_steps - 1 + time_shift]
    cos, sin = theta.cos(), theta.sin()
    rot_mat = theta.new_zeros(position.size(0), 2, 2)
    rot_mat[:, 0, 0] = cos
    rot_mat[:, 0, 1] = -sin
    rot_mat[:, 1, 0] = sin
    rot_mat[:, 1, 1] = cos
    target = origin.new_zeros(position.size(0), num_future_steps - 10, 4)
    target[..., :2] = torch.bmm(position[:, num_historical_steps + time_shift:, :2] - origin[:, :2].unsqueeze(1), rot_mat)
    if position.size(2) == 3:
        target[..., 2] = position[:, num_historical_steps + time_shift:, 2] - origin[:, 2].unsqueeze(-1)
    target[..., 3] = wrap_angle(heading[:, num_historical_steps + time_shift:] - theta.unsqueeze(-1))
    return target


def postprocess(position: Tensor, heading: Tensor, eval_mask: Tensor, traj_refine: Tensor, pi: Tensor, time_shift: int,
                num_historical_steps: int = 50) -> Tuple[Tensor, Tensor]:
    """predictors/qcnet_sliding.py:179-197."""
    origin_eval = position[eval_mask, num_historical_steps - 1 + time_shift]
    theta_eval = heading[eval_mask, num_historical_steps - 1 + time_shift]
    cos, sin = theta_eval.cos(), theta_eval.sin()
    rot_mat = torch.zeros(int(eval_mask.sum()), 2, 2)
    rot_mat[:, 0, 0] = cos
    rot_mat[:, 0, 1] = sin
    rot_mat[:, 1, 0] = -sin
    rot_mat[:, 1, 1] = cos
    traj_eval = torch.matmul(traj_refine[eval_mask, :, :, :2], rot_mat.unsqueeze(1)) + origin_eval[:, :2].reshape(-1, 1, 1, 2)
    return traj_eval[:, :, 10 - time_shift: 10 - time_shift + 50], torch.softmax(pi[eval_mask], dim=-1)


def validation_metrics(traj_refine: Tensor, pi: Tensor, target: Tensor, reg_mask: Tensor, eval_mask: Tensor,
                       output_dim: int = 2):
    """predictors/temp_ens_learn_agg.py:249-275 (output_head False): {metric name: (sum, count)}."""
    gt = torch.cat([target[..., :output_dim], target[..., -1:]], dim=-1)
    valid_mask_eval = reg_mask[eval_mask]
    traj_eval = traj_refine[eval_mask, :, :, :output_dim]
    start = torch.cat([traj_eval.new_zeros((traj_eval.size(0), traj_eval.size(1), 1, 2)), traj_eval[..., :2]], dim=-2)
    motion = start[:, :, 1:] - start[:, :, :-1]
    head_eval = torch.atan2(motion[..., 1], motion[..., 0])
    traj_eval = torch.cat([traj_eval, head_eval.unsqueeze(-1)], dim=-1)
    pi_eval = torch.softmax(pi[eval_mask], dim=-1)
    gt_eval = gt[eval_mask]
    pos, gpos = traj_eval[..., :output_dim], gt_eval[..., :output_dim]
    return {
        'Brier': metric_update('Brier', pos, gpos, pi_eval, valid_mask_eval),
        'minADE': metric_update('minADE', pos, gpos, pi_eval, valid_mask_eval),
        'minADE_best': metric_update('minADE', pos, gpos, pi_eval, valid_mask_eval, criterion='ADE'),
        'minAHE': metric_update('minAHE', traj_eval, gt_eval, pi_eval, valid_mask_eval),
        'minFDE': metric_update('minFDE', pos, gpos, pi_eval, valid_mask_eval),
        'minFHE': metric_update('minFHE', traj_eval, gt_eval, pi_eval, valid_mask_eval),
        'MR': metric_update('MR', pos, gpos, pi_eval, valid_mask_eval),
    }


def nms(out: np.ndarray, clfs: np.ndarray, tgt_selection: int = 6, radius: float = 2.5) -> Tuple[np.ndarray, np.ndarray]:
    """utils/nms.py:11-38: greedy selection by descending confidence; a candidate whose END POINT lies within ``radius``
    of a selected trajectory's end point is suppressed; short selections are topped up by confidence.  Returns the
    selected indices and their confidences."""
    index = np.argsort(clfs, axis=0)
    selected = []
    candidates = index.tolist()
    while len(candidates) > 0:
        selected.append(candidates.pop())
        end = out[selected[-1]][-1]
        candidates = [i for i in candidates if not np.linalg.norm(end - out[i][-1]) < radius]
        if len(selected) == tgt_selection:
            break
    for i in index[::-1].tolist():
        if len(selected) >= tgt_selection:
            break
        if i not in selected:
            selected.append(i)
    sel = np.asarray(selected, dtype=np.int64)
    return sel, clfs[sel]


def calculate_metrics(trajs: np.ndarray, gt: np.ndarray) -> Tuple[float, float, int]:
    """predictors/qcnet_sliding.py:48-54 over av2's compute_fde / compute_ade (restated, see module docstring)."""
    disp = np.linalg.norm(trajs - gt[None], axis=-1)           # [K, T]
    min_fde = disp[:, -1].min()
    min_ade = disp.mean(axis=-1).min()
    return float(min_fde), float(min_ade), 1 if min_fde > 2.0 else 0
