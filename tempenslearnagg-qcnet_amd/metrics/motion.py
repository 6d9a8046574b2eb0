"""Metric classes with the reference's names, constructor arguments, ``update`` signatures and ``sum`` / ``count`` states
(metrics/brier.py, min_ade.py, min_ahe.py, min_fde.py, min_fhe.py, mr.py, prob_mr.py, average_meter.py).

The reference derives them from ``torchmetrics.Metric`` (absent here); ``Metric`` below keeps the part of that interface the
reference uses: ``add_state(name, default, dist_reduce_fx='sum')``, ``update``, ``compute``, ``reset`` and the sum-reduction of
the states across ranks (``sync``: one ``all_reduce`` over RCCL, torchmetrics' ``dist_reduce_fx='sum'``).

Every ``update`` is ONE launch of ``qcn_motion_metrics_f32`` (csrc/metrics.hip): the kernel applies valid_filter and topk
(metrics/utils.py) per sample and writes the per-sample addends of all classes and criteria; the class sums its column on
the device (no host round trip; ``count`` stays a device tensor until ``compute``).  ``MotionMetrics`` shares one launch
between the seven metric objects of a validation step (predictors/temp_ens_learn_agg.py:264-275).
"""
from typing import Dict, Optional

import torch
import torch.distributed as dist

from .. import ops


class Metric:
    """The slice of ``torchmetrics.Metric`` the reference's metric classes rely on."""

    def __init__(self, **kwargs) -> None:
        self._defaults: Dict[str, torch.Tensor] = {}
        self._device = torch.device(kwargs.get('device', 'cpu'))

    @property
    def device(self) -> torch.device:
        return self._device

    def add_state(self, name: str, default: torch.Tensor, dist_reduce_fx: str = 'sum') -> None:
        if dist_reduce_fx != 'sum':
            raise ValueError('only sum-reduced states are used by the reference metrics')
        self._defaults[name] = default.clone()
        setattr(self, name, default.clone().to(self._device))

    def to(self, device) -> 'Metric':
        self._device = torch.device(device)
        for name in self._defaults:
            setattr(self, name, getattr(self, name).to(self._device))
        return self

    def reset(self) -> None:
        for name, default in self._defaults.items():
            setattr(self, name, default.clone().to(self._device))

    def sync(self) -> None:
        """Sum the states over the ranks of the default process group (no-op in a single process)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for name in self._defaults:
                state = getattr(self, name)
                dist.all_reduce(state, op=dist.ReduceOp.SUM)

    def _follow(self, t: torch.Tensor) -> None:
        if t.device != self._device:
            self.to(t.device)

    def __call__(self, *args, **kwargs):
        self.update(*args, **kwargs)
        return self.compute()


class AverageMeter(Metric):
    """metrics/average_meter.py:18-30."""

    def __init__(self, **kwargs) -> None:
        super().__init__(**kwargs)
        self.add_state('sum', default=torch.tensor(0.0), dist_reduce_fx='sum')
        self.add_state('count', default=torch.tensor(0), dist_reduce_fx='sum')

    def update(self, val: torch.Tensor) -> None:
        self._follow(val)
        self.sum += val.sum()
        self.count += val.numel()

    def compute(self) -> torch.Tensor:
        return self.sum / self.count


def _records(pred, target, prob, valid_mask, keep_invalid_final_step, max_guesses, with_heading=False, miss_threshold=2.0,
             select=None) -> torch.Tensor:
    """One kernel launch; with_heading: the last component of pred / target is the heading (metrics/min_ahe.py:47-52)."""
    return ops.motion_metrics(pred, target, prob=prob, valid_mask=valid_mask, select=select,
                              pred_head=pred.size(-1) - 1 if with_heading else ops.HEAD_NONE, target_head=-1,
                              max_guesses=max_guesses, keep_invalid_final_step=keep_invalid_final_step,
                              miss_threshold=miss_threshold)


class _MotionMetric(Metric):
    with_heading = False

    def __init__(self, max_guesses: int = 6, **kwargs) -> None:
        super().__init__(**kwargs)
        self.add_state('sum', default=torch.tensor(0.0), dist_reduce_fx='sum')
        self.add_state('count', default=torch.tensor(0), dist_reduce_fx='sum')
        self.max_guesses = max_guesses

    def _column(self, criterion: str) -> str:
        raise NotImplementedError

    def _accumulate(self, record: torch.Tensor, criterion: str) -> None:
        self._follow(record)
        self.sum += record[:, ops.METRIC_COLUMNS[self._column(criterion)]].sum()
        self.count += record[:, ops.METRIC_COLUMNS['counted']].sum().to(self.count.dtype)

    def compute(self) -> torch.Tensor:
        return self.sum / self.count


def _criterion(value: str, allowed) -> str:
    if value not in allowed:
        raise ValueError('{} is not a valid criterion'.format(value))
    return value


class Brier(_MotionMetric):
    """metrics/brier.py:24-56."""

    def _column(self, criterion):
        return {'FDE': 'brier_fde', 'ADE': 'brier_ade'}[criterion]

    def update(self, pred: torch.Tensor, target: torch.Tensor, prob: Optional[torch.Tensor] = None,
               valid_mask: Optional[torch.Tensor] = None, keep_invalid_final_step: bool = True,
               min_criterion: str = 'FDE') -> None:
        _criterion(min_criterion, ('FDE', 'ADE'))
        self._accumulate(_records(pred, target, prob, valid_mask, keep_invalid_final_step, self.max_guesses), min_criterion)


class minADE(_MotionMetric):
    """metrics/min_ade.py:24-57."""

    def _column(self, criterion):
        return {'FDE': 'ade_fde', 'ADE': 'ade_ade'}[criterion]

    def update(self, pred: torch.Tensor, target: torch.Tensor, prob: Optional[torch.Tensor] = None,
               valid_mask: Optional[torch.Tensor] = None, keep_invalid_final_step: bool = True,
               min_criterion: str = 'FDE') -> None:
        _criterion(min_criterion, ('FDE', 'ADE'))
        self._accumulate(_records(pred, target, prob, valid_mask, keep_invalid_final_step, self.max_guesses), min_criterion)


class minAHE(_MotionMetric):
    """metrics/min_ahe.py:25-58: pred / target carry the heading as their last component."""

    def _column(self, criterion):
        return {'FDE': 'ahe_fde', 'ADE': 'ahe_ade'}[criterion]

    def update(self, pred: torch.Tensor, target: torch.Tensor, prob: Optional[torch.Tensor] = None,
               valid_mask: Optional[torch.Tensor] = None, keep_invalid_final_step: bool = True,
               min_criterion: str = 'FDE') -> None:
        _criterion(min_criterion, ('FDE', 'ADE'))
        self._accumulate(_records(pred, target, prob, valid_mask, keep_invalid_final_step, self.max_guesses, with_heading=True),
                         min_criterion)


class minFDE(_MotionMetric):
    """metrics/min_fde.py:24-48."""

    def _column(self, criterion):
        return 'fde'

    def update(self, pred: torch.Tensor, target: torch.Tensor, prob: Optional[torch.Tensor] = None,
               valid_mask: Optional[torch.Tensor] = None, keep_invalid_final_step: bool = True) -> None:
        self._accumulate(_records(pred, target, prob, valid_mask, keep_invalid_final_step, self.max_guesses), 'FDE')


class minFHE(_MotionMetric):
    """metrics/min_fhe.py:25-51."""

    def _column(self, criterion):
        return 'fhe'

    def update(self, pred: torch.Tensor, target: torch.Tensor, prob: Optional[torch.Tensor] = None,
               valid_mask: Optional[torch.Tensor] = None, keep_invalid_final_step: bool = True) -> None:
        self._accumulate(_records(pred, target, prob, valid_mask, keep_invalid_final_step, self.max_guesses, with_heading=True),
                         'FDE')


class MR(_MotionMetric):
    """metrics/mr.py:24-57."""

    def _column(self, criterion):
        return {'FDE': 'miss_fde', 'MAXDE': 'miss_maxde'}[criterion]

    def update(self, pred: torch.Tensor, target: torch.Tensor, prob: Optional[torch.Tensor] = None,
               valid_mask: Optional[torch.Tensor] = None, keep_invalid_final_step: bool = True,
               miss_criterion: str = 'FDE', miss_threshold: float = 2.0) -> None:
        _criterion(miss_criterion, ('FDE', 'MAXDE'))
        self._accumulate(_records(pred, target, prob, valid_mask, keep_invalid_final_step, self.max_guesses,
                                  miss_threshold=miss_threshold), miss_criterion)


class ProbMR(_MotionMetric):
    """metrics/prob_mr.py:24-62."""

    def _column(self, criterion):
        return {'FDE': 'probmiss_fde', 'MAXDE': 'probmiss_maxde'}[criterion]

    def update(self, pred: torch.Tensor, target: torch.Tensor, prob: Optional[torch.Tensor] = None,
               valid_mask: Optional[torch.Tensor] = None, keep_invalid_final_step: bool = True,
               miss_criterion: str = 'FDE', miss_threshold: float = 2.0) -> None:
        _criterion(miss_criterion, ('FDE', 'MAXDE'))
        self._accumulate(_records(pred, target, prob, valid_mask, keep_invalid_final_step, self.max_guesses,
                                  miss_threshold=miss_threshold), miss_criterion)


class MotionMetrics:
    """The seven metric objects of ``TempEnsLearnAgg.validation_step`` (predictors/temp_ens_learn_agg.py:96-102,264-282) fed
    by ONE kernel launch per batch: the eval-mask selection (:250-254), the motion-vector headings (:255-260) and all seven
    state updates (:264-275) happen inside ``qcn_motion_metrics_f32``."""

    NAMES = ('Brier', 'minADE', 'minADE_best', 'minAHE', 'minFDE', 'minFHE', 'MR')
    _FEED = {'Brier': 'FDE', 'minADE': 'FDE', 'minADE_best': 'ADE', 'minAHE': 'FDE', 'minFDE': 'FDE', 'minFHE': 'FDE', 'MR': 'FDE'}

    def __init__(self, max_guesses: int = 6) -> None:
        self.Brier = Brier(max_guesses=max_guesses)
        self.minADE = minADE(max_guesses=max_guesses)
        self.minADE_best = minADE(max_guesses=max_guesses)
        self.minAHE = minAHE(max_guesses=max_guesses)
        self.minFDE = minFDE(max_guesses=max_guesses)
        self.minFHE = minFHE(max_guesses=max_guesses)
        self.MR = MR(max_guesses=max_guesses)
        self.max_guesses = max_guesses

    def update(self, traj: torch.Tensor, pi_softmax: torch.Tensor, target: torch.Tensor, valid_mask: torch.Tensor,
               eval_mask: torch.Tensor, output_dim: int = 2, output_head: bool = False) -> torch.Tensor:
        """traj [A, K, T, >= output_dim (+1)] agent-frame predictions of ALL agents, target [A, T, 4] (get_ground_truth),
        valid_mask [A, T], eval_mask [A].  Returns the per-agent records."""
        record = ops.motion_metrics(traj, target, prob=pi_softmax, valid_mask=valid_mask, select=eval_mask, pos_dim=output_dim,
                                    pred_head=output_dim if output_head else ops.HEAD_FROM_MOTION, target_head=-1,
                                    max_guesses=self.max_guesses)
        for name in self.NAMES:
            getattr(self, name)._accumulate(record, self._FEED[name])
        return record

    def sync(self) -> None:
        for name in self.NAMES:
            getattr(self, name).sync()

    def reset(self) -> None:
        for name in self.NAMES:
            getattr(self, name).reset()

    def compute(self) -> Dict[str, torch.Tensor]:
        return {'val_' + name: getattr(self, name).compute() for name in self.NAMES}
