"""Evaluation metrics of the reference's ``metrics`` package (metrics/__init__.py:14-21) on the HIP metrics kernel."""
from .motion import AverageMeter, Brier, MR, Metric, MotionMetrics, ProbMR, minADE, minAHE, minFDE, minFHE

__all__ = ['AverageMeter', 'Brier', 'MR', 'Metric', 'MotionMetrics', 'ProbMR', 'minADE', 'minAHE', 'minFDE', 'minFHE']
