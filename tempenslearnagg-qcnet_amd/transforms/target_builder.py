"""``TargetBuilder`` with the reference's constructor and call signature (transforms/target_builder.py:21-46): the regression
targets of every agent in the frame of its pose at the last historical step.  On a GPU-resident scenario it is one launch of
``qcn_to_agent_frame_f32``; the dataset transform runs on host tensors (DataLoader workers, before the batch is moved to the
GPU), where the same arithmetic is a handful of tensor ops."""
import torch

from .. import ops
from ..utils.geometry import wrap_angle


class TargetBuilder:

    def __init__(self, num_historical_steps: int, num_future_steps: int) -> None:
        self.num_historical_steps = num_historical_steps
        self.num_future_steps = num_future_steps

    def __call__(self, data):
        T = self.num_historical_steps
        agent = data['agent']
        position, heading = agent['position'], agent['heading']
        if position.is_cuda:
            agent['target'] = ops.to_agent_frame(position.contiguous(), heading.contiguous(), T - 1, T, self.num_future_steps)
            return data
        origin, theta = position[:, T - 1], heading[:, T - 1]
        cos, sin = theta.cos(), theta.sin()
        delta = position[:, T:T + self.num_future_steps, :2] - origin[:, :2].unsqueeze(1)
        target = origin.new_zeros(position.size(0), self.num_future_steps, 4)
        target[..., 0] = delta[..., 0] * cos.unsqueeze(1) + delta[..., 1] * sin.unsqueeze(1)
        target[..., 1] = -delta[..., 0] * sin.unsqueeze(1) + delta[..., 1] * cos.unsqueeze(1)
        if position.size(2) == 3:
            target[..., 2] = position[:, T:T + self.num_future_steps, 2] - origin[:, 2].unsqueeze(-1)
        target[..., 3] = wrap_angle(heading[:, T:T + self.num_future_steps] - theta.unsqueeze(-1))
        agent['target'] = target
        return data
