"""Glue mirrored from the reference's src/model.py: the 1-step scheduler description.

The patched VAE forwards of src/model.py:14-54 have no Python counterpart here: skip capture and the
``sample + skip_conv_i(skip * gamma)`` injection are part of the planned op program (plan.py).
"""
from dataclasses import dataclass

import torch

from .plan import one_step_scheduler_constants


@dataclass
class OneStepScheduler:
    """What make_1step_sched() (src/model.py:7-11) yields, reduced to what the forward uses:
    DDPMScheduler(sd-turbo config).set_timesteps(1) -> timesteps [999] and the closed-form step
    prev = (x - sqrt(1-abar) eps) / sqrt(abar)  (the 1e-10-scaled variance noise is numerically dead)."""
    timestep: int = 999

    def __post_init__(self):
        self.sqrt_abar, self.sqrt_one_minus_abar = one_step_scheduler_constants(self.timestep)
        self.timesteps = torch.tensor([self.timestep], dtype=torch.long)


def make_1step_sched():
    return OneStepScheduler()
