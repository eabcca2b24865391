"""deeprl_amd -- MI355X-native rollout -> replay -> update hot path behind the deep_rl surface.

`from deeprl_amd import *` exposes the names `from deep_rl import *` does (examples.py:7 relies on
the star import for Config, Task, run_steps, the replay / network / agent classes AND for `torch`,
`np`, `F`, `nn`, `mp`).  `install_as_deep_rl()` registers this package under the name `deep_rl` so
the reference's examples.py runs unchanged.
"""
import sys

import numpy as np
import torch
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from ._lib import DraError
from .support import (Config, ConstantSchedule, LinearSchedule, Logger, close_obj, epsilon_greedy, generate_tag,
                      get_default_log_dir, get_logger, get_time_str, huber, is_plain_type, mkdir, random_sample,
                      random_seed, range_tensor, run_steps, select_device, set_one_thread, tensor, to_np)
from .normalizers import (BaseNormalizer, ImageNormalizer, MeanStdNormalizer, RescaleNormalizer, RunningMeanStd,
                          SignNormalizer)
from .replay import (PrioritizedReplay, PrioritizedTransition, ReplayWrapper, Storage, Transition, UniformReplay)
from .envs import LazyFrames, Task
from .nets import (BaseNet, CategoricalActorCriticNet, CategoricalNet, DDPGConvBody, DeterministicActorCriticNet,
                   DuelingNet, DummyBody, FCBody, GaussianActorCriticNet, NatureConvBody, NoisyLinear, OptionCriticNet,
                   QuantileNet, RainbowNet, TD3Net, VanillaNet, layer_init)
from .agents import (A2CAgent, BaseActor, BaseAgent, CategoricalDQNActor, CategoricalDQNAgent, DDPGAgent, DQNActor,
                     DQNAgent, NStepDQNAgent, OptionCriticAgent, PPOAgent, QuantileRegressionDQNActor,
                     QuantileRegressionDQNAgent, TD3Agent)
from .random_process import GaussianProcess, OrnsteinUhlenbeckProcess, RandomProcess


def install_as_deep_rl():
    """Registers this package as `deep_rl` (and the usual sub-module names) in sys.modules."""
    me = sys.modules[__name__]
    for name in ('deep_rl', 'deep_rl.agent', 'deep_rl.component', 'deep_rl.network', 'deep_rl.utils'):
        sys.modules[name] = me
    return me
