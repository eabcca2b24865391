"""Host-side support for the drop-in surface: `Config`, device selection and the
`tensor()` / `to_np()` boundary, schedules, the `run_steps` driver, tag / path helpers and the
logger facade.

One module mirrors what the reference spreads over deep_rl/utils/{config,torch_utils,schedule,
misc,logger}.py; every public name, signature and default is kept so that `examples.py` runs
against this package unchanged.  Reference lines are cited per item.
"""
import argparse
import datetime
import logging
import os
import time
from pathlib import Path

import numpy as np
import torch

from .normalizers import RescaleNormalizer


# ----------------------------------------------------------------------------------------------------
# Config (deep_rl/utils/config.py:11-89)
_DEFAULTS = dict(
    task_fn=None, optimizer_fn=None, actor_optimizer_fn=None, critic_optimizer_fn=None, network_fn=None,
    actor_network_fn=None, critic_network_fn=None, replay_fn=None, random_process_fn=None, discount=None,
    target_network_update_freq=None, exploration_steps=None, log_level=0, history_length=None, double_q=False,
    tag='vanilla', num_workers=1, gradient_clip=None, entropy_weight=0, use_gae=False, gae_tau=1.0,
    target_network_mix=0.001, min_memory_size=None, max_steps=0, rollout_length=None, value_loss_weight=1.0,
    iteration_log_interval=30, categorical_v_min=None, categorical_v_max=None, categorical_n_atoms=51,
    num_quantiles=None, optimization_epochs=4, mini_batch_size=64, termination_regularizer=0,
    sgd_update_frequency=None, random_action_prob=None, log_interval=int(1e3), save_interval=0, eval_interval=0,
    eval_episodes=10, async_actor=True, tasks=False, decaying_lr=False, shared_repr=False, noisy_linear=False,
    n_step=1,
)


class Config:
    DEVICE = torch.device('cpu')
    NOISY_LAYER_STD = 0.1
    DEFAULT_REPLAY = 'replay'
    PRIORITIZED_REPLAY = 'prioritized_replay'

    def __init__(self):
        self.parser = argparse.ArgumentParser()
        for name, value in _DEFAULTS.items():
            setattr(self, name, value)
        self.state_normalizer = RescaleNormalizer()
        self.reward_normalizer = RescaleNormalizer()
        self.replay_type = Config.DEFAULT_REPLAY
        self.__eval_env = None

    @property
    def eval_env(self):
        return self.__eval_env

    @eval_env.setter
    def eval_env(self, env):
        self.__eval_env = env
        self.state_dim = env.state_dim
        self.action_dim = env.action_dim
        self.task_name = env.name

    def add_argument(self, *args, **kwargs):
        self.parser.add_argument(*args, **kwargs)

    def merge(self, config_dict=None):
        if config_dict is None:
            config_dict = self.parser.parse_args().__dict__
        for key, value in config_dict.items():
            setattr(self, key, value)


# ----------------------------------------------------------------------------------------------------
# device + boundary helpers (deep_rl/utils/torch_utils.py:12-58)
def select_device(gpu_id):
    """gpu_id >= 0 selects that MI355X ('cuda:N' is the ROCm device in PyTorch-ROCm).  A negative
    id records 'cpu' like the reference, but every kernel-backed op then raises: there is no CPU path."""
    if gpu_id >= 0:
        Config.DEVICE = torch.device('cuda:%d' % gpu_id)
        if torch.cuda.is_available():
            torch.cuda.set_device(Config.DEVICE)
    else:
        Config.DEVICE = torch.device('cpu')


def tensor(x):
    if isinstance(x, torch.Tensor):
        return x
    x = np.asarray(x, dtype=np.float32)
    return torch.from_numpy(x).to(Config.DEVICE)


def range_tensor(end):
    return torch.arange(end).long().to(Config.DEVICE)


def to_np(t):
    return t.cpu().detach().numpy()


def random_seed(seed=None):
    np.random.seed(seed)
    torch.manual_seed(np.random.randint(int(1e6)))


def set_one_thread():
    os.environ['OMP_NUM_THREADS'] = '1'
    os.environ['MKL_NUM_THREADS'] = '1'
    torch.set_num_threads(1)


def huber(x, k=1.0):
    return torch.where(x.abs() < k, 0.5 * x.pow(2), k * (x.abs() - 0.5 * k))


def epsilon_greedy(epsilon, x):
    """RNG-order sensitive (torch_utils.py:51-58): 1-D draws rand() then maybe randint(); 2-D
    draws randint(A, size=N), then rand(N)."""
    if len(x.shape) == 1:
        return np.random.randint(len(x)) if np.random.rand() < epsilon else np.argmax(x)
    elif len(x.shape) == 2:
        random_actions = np.random.randint(x.shape[1], size=x.shape[0])
        greedy_actions = np.argmax(x, axis=-1)
        dice = np.random.rand(x.shape[0])
        return np.where(dice < epsilon, random_actions, greedy_actions)


# ----------------------------------------------------------------------------------------------------
# schedules (deep_rl/utils/schedule.py:7-31)
class ConstantSchedule:
    def __init__(self, val):
        self.val = val

    def __call__(self, steps=1):
        return self.val


class LinearSchedule:
    def __init__(self, start, end=None, steps=None):
        if end is None:
            end, steps = start, 1
        self.inc = (end - start) / float(steps)
        self.current = start
        self.end = end
        self.bound = min if end > start else max

    def __call__(self, steps=1):
        value = self.current
        self.current = self.bound(self.current + self.inc * steps, self.end)
        return value


# ----------------------------------------------------------------------------------------------------
# driver + helpers (deep_rl/utils/misc.py:19-84)
def run_steps(agent):
    """The training driver (the cadence of misc.py:19-35): before EVERY agent.step() -- checkpoint every save_interval
    steps to data/<Agent>-<tag>-<steps>, a 'steps %d, %.2f steps/s' line every log_interval steps (rate since the previous
    line), evaluation every eval_interval steps, stop (and close the agent) once total_steps reaches max_steps; after the
    step, agent.switch_task().  The interval tests are on total_steps exactly, so an agent whose step() advances
    total_steps by k only triggers on multiples that it lands on (reference behaviour)."""
    config = agent.config
    name = type(agent).__name__
    mark = time.time()

    def due(interval):
        return bool(interval) and agent.total_steps % interval == 0

    while True:
        if due(config.save_interval):
            agent.save('data/%s-%s-%d' % (name, config.tag, agent.total_steps))
        if due(config.log_interval):
            now = time.time()
            agent.logger.info('steps %d, %.2f steps/s' % (agent.total_steps, config.log_interval / (now - mark)))
            mark = time.time()
        if due(config.eval_interval):
            agent.eval_episodes()
        if config.max_steps and agent.total_steps >= config.max_steps:
            agent.close()
            return
        agent.step()
        agent.switch_task()


def get_time_str():
    return datetime.datetime.now().strftime("%y%m%d-%H%M%S")


def get_default_log_dir(name):
    return './log/%s-%s' % (name, get_time_str())


def mkdir(path):
    Path(path).mkdir(parents=True, exist_ok=True)


def close_obj(obj):
    if hasattr(obj, 'close'):
        obj.close()


def random_sample(indices, batch_size):
    """misc.py:55-62: one np.random.permutation; full minibatches, then the remainder."""
    indices = np.asarray(np.random.permutation(indices))
    full = len(indices) // batch_size * batch_size
    for batch in indices[:full].reshape(-1, batch_size):
        yield batch
    if len(indices) % batch_size:
        yield indices[full:]


def is_plain_type(x):
    return isinstance(x, (str, int, float, bool))


def generate_tag(params):
    """misc.py:72-84: builds params['tag'] from the sorted kwargs unless one is given."""
    if 'tag' in params.keys():
        return
    game = params['game']
    params.setdefault('run', 0)
    run = params['run']
    del params['game']
    del params['run']
    parts = ['%s_%s' % (k, v if is_plain_type(v) else v.__name__) for k, v in sorted(params.items())]
    params['tag'] = '%s-%s-run-%d' % (game, '-'.join(parts), run)
    params['game'] = game
    params['run'] = run


# ----------------------------------------------------------------------------------------------------
# logger facade (the interface of deep_rl/utils/logger.py:17-73)
def get_logger(tag='default', log_level=0):
    """A Logger whose text lines also go to ./log/<tag>-<time>.txt and whose scalars go to a tensorboard run under
    ./tf_log/ (when tensorboard is installed) -- the two places the reference's plot tools read."""
    text = logging.getLogger()
    text.setLevel(logging.INFO)
    if tag is not None:
        try:
            mkdir('./log')
            handler = logging.FileHandler('./log/%s-%s.txt' % (tag, get_time_str()))
        except OSError:
            handler = None
        if handler is not None:
            handler.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s: %(message)s'))
            handler.setLevel(logging.INFO)
            text.addHandler(handler)
    return Logger(text, './tf_log/logger-%s-%s' % (tag, get_time_str()), log_level)


class Logger(object):
    """info / debug / warning of a python logger + add_scalar / add_histogram with the reference's signatures
    (tag, value, step=None, log_level=0): entries above the logger's level are dropped, a missing step is the count of
    earlier entries under the same tag.  Scalars are also kept in `self.scalars[tag]` as (step, value)."""

    def __init__(self, vanilla_logger, log_dir, log_level=0):
        self.log_level, self.log_dir = log_level, log_dir
        self.writer = None              # created on first use; False when tensorboard is unavailable
        self.all_steps, self.scalars = {}, {}
        if vanilla_logger is not None:
            self.info, self.debug, self.warning = vanilla_logger.info, vanilla_logger.debug, vanilla_logger.warning

    def lazy_init_writer(self):
        if self.writer is None:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(self.log_dir)
            except Exception:
                self.writer = False
        return self.writer

    def get_step(self, tag):
        step = self.all_steps.get(tag, 0)
        self.all_steps[tag] = step + 1
        return step

    def _entry(self, tag, step, log_level):
        if log_level > self.log_level:
            return None, None
        return (self.get_step(tag) if step is None else step), self.lazy_init_writer()

    def add_scalar(self, tag, value, step=None, log_level=0):
        step, writer = self._entry(tag, step, log_level)
        if step is None:
            return
        self.scalars.setdefault(tag, []).append((step, float(value)))
        if writer:
            writer.add_scalar(tag, value, step)

    def add_histogram(self, tag, values, step=None, log_level=0):
        step, writer = self._entry(tag, step, log_level)
        if step is not None and writer:
            writer.add_histogram(tag, values, step)
