"""Agents with the reference's `Agent(config).step()` surface and control flow, whose per-step
arithmetic runs on the HIP kernels.

Reference: deep_rl/agent/BaseAgent.py:15-182 (BaseAgent, BaseActor), DQN_agent.py:14-138,
CategoricalDQN_agent.py:14-89, QuantileRegressionDQN_agent.py:14-77, A2C_agent.py:12-64,
PPO_agent.py:12-100, NStepDQN_agent.py:12-67.

Differences that are the point of the rewrite (results unchanged):
  * actor / replay "processes" are in-process objects: the replay is an HBM ring, so the
    pickle-over-pipe hops of BaseAgent.py:142-172 / replay.py:219-278 have nothing left to hide;
    `async_replay` is accepted and has no effect; `config.async_actor` selects, for DQNAgent on the dqn_pixel
    configuration with a device-resident environment, the two-stream actor / learner pipeline of csrc/learner.hip
    (DQNAgent._attach_device_pipeline) and is otherwise a no-op;
  * TD / C51 / QR / PPO / A2C losses are single fused kernels that emit the gradient w.r.t. the
    network output, which is pushed through the HIP contractions by autograd;
  * clip_grad_norm_ + optimizer.step() are two launches over one flat buffer (optim.py);
  * GAE / n-step returns are one chunked-scan launch instead of T python iterations;
  * DQNAgent on the dqn_pixel configuration (VanillaNet(NatureConvBody), RMSprop, UniformReplay of 84x84 uint8
    frames, history 4 -- examples.py:55-97) switches to the fused learner of csrc/learner.hip: one C call per
    update (gather + forward + TD loss + backward + clip + RMSprop as a captured hipGraph; PrioritizedReplay adds
    the IS-weight / priority kernel and runs the chain eagerly) and one C call per actor forward;
    `config.fused_learner = False` keeps the generic autograd path.
"""
import pickle
import os

import contextlib
import gc

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .envs import LazyFrames
from .optim import FlatParams, FusedOptimizer, nature_conv_weights
from .replay import PrioritizedTransition, Storage
from ._lib import DraError
from .support import Config, LinearSchedule, close_obj, epsilon_greedy, get_logger, random_sample, range_tensor, tensor, to_np



@contextlib.contextmanager
def _capture(graph):
    """torch.cuda.graph(graph) with the cyclic garbage collector held off for the duration of the capture.  A collection that
    runs INSIDE a capture may finalise objects of an earlier agent that own HIP resources (pinned host tensors, the fused
    learner's library handle): their frees / device synchronisation are illegal while a stream captures and end the process
    (std::terminate out of a deleter) -- seen as a sporadic abort of the test suite at the first captured update that followed a
    closed device-pipeline agent.  Collect first, then capture without the collector."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was:
            gc.enable()


class _NullLock:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class BaseAgent:
    """What every agent shares with the outside world (the surface of BaseAgent.py:15-105): checkpoint files in the
    reference's format, evaluation episodes, the episodic-return log lines / scalars its plotting tools parse, task
    switching.  The file formats and log strings are interchange formats and are kept byte for byte:
      '<name>.model'  torch.save of network.state_dict() -- the reference's key names, [OC,C,KH,KW] conv weights;
      '<name>.stats'  pickle of config.state_normalizer.state_dict();
      'steps %d, episodic_return_test %.2f(%.2f)' / 'steps %d, episodic_return_train %s' (logger.py / plot.py tags)."""

    def __init__(self, config):
        self.config = config
        self.logger = get_logger(tag=config.tag, log_level=config.log_level)
        self.task_ind = 0

    def close(self):
        close_obj(self.task)

    # -- checkpoints -------------------------------------------------------------------------------------------
    def save(self, filename):
        dp = getattr(self, 'dp', None)
        if dp is not None and not dp.is_main:
            return     # data parallel: every rank holds the same parameters; rank 0 writes the one checkpoint file
        # parameters may be strided views of a flat (KOC) buffer: write plain contiguous CPU tensors, as the reference does
        weights = {k: v.detach().cpu().contiguous().clone() for k, v in self.network.state_dict().items()}
        torch.save(weights, filename + '.model')
        with open(filename + '.stats', 'wb') as f:
            pickle.dump(self.config.state_normalizer.state_dict(), f)

    def load(self, filename):
        weights = torch.load(filename + '.model', map_location='cpu')
        self.network.load_state_dict(weights)
        with open(filename + '.stats', 'rb') as f:
            self.config.state_normalizer.load_state_dict(pickle.load(f))

    # -- evaluation --------------------------------------------------------------------------------------------
    def eval_step(self, state):
        raise NotImplementedError

    def eval_episode(self):
        env = self.config.eval_env
        state = env.reset()
        ret = None
        while ret is None:
            state, _, _, info = env.step(self.eval_step(state))
            ret = info[0]['episodic_return']
        return ret

    def eval_episodes(self):
        returns = np.asarray([np.sum(self.eval_episode()) for _ in range(self.config.eval_episodes)])
        mean, sem = returns.mean(), returns.std() / np.sqrt(len(returns))
        self.logger.info('steps %d, episodic_return_test %.2f(%.2f)' % (self.total_steps, mean, sem))
        self.logger.add_scalar('episodic_return_test', mean, self.total_steps)
        return {'episodic_return_test': mean}

    def record_online_return(self, info, offset=0):
        """info: one env's dict or the vector env's tuple of dicts; a finished episode logs its return at
        total_steps + the env's position in the tuple."""
        if isinstance(info, tuple):
            for i, one in enumerate(info):
                self.record_online_return(one, i)
            return
        if not isinstance(info, dict):
            raise NotImplementedError
        ret = info['episodic_return']
        if ret is not None:
            at = self.total_steps + offset
            self.logger.add_scalar('episodic_return_train', ret, at)
            self.logger.info('steps %d, episodic_return_train %s' % (at, ret))

    def switch_task(self):
        """config.tasks: consecutive equal shares of max_steps per task."""
        tasks = self.config.tasks
        if not tasks:
            return
        boundary = np.linspace(0, self.config.max_steps, len(tasks) + 1)[self.task_ind + 1]
        if self.total_steps > boundary:
            self.task_ind += 1
            self.task = tasks[self.task_ind]
            self.states = self.config.state_normalizer(self.task.reset())


class BaseActor:
    """BaseAgent.py:108-182 without the process: `step()` returns `sgd_update_frequency`
    transitions, in the order the reference's sync path (`async_actor=False`) produces them."""
    STEP, RESET, EXIT, SPECS, NETWORK, CACHE = range(6)

    def __init__(self, config):
        self.config = config
        self._state = None
        self._task = None
        self._network = None
        self._total_steps = 0
        self._set_up()
        self._task = config.task_fn()

    def start(self):
        return None

    def _sample(self):
        transitions = []
        for _ in range(self.config.sgd_update_frequency):
            transition = self._transition()
            if transition is not None:
                transitions.append(transition)
        return transitions

    def step(self):
        return self._sample()

    def _transition(self):
        raise NotImplementedError

    def _set_up(self):
        pass

    def close(self):
        close_obj(self._task)

    def set_network(self, net):
        self._network = net


# ==================================================================================================== DQN family
class DQNActor(BaseActor):
    """DQN_agent.py:14-45."""

    def __init__(self, config):
        BaseActor.__init__(self, config)
        self.config = config
        self._fast_q = None
        self._graphed_q = _GraphedQ(self) if Config.DEVICE.type == 'cuda' else None
        self.start()

    def compute_q(self, prediction):
        return to_np(prediction['q'])

    def q_device(self, prediction):
        """The action values compute_q() brings to the host, as a device tensor (graph-captured actor forward)."""
        return prediction['q']

    def _transition(self):
        if self._state is None:
            self._state = self._task.reset()
        config = self.config
        if config.noisy_linear:
            self._network.reset_noise()
        q_values = None
        if self._fast_q is not None:     # fused learner attached (DQNAgent._attach_fused_learner)
            q_values = self._fast_q(self._state)
        elif self._graphed_q is not None and self._graphed_q.usable(self._state):
            q_values = self._graphed_q(self._state)     # None during its warm-up calls
        if q_values is None:
            with config.lock:
                with torch.no_grad():
                    prediction = self._network(config.state_normalizer(self._state))
            q_values = self.compute_q(prediction)
        if config.noisy_linear:
            epsilon = 0
        elif self._total_steps < config.exploration_steps:
            epsilon = 1
        else:
            epsilon = config.random_action_prob()
        action = epsilon_greedy(epsilon, q_values)
        next_state, reward, done, info = self._task.step(action)
        entry = [self._state, action, reward, next_state, done, info]
        self._total_steps += 1
        self._state = next_state
        return entry


def _capture_failed(config, what, error):
    """A hipGraph capture failed.  Falling back to the eager kernels silently would turn a broken capture into nothing but a
    slower number, so it is an error unless the configuration opts into the fallback (config.allow_eager_fallback = True:
    e.g. a user network with host-side control flow)."""
    if not getattr(config, 'allow_eager_fallback', False):
        raise RuntimeError("%s could not be captured as a hipGraph (%r); set config.allow_eager_fallback = True to run it "
                           "eagerly, or config.graph_update = False to skip capturing" % (what, error)) from error
    import warnings
    warnings.warn("%s could not be captured as a graph (%r); using the eager path" % (what, error))


class _GraphedQ:
    """DQNActor's forward (DQN_agent.py:29-33) as ONE hipGraph replay for image observations: the uint8
    [1,C,H,W] observation goes through pinned staging into a static device buffer; normaliser table kernel,
    network forward and the action-value reduction replay from a graph captured (torch.cuda.graphs) after four
    eager calls (one agent step: by then a DQNAgent that qualifies for the fused learner has switched to it).
    Same kernels, same arguments as the eager path -> bit-identical action values."""
    WARMUP = 4

    def __init__(self, actor):
        self.actor = actor
        self.calls = 0
        self.graph = None
        self.static_in = self.static_out = None
        self.stage = []
        self.events = []
        self.k = 0
        self.failed = False

    def usable(self, state):
        from .normalizers import RescaleNormalizer
        cfg = self.actor.config
        if self.failed or cfg.noisy_linear or getattr(cfg, 'graph_update', True) is False:
            return False
        if not isinstance(cfg.state_normalizer, RescaleNormalizer):
            return False
        first = state[0] if isinstance(state, (list, tuple)) else state
        return isinstance(first, LazyFrames) or (isinstance(first, np.ndarray) and first.dtype == np.uint8)

    def __call__(self, state):
        a = self.actor
        cfg = a.config
        self.calls += 1
        if self.calls <= self.WARMUP:
            return None
        x = np.ascontiguousarray(np.asarray(state, dtype=np.uint8))
        dev = next(a._network.parameters()).device
        if self.graph is None:
            try:
                self.static_in = torch.zeros(x.shape, dtype=torch.uint8, device=dev)
                self.stage = [torch.empty(x.shape, dtype=torch.uint8).pin_memory() for _ in range(4)]
                self.events = [None] * 4
                with torch.no_grad():   # eager dry run of exactly what is captured: creates every lazily built
                    a.q_device(a._network(cfg.state_normalizer(self.static_in)))   # object (normaliser table, ...)
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with _capture(g):
                    with torch.no_grad():
                        self.static_out = a.q_device(a._network(cfg.state_normalizer(self.static_in)))
                self.graph = g
            except Exception as e:      # e.g. a network with host-side control flow: stay on the eager path
                _capture_failed(cfg, "the actor forward", e)
                self.failed = True
                self.graph = None
                return None
        k = self.k
        self.k = (k + 1) % len(self.stage)
        if self.events[k] is not None:
            self.events[k].synchronize()
        self.stage[k].numpy()[...] = x
        self.static_in.copy_(self.stage[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        self.graph.replay()
        return to_np(self.static_out)


class _GraphedUpdate:
    """One DQN-family update (DQN_agent.py:114-134: normalise, target / online forwards, fused loss kernel,
    backward, clip, optimizer) as ONE hipGraph replay over static minibatch buffers that the ring gather refills
    in place.  Captured with torch.cuda.graphs after two eager updates (which also run every lazy one-time
    initialisation); same kernels and arguments as the eager path -> bit-identical parameters.  Uniform replay
    only: the PER importance exponent changes every update and is a kernel argument."""
    WARMUP = 2

    def __init__(self, agent):
        self.agent = agent
        self.updates = 0
        self.graph = None
        self.static = None
        self.out = None
        self.signature = None
        self.failed = False

    def usable(self, rp):
        cfg = self.agent.config
        return not (self.failed or cfg.noisy_linear or getattr(cfg, 'graph_update', True) is False or hasattr(rp, 'draw')
                    or self.agent._fused is None)

    def run(self, rp):
        """Returns the loss-kernel outputs of the update it performed, or None (caller runs the eager update)."""
        agent = self.agent
        cfg = agent.config
        self.updates += 1
        if self.updates <= self.WARMUP:
            return None
        opt = agent._fused
        if self.graph is not None and self.signature != opt.hyper_signature():
            self.graph = None       # a hyper-parameter baked into the graph changed (lr schedule): re-capture
        idx = rp.draw_indices()
        if self.graph is None:
            try:
                self.static = rp.gather(idx)
                tr = rp.TransitionCLS(state=self.static['state'], action=self.static['action'], reward=self.static['reward'],
                                      next_state=self.static['next_state'], mask=self.static['mask'])
                opt.enable_graph_mode()
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with _capture(g):
                    out, (net_out, grad) = agent._loss_grad(tr, None)
                    opt.zero_grad()
                    net_out.backward(grad)
                    opt.step(cfg.gradient_clip)
                self.out = out
                self.graph = g
                self.signature = opt.hyper_signature()
            except Exception as e:
                _capture_failed(cfg, "the update", e)
                self.failed = True
                self.graph = None
                opt.graph_mode = False
                return agent._learn(rp.TransitionCLS(**{k: v for k, v in rp.gather(idx).items()
                                                        if k in rp.TransitionCLS._fields}))
        else:
            rp.gather(idx, out=self.static)
        opt.prepare_step()
        self.graph.replay()
        return self.out


class DQNAgent(BaseAgent):
    """DQN_agent.py:48-138."""
    ActorCLS = DQNActor

    def __init__(self, config):
        BaseAgent.__init__(self, config)
        self.config = config
        config.lock = _NullLock()
        self._pre_init()
        self.replay = config.replay_fn()
        self.actor = self.ActorCLS(config)
        self.network = config.network_fn()
        self.target_network = config.network_fn()
        self.target_network.load_state_dict(self.network.state_dict())
        self.optimizer = config.optimizer_fn(self.network.parameters())
        self._fused = FusedOptimizer.adopt(self.optimizer)           # re-homes network params in one flat buffer
        self._target_flat = FlatParams(list(self.target_network.parameters()),
                                       koc=nature_conv_weights(list(self.target_network.parameters())))
        self.actor.set_network(self.network)
        self.total_steps = 0
        self._learner = None            # fused learner (csrc/learner.hip), attached lazily when eligible
        self._host_async = False        # host environment + config.async_actor: the two-stream schedule of _step_host_async
        self._ahead = None
        self._learner_lr = None
        self._pipe = None               # device-resident actor / environment pipeline
        self._fused_checked = False
        self._graphed = _GraphedUpdate(self)   # generic path: whole update as one graph replay (uniform replay)
        self._post_init()
        env = self._device_env()
        if env is not None:
            self._attach_device_pipeline(env)

    # -- fused fast path ---------------------------------------------------------------------------------
    def _inner_replay(self):
        return getattr(self.replay, 'replay', self.replay)

    def _fused_head(self):
        """(head_kind, n_atoms, v_min, v_max) when this agent class / network pair is one csrc/learner.hip implements:
        DQNAgent + VanillaNet, CategoricalDQNAgent + CategoricalNet, QuantileRegressionDQNAgent + QuantileNet, all over
        NatureConvBody (examples.py:55-97, 127-158, 192-222)."""
        from . import learner as L
        from .nets import CategoricalNet, NatureConvBody, QuantileNet, VanillaNet
        cfg, net = self.config, self.network
        if type(getattr(net, 'body', None)) is not NatureConvBody or len(list(net.parameters())) != 10:
            return None
        if type(self) is DQNAgent and type(net) is VanillaNet:
            return (L.HEAD_VANILLA, 0, 0.0, 0.0)
        if type(self) is CategoricalDQNAgent and type(net) is CategoricalNet:
            return (L.HEAD_CATEGORICAL, int(cfg.categorical_n_atoms), float(cfg.categorical_v_min), float(cfg.categorical_v_max))
        if type(self) is QuantileRegressionDQNAgent and type(net) is QuantileNet:
            return (L.HEAD_QUANTILE, int(cfg.num_quantiles), 0.0, 0.0)
        return None

    def _fused_optimizer(self):
        """Keyword arguments of DQNLearner for this agent's torch optimizer (RMSprop or Adam, plain: no momentum, weight
        decay, amsgrad), or None."""
        from . import learner as L
        opt = self.optimizer
        if len(opt.param_groups) != 1:
            return None
        g = opt.param_groups[0]
        if g.get('weight_decay', 0) != 0:
            return None
        if isinstance(opt, torch.optim.RMSprop) and g.get('momentum', 0) == 0:
            return dict(lr=g['lr'], alpha=g['alpha'], eps=g['eps'], centered=bool(g['centered']), optimizer=L.OPT_RMSPROP)
        if type(opt) is torch.optim.Adam and not g.get('amsgrad', False) and not g.get('maximize', False):
            return dict(lr=g['lr'], alpha=0.0, eps=g['eps'], centered=False, optimizer=L.OPT_ADAM, betas=tuple(g['betas']))
        return None

    def _fused_eligible(self):
        """The configurations csrc/learner.hip implements: examples.py:55-97 (dqn_pixel), :127-158 (categorical_dqn_pixel),
        :192-222 (quantile_regression_dqn_pixel) with uniform replay, the first two also with PrioritizedReplay."""
        from .normalizers import ImageNormalizer
        from .replay import PrioritizedReplay, UniformReplay
        cfg = self.config
        if getattr(cfg, 'fused_learner', True) is False or cfg.noisy_linear:
            return False
        head = self._fused_head()
        if head is None or self._fused_optimizer() is None:
            return False
        net = self.network
        rp = self._inner_replay()
        if type(rp) not in (UniformReplay, PrioritizedReplay) or rp._ring is None or rp.history_length != 4:
            return False
        if head[0] == 2 and type(rp) is not UniformReplay:     # QR-DQN + PER is not a valid reference configuration
            return False
        if rp._ring.frame_bytes != 7056 or rp._ring.action_bytes != 8 or rp._state_dtype != torch.uint8:
            return False
        if type(cfg.state_normalizer) is not ImageNormalizer:
            return False
        n_out = cfg.action_dim * max(1, head[1])
        return cfg.batch_size <= 1024 and cfg.action_dim <= 64 and n_out <= 4096 and net.body.conv1.weight.shape[1] == 4

    def _make_learner(self, rp, cu_partition, **extra):
        from .learner import DQNLearner
        cfg = self.config
        head_kind, n_atoms, v_min, v_max = self._fused_head()
        o = self._fused_optimizer()
        self._learner_lr = o['lr']
        return DQNLearner(self.network, self.target_network, rp._ring, cfg.batch_size, cfg.action_dim,
                          cfg.discount ** cfg.n_step, cfg.gradient_clip or 0.0, o['lr'], o['alpha'], o['eps'],
                          centered=o['centered'], double_q=bool(cfg.double_q), u8_coef=cfg.state_normalizer.coef,
                          cu_partition=cu_partition, replay_eps=getattr(cfg, 'replay_eps', 0.01),
                          replay_alpha=getattr(cfg, 'replay_alpha', 0.5), head_kind=head_kind, n_atoms=n_atoms, v_min=v_min,
                          v_max=v_max, optimizer=o['optimizer'], betas=o.get('betas', (0.9, 0.999)), **extra)

    def _attach_fused_learner(self):
        """Host environment (a real emulator, or device_env=False): the update runs in csrc/learner.hip, observations and
        actions cross the host link.  config.async_actor (BaseAgent.py:142-162) is honoured as a two-stream schedule here too:
        the transitions of agent step t+1 (batch-1 forwards on the actor stream + CU partition, from the parameters of
        update t-1, and the emulator steps on the host) are produced while update t runs on the update stream
        (_step_host_async); async_actor=False keeps everything in order on one stream."""
        from . import ops as _ops
        rp = self._inner_replay()
        torch.cuda.synchronize()        # everything issued so far (feeds, initialisation) is on other streams
        want_async = bool(self.config.async_actor) and not hasattr(rp, 'draw')      # (PER: its draw waits for the update anyway)
        self._learner = self._make_learner(rp, cu_partition=want_async)
        self._fused = None              # its flat buffer no longer backs the parameters
        self._target_flat = None
        learner = self._learner
        self._host_async = want_async and bool(learner.variant & _ops.VAR_ACTOR_PARAMS)
        if self._host_async:
            self._ahead = None
            self.actor._fast_q = lambda state: learner.q_host_async(np.asarray(state, dtype=np.uint8)).reshape(1, -1)
        else:
            self.actor._fast_q = lambda state: learner.q_host(np.asarray(state, dtype=np.uint8)).reshape(1, -1)

    # -- device-resident environment + actor (BaseAgent.py:108-182 for async_actor) -----------
    def _device_env(self):
        """The synthetic Atari environment of this agent's actor if the whole rollout -> replay -> update loop can live on
        the device: the dqn_pixel configuration (see _fused_eligible), UniformReplay, ONE SyntheticAtari environment,
        sign / identity reward normaliser.  `config.device_env = False` keeps the environment on the host."""
        from .envs import SyntheticAtari
        from .normalizers import RescaleNormalizer, SignNormalizer
        from .replay import PrioritizedReplay, UniformReplay
        cfg = self.config
        if Config.DEVICE.type != 'cuda' or getattr(cfg, 'device_env', True) is False:
            return None
        task = getattr(self.actor, '_task', None)
        envs = getattr(getattr(task, 'env', None), 'envs', None)
        if not envs or len(envs) != 1 or type(envs[0]) is not SyntheticAtari:
            return None
        env = envs[0]
        rn = cfg.reward_normalizer
        if not (type(rn) is SignNormalizer or (type(rn) is RescaleNormalizer and rn.coef == 1.0)):
            return None
        rp = self._inner_replay()
        if type(rp) not in (UniformReplay, PrioritizedReplay) or rp.history_length != 4 or env.history != 4 or env.n_actions != cfg.action_dim:
            return None
        if env.frames is not None:        # somebody already stepped it on the host: leave it there
            return None
        rp.device_ring()                  # the ring feed() would create from the first transition
        return env if self._fused_eligible() else None

    def _attach_device_pipeline(self, env):
        """config.async_actor (BaseAgent.py:142-162) is honoured as the two-stream pipeline of csrc/learner.hip: the actor
        (forward, epsilon-greedy, environment step, replay feed) runs one agent step ahead of the learner on its own
        stream and CU partition; async_actor=False runs the same kernels in order."""
        from .learner import DeviceActorPipeline, SyntheticEpisodeStream
        from .replay import PrioritizedReplay
        cfg = self.config
        rp = self._inner_replay()
        torch.cuda.synchronize()
        async_actor = bool(cfg.async_actor)
        self._learner = self._make_learner(rp, cu_partition=async_actor, env_seed=env.seed, env_done_period=env.done_period)
        self._fused = None
        self._target_flat = None
        self._fused_checked = True
        actor = self.actor

        def epsilon():                     # DQN_agent.py:34-39, the actor's own step counter
            if actor._total_steps < cfg.exploration_steps:
                eps = 1
            else:
                eps = cfg.random_action_prob()
            actor._total_steps += 1
            return eps

        stream = SyntheticEpisodeStream(env.seed, env.counter, env.done_period, env.history)
        self._pipe = DeviceActorPipeline(self._learner, rp, stream, cfg.action_dim, cfg.sgd_update_frequency, epsilon,
                                         async_actor, actor_seed=int(np.random.randint(1 << 31)) if async_actor else None,
                                         beta_fn=getattr(cfg, 'replay_beta', None))

    def _step_device(self):
        cfg = self.config

        def account(infos):               # DQN_agent.py:103-113 for transitions that never leave the device
            for reward, done, info in infos:
                self.record_online_return((info,))
                self.total_steps += 1
            return self.total_steps > cfg.exploration_steps

        pipe = self._pipe
        if pipe.async_actor and not pipe.per:
            # the uniform two-stream pipeline names its stream in every call (learner._sp): no torch stream context -- entering and
            # leaving one is ~15 us of host time per agent step, a sixth of the device-side step
            pipe.step(account)
            if self.total_steps / cfg.sgd_update_frequency % cfg.target_network_update_freq == 0:
                self.sync_target()
            return
        with torch.cuda.stream(self._learner.stream):
            pipe.step(account)
            if self.total_steps / cfg.sgd_update_frequency % cfg.target_network_update_freq == 0:
                self.sync_target()

    def _pre_init(self):
        pass

    def _post_init(self):
        pass

    def save(self, filename):
        if self._learner is not None:     # updates run asynchronously on the learner's streams
            self._learner.synchronize()
        BaseAgent.save(self, filename)

    def load(self, filename):
        if self._learner is not None:
            self._learner.synchronize()
        BaseAgent.load(self, filename)
        if self._learner is not None:     # parameters changed behind the learner: actor copies are stale
            torch.cuda.synchronize()
            self._learner.invalidate_actor_copy()
        self._ahead = None                # (host-emulator async actor) transitions produced ahead acted on the old parameters

    # -- true resume (the reference's save() keeps weights + normaliser only, BaseAgent.py:24-33) ------
    def save_full(self, filename):
        """Everything a bit-exact continuation needs, next to the unchanged `.model` / `.stats` files: parameters, target
        and optimizer state in their device layout, the learner's pipeline state (actor parameter copies, blocks generated
        ahead, the pending observation), the replay ring (sharded) and its cursor / priority tree, schedules, step counters
        and every random stream (numpy global, python `random`, torch CPU / device, the async actor's own RandomState).
        40 steps + save_full + a fresh process + load_full + 20 steps == 60 uninterrupted steps, bit for bit
        (tests/test_gpu_resume.py).  Device-resident pipeline (the benchmarked dqn_pixel-family configurations) only."""
        import pickle
        import random as pyrandom
        if self._pipe is None or self._learner is None:
            raise NotImplementedError("save_full: this agent runs a host environment / the generic update path; only the "
                                      "device-resident pipeline (config.device_env, synthetic Atari) has a full resume")
        self.save(filename)
        cfg = self.config
        sched = {}
        for name in ("random_action_prob", "replay_beta"):
            sc = getattr(cfg, name, None)
            if isinstance(sc, LinearSchedule):
                sched[name] = dict(current=sc.current, inc=sc.inc, end=sc.end)
        state = dict(learner=self._learner.resume_state(), pipe=self._pipe.state_dict(), total_steps=self.total_steps,
                     actor_total_steps=self.actor._total_steps, schedules=sched, learner_lr=self._learner_lr,
                     shape=self._resume_shape(),
                     np_random=np.random.get_state(), py_random=pyrandom.getstate(), torch_cpu=torch.get_rng_state(),
                     torch_cuda=torch.cuda.get_rng_state(Config.DEVICE), agent=type(self).__name__)
        self._inner_replay().save_full(filename, ahead=2 * self._pipe.n_env)   # the actor is one agent step ahead of the cursor
        with open(filename + '.resume', 'wb') as f:
            pickle.dump(state, f)

    def _resume_shape(self):
        cfg = self.config
        return dict(n_env=int(self._pipe.n_env), batch_size=int(cfg.batch_size), history_length=int(cfg.history_length),
                    n_step=int(cfg.n_step), sgd_update_frequency=int(cfg.sgd_update_frequency), lr=float(self._learner_lr),
                    async_actor=bool(self._pipe.async_actor), per=bool(self._pipe.per))

    def load_full(self, filename):
        """Into a freshly constructed agent of the same Config (same network / replay / pipeline shape), before its first step."""
        import pickle
        import random as pyrandom
        if self._pipe is None or self._learner is None:
            raise NotImplementedError("load_full needs the device-resident pipeline (see save_full)")
        with open(filename + '.resume', 'rb') as f:
            state = pickle.load(f)
        if state["agent"] != type(self).__name__:
            raise DraError("%s.resume was written by a %s" % (filename, state["agent"]))
        # the Config fields that shape the pipeline state being restored (pending blocks, slots written ahead, minibatch
        # buffers, the optimizer's step size): a resume under a different Config must not continue silently
        mine, theirs = self._resume_shape(), state.get("shape")
        if theirs is not None and theirs != mine:
            diff = {k: (theirs.get(k), mine.get(k)) for k in set(mine) | set(theirs) if theirs.get(k) != mine.get(k)}
            raise DraError("%s.resume was written under a different Config (checkpoint, this agent): %s" % (filename, diff))
        with open(filename + '.stats', 'rb') as f:
            self.config.state_normalizer.load_state_dict(pickle.load(f))
        self._inner_replay().load_full(filename)
        self._learner.load_resume_state(state["learner"])
        self._pipe.load_state_dict(state["pipe"])
        self.total_steps, self.actor._total_steps = state["total_steps"], state["actor_total_steps"]
        for name, st in state["schedules"].items():
            sc = getattr(self.config, name)
            sc.current, sc.inc, sc.end = st["current"], st["inc"], st["end"]
        np.random.set_state(state["np_random"])
        pyrandom.setstate(state["py_random"])
        torch.set_rng_state(state["torch_cpu"])
        torch.cuda.set_rng_state(state["torch_cuda"], Config.DEVICE)

    def sync_host(self):
        """Between two step() calls: everything issued has been accounted for on the host and python's `random` stands where the
        reference's would (with PrioritizedReplay drawn on the device the generator runs ahead of the kernels otherwise:
        replay.DeviceDraw).  save_full(), eval_episodes() and close() call it."""
        if getattr(self, '_pipe', None) is not None:
            self._pipe.sync_host()

    def eval_episodes(self):
        self.sync_host()
        return super().eval_episodes()

    def close(self):
        if getattr(self, '_learner', None) is not None:
            self.sync_host()
            self._learner.synchronize()
            self._learner.close()
            self._learner = None
        close_obj(self.replay)
        close_obj(self.actor)

    def eval_step(self, state):
        self.config.state_normalizer.set_read_only()
        state = self.config.state_normalizer(state)
        if self._learner is not None:
            self._learner.synchronize()
        with torch.no_grad():
            q = self.network(state)['q']
        action = to_np(q.argmax(-1))
        self.config.state_normalizer.unset_read_only()
        return action

    # -- reference-shaped hooks, kept for callers that use them directly -------------------------------
    def reduce_loss(self, loss):
        return ops.weighted_mean(loss.pow(2).mul(0.5).contiguous())

    def compute_loss(self, transitions):
        """DQN_agent.py:81-99: returns the TD-error vector (no autograd graph; step() uses _loss_grad)."""
        out, _ = self._loss_grad(transitions, per=None)
        return out['delta']

    @staticmethod
    def _f32(x):
        return x if (isinstance(x, torch.Tensor) and x.dtype == torch.float32) else tensor(to_np(x) if isinstance(x, torch.Tensor) else x)

    def _batch_scalars(self, transitions):
        reward = transitions.reward
        mask = transitions.mask
        reward = reward.float() if isinstance(reward, torch.Tensor) else tensor(reward)
        mask = mask.float() if isinstance(mask, torch.Tensor) else tensor(mask)
        action = transitions.action
        if not isinstance(action, torch.Tensor):
            action = torch.from_numpy(np.asarray(action, dtype=np.int64)).to(reward.device)
        elif action.dtype not in (torch.int64, torch.float32):
            action = action.long()
        return action.contiguous(), reward.contiguous(), mask.contiguous()

    def _per_args(self, transitions):
        config = self.config
        sp = transitions.sampling_prob
        sp = sp.float() if isinstance(sp, torch.Tensor) else tensor(sp)
        return dict(sampling_prob=sp.contiguous(), beta=config.replay_beta(), replay_eps=config.replay_eps,
                    replay_alpha=config.replay_alpha)

    def _loss_grad(self, transitions, per):
        """Forward passes + the fused TD kernel.  Returns (kernel outputs, (net_output, grad))."""
        config = self.config
        states = config.state_normalizer(transitions.state)
        next_states = config.state_normalizer(transitions.next_state)
        with torch.no_grad():
            q_next = self.target_network(next_states)['q']
            q_next_online = self.network(next_states)['q'] if config.double_q else None
        action, reward, mask = self._batch_scalars(transitions)
        q = self.network(states)['q']
        out = ops.td_loss(q.detach(), q_next, action, reward, mask, config.discount ** config.n_step,
                          q_next_online=q_next_online, **(per or {}))
        return out, (q, out['dq'])

    def _learn(self, transitions):
        config = self.config
        is_per = isinstance(transitions, PrioritizedTransition)
        out, (net_out, grad) = self._loss_grad(transitions, self._per_args(transitions) if is_per else None)
        if is_per:
            idxs = transitions.idx
            idxs = to_np(idxs.long()) if isinstance(idxs, torch.Tensor) else np.asarray(idxs, dtype=np.int64)
            self.replay.update_priorities(zip(idxs, to_np(out['prio'])))
        self._fused.zero_grad()
        net_out.backward(grad)
        with config.lock:
            self._fused.step(config.gradient_clip)
        return out

    def step(self):
        if self._learner is not None and self.optimizer.param_groups[0]['lr'] != self._learner_lr:
            raise NotImplementedError("the fused DQN learner bakes the learning rate into its captured graphs; a scheduled "
                                      "lr needs config.fused_learner = False (generic path)")
        if self._pipe is not None:
            return self._step_device()
        if self._learner is not None:   # ring feeds, actor forwards and updates share the learner's stream
            with torch.cuda.stream(self._learner.stream):
                if self._host_async:
                    return self._step_host_async()
                return self._step()
        self._step()
        if not self._fused_checked and self._inner_replay().size() > 0:
            self._fused_checked = True
            if self._fused_eligible():
                self._attach_fused_learner()

    def _step(self):
        config = self.config
        transitions = self.actor.step()
        for states, actions, rewards, next_states, dones, info in transitions:
            self.record_online_return(info)
            self.total_steps += 1
            self.replay.feed(dict(
                state=np.array([s[-1] if isinstance(s, LazyFrames) else s for s in states]),
                action=actions,
                reward=[config.reward_normalizer(r) for r in rewards],
                mask=1 - np.asarray(dones, dtype=np.int32),
            ))
        if self.total_steps > config.exploration_steps:
            if self._learner is not None:
                rp = self._inner_replay()
                if hasattr(rp, 'draw'):
                    # PER (DQN_agent.py:120-127): tree descent on device with host-drawn uniforms, one D2H of the
                    # indices (validity / padding stay on the host, draw for draw); the update applies the IS
                    # weights and emits the new priorities, which the tree takes without a second host round trip
                    tree_idx, prob, data_idx = rp.draw()
                    self._learner.update(data_idx, use_graph=True, sampling_prob=prob, beta=config.replay_beta())
                    rp.commit_device(tree_idx, self._learner.prio)      # replay.py:193-196, priorities stay on the device
                else:
                    # same index draws as replay.sample() (replay.py:92-103); gather + update are one graph replay
                    self._learner.update(rp.draw_indices(), use_graph=True)
            elif self._graphed.usable(self._inner_replay()) and self._graphed.run(self._inner_replay()) is not None:
                pass
            else:
                transitions = self.replay.sample()
                if config.noisy_linear:
                    self.target_network.reset_noise()
                    self.network.reset_noise()
                self._learn(transitions)
        if self.total_steps / config.sgd_update_frequency % config.target_network_update_freq == 0:
            self.sync_target()

    def _step_host_async(self):
        """DQN_agent.py:101-138 with async_actor=True over a HOST environment.  The reference's actor process produces the
        transitions of the next agent step while the learner trains (BaseAgent.py:142-162: step() returns the cached
        transitions and immediately asks for the next ones).  Here: feed the transitions produced during the previous call,
        enqueue update t (non-blocking, update stream), then produce the transitions of step t+1 -- forward passes on the
        actor stream from the parameter copy of update t-1 (csrc/learner.hip dra_dqn_learner_q_host_async), emulator steps
        on the host -- while the update runs."""
        config = self.config
        transitions = self._ahead if self._ahead is not None else self.actor.step()
        self._ahead = None
        rp = self._inner_replay()
        for states, actions, rewards, next_states, dones, info in transitions:
            self.record_online_return(info)
            self.total_steps += 1
            self.replay.feed(dict(
                state=np.array([s[-1] if isinstance(s, LazyFrames) else s for s in states]),
                action=actions,
                reward=[config.reward_normalizer(r) for r in rewards],
                mask=1 - np.asarray(dones, dtype=np.int32),
            ))
        if self.total_steps > config.exploration_steps:
            # the reference's rejection loop (replay.py:92-103) drawn in blocks: same np.random stream, same indices
            # (learner.draw_uniform_indices; tests/test_host_utils_vs_reference.py), a fifth of the host time
            from .learner import draw_uniform_indices
            idx = draw_uniform_indices(rp.size(), rp.pos, rp.batch_size, rp.history_length, rp.n_step)
            self._learner.update_async(idx, use_graph=True)
        if self.total_steps / config.sgd_update_frequency % config.target_network_update_freq == 0:
            self.sync_target()
        self._ahead = self.actor.step()          # overlaps the update just enqueued

    def sync_target(self):
        """DQN_agent.py:136-138 as one device-to-device copy of the flat parameter buffer."""
        if self._learner is not None:
            self._learner.sync_target()
            return
        ops.copy_f32(self._target_flat.flat, self._fused.flat.flat)
        for tb, b in zip(self.target_network.buffers(), self.network.buffers()):
            tb.copy_(b)


class CategoricalDQNActor(DQNActor):
    """CategoricalDQN_agent.py:14-24."""

    def _set_up(self):
        self.config.atoms = tensor(self.config.atoms)

    def compute_q(self, prediction):
        return to_np((prediction['prob'] * self.config.atoms).sum(-1))

    def q_device(self, prediction):
        return (prediction['prob'] * self.config.atoms).sum(-1)


class CategoricalDQNAgent(DQNAgent):
    """CategoricalDQN_agent.py:27-89."""
    ActorCLS = CategoricalDQNActor

    def _pre_init(self):
        config = self.config
        config.atoms = np.linspace(config.categorical_v_min, config.categorical_v_max, config.categorical_n_atoms)

    def _post_init(self):
        config = self.config
        self.batch_indices = range_tensor(config.batch_size)
        self.atoms = tensor(config.atoms)
        self.delta_atom = (config.categorical_v_max - config.categorical_v_min) / float(config.categorical_n_atoms - 1)

    def eval_step(self, state):
        self.config.state_normalizer.set_read_only()
        state = self.config.state_normalizer(state)
        if self._learner is not None:
            self._learner.synchronize()
        with torch.no_grad():
            prediction = self.network(state)
        action = to_np((prediction['prob'] * self.atoms).sum(-1).argmax(-1))
        self.config.state_normalizer.unset_read_only()
        return action

    def compute_loss(self, transitions):
        return self._loss_grad(transitions, per=None)[0]['kl']

    def reduce_loss(self, loss):
        return ops.weighted_mean(loss.contiguous())

    def _loss_grad(self, transitions, per):
        config = self.config
        states = config.state_normalizer(transitions.state)
        next_states = config.state_normalizer(transitions.next_state)
        with torch.no_grad():
            logits_t = self.target_network(next_states)['logits']
            logits_o = self.network(next_states)['logits'] if config.double_q else None
        action, reward, mask = self._batch_scalars(transitions)
        logits = self.network(states)['logits']
        weights = prio_w = None
        if per is not None:
            _, weights = ops.per_weights(None, per['sampling_prob'], per['beta'], per['replay_eps'], per['replay_alpha'])
        out = ops.c51_loss(logits.detach().contiguous(), logits_t.contiguous(), action, reward, mask,
                           config.discount ** config.n_step, self.atoms, config.categorical_v_min,
                           config.categorical_v_max, logits_next_online=logits_o, weights=weights)
        if per is not None:
            out['prio'], _ = ops.per_weights(out['kl'], per['sampling_prob'], per['beta'], per['replay_eps'],
                                             per['replay_alpha'])
            out['weights'] = weights
        return out, (logits, out['dlogits'])


class QuantileRegressionDQNActor(DQNActor):
    """QuantileRegressionDQN_agent.py:13-20."""

    def compute_q(self, prediction):
        return to_np(prediction['quantile'].mean(-1))

    def q_device(self, prediction):
        return prediction['quantile'].mean(-1)


class QuantileRegressionDQNAgent(DQNAgent):
    """QuantileRegressionDQN_agent.py:23-77 (UniformReplay only: the PER branch mis-shapes in the
    reference)."""
    ActorCLS = QuantileRegressionDQNActor

    def _post_init(self):
        config = self.config
        self.batch_indices = range_tensor(config.batch_size)
        self.quantile_weight = 1.0 / config.num_quantiles
        self.cumulative_density = tensor((2 * np.arange(config.num_quantiles) + 1) / (2.0 * config.num_quantiles)).view(1, -1)

    def eval_step(self, state):
        self.config.state_normalizer.set_read_only()
        state = self.config.state_normalizer(state)
        if self._learner is not None:
            self._learner.synchronize()
        with torch.no_grad():
            q = self.network(state)['quantile'].mean(-1)
        action = np.argmax(to_np(q).flatten())
        self.config.state_normalizer.unset_read_only()
        return [action]

    def compute_loss(self, transitions):
        return self._loss_grad(transitions, per=None)[0]['loss_vec']

    def reduce_loss(self, loss):
        return ops.weighted_mean(loss.contiguous())

    def _loss_grad(self, transitions, per):
        if per is not None:
            raise NotImplementedError("QR-DQN with PrioritizedReplay is not a valid reference configuration")
        config = self.config
        states = config.state_normalizer(transitions.state)
        next_states = config.state_normalizer(transitions.next_state)
        with torch.no_grad():
            theta_t = self.target_network(next_states)['quantile']
        action, reward, mask = self._batch_scalars(transitions)
        theta = self.network(states)['quantile']
        out = ops.qr_loss(theta.detach().contiguous(), theta_t.contiguous(), action, reward, mask,
                          config.discount ** config.n_step)
        return out, (theta, out['dtheta'])


# ==================================================================================================== on-policy
def _rollout_scan(storage, config, bootstrap_v):
    """PPO_agent.py:51-61 / A2C_agent.py:43-53 as one dra_gae launch.  Fills storage.advantage / .ret
    with per-step [N,1] tensors (views of the [T,N,1] results)."""
    t_len = config.rollout_length
    reward = torch.stack(storage.reward[:t_len]).contiguous()
    mask = torch.stack(storage.mask[:t_len]).contiguous()
    value = torch.stack([v.detach() for v in storage.v[:t_len]] + [bootstrap_v.detach()]).contiguous()
    adv, ret = ops.gae(reward, mask, value, config.discount, config.gae_tau, config.use_gae)
    for i in range(t_len):
        storage.advantage[i] = adv[i]
        storage.ret[i] = ret[i]
    return adv, ret


def _begin_rollout(agent, n_env):
    """Announces a rollout of rollout_length + 1 no-grad forwards over n_env environments to a categorical actor-critic
    (nets.RolloutSlots: one uniform draw for the whole rollout, results written into persistent rows) -> the slots, or None
    (another network family, or a rank-invariant sampler is installed)."""
    slots = getattr(agent.network, 'rollout_slots', None)
    if slots is None or getattr(agent.network, 'sampler', None) is not None or Config.DEVICE.type != 'cuda':
        return None
    slots.begin(int(agent.config.rollout_length) + 1, int(n_env))
    return slots


class _PixelRollout:
    """The no-grad forwards of one A2C / PPO rollout over CategoricalActorCriticNet(NatureConvBody) and device-resident synthetic
    Atari environments as FOUR launches per step instead of five (csrc/conv_v2.hip: rollout_conv1_heads_kernel): [conv1 of step t
    | fc4's finish + policy head of step t - 1], conv2, conv3, fc4's 28 K-slice partial sums -- every launch of such a step is a few dozen workgroups at its latency floor,
    so the step costs what its launches cost.  The head of step t - 1 can share conv1's launch because the planned observations
    do not depend on the actions (DeviceAtariVec.states_all).  Same device functions as the module path: the rollout's actions,
    log-probabilities, entropies and values are bit-identical (tests/test_gpu_agents.py).  a2c_pixel 209 k -> 224 k, ppo_pixel
    109 k -> 118 k env-steps/s (profiles/r05x_bench_agents_c3fc4_ab.jsonl; conv3 + fc4 as one launch was measured too and lost)."""

    def __init__(self, agent):
        self.agent = agent
        self.bufs = None

    def eligible(self):
        from .device_env import DeviceAtariVec
        from .nets import CategoricalActorCriticNet, Conv2d, DummyBody, Linear, NatureConvBody
        a = self.agent
        net, cfg = a.network, a.config
        if getattr(cfg, 'fused_rollout', True) is False or Config.DEVICE.type != 'cuda' or not isinstance(a.task, DeviceAtariVec):
            return False
        if type(net) is not CategoricalActorCriticNet or getattr(net, 'sampler', None) is not None or not getattr(net, 'rollout_fc4_slices', True):
            return False
        body = net.phi_body
        if type(body) is not NatureConvBody or type(net.actor_body) is not DummyBody or type(net.critic_body) is not DummyBody:
            return False
        convs = [body.conv1, body.conv2, body.conv3]
        if not all(type(c) is Conv2d and c.bias is not None and c.weight.permute(1, 2, 3, 0).is_contiguous() for c in convs):
            return False
        if getattr(body.conv1, 'u8_coef', None) is None or a.task.history != 4:
            return False
        heads = [body.fc4, net.fc_action, net.fc_critic]
        if not all(type(m) is Linear and m.bias is not None and m.weight.is_contiguous() for m in heads):
            return False
        return (body.fc4.fused_act == "relu" and tuple(body.fc4.weight.shape) == (512, 3136) and net.fc_action.fused_act is None
                and net.fc_critic.fused_act is None and net.fc_action.weight.shape[0] <= 64 and net.fc_critic.weight.shape[0] == 1
                and a.task.num_envs <= 32)

    def run(self, frames, slots):
        """frames: uint8 [T + 1, N, 4, 84, 84]; fills rows 0..T of `slots` (action, log_pi_a, entropy, v) from slots.uniform."""
        import ctypes
        from ._lib import lib, stream_ptr
        a = self.agent
        net = a.network
        body = net.phi_body
        rows, n = int(frames.shape[0]), int(frames.shape[1])
        dev = frames.device
        if self.bufs is None or self.bufs['n'] != n or self.bufs['rows'] < rows:
            f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            # (the activations of every step are kept: A2C's update backpropagates through them, as the reference does through the
            # rollout's own forward graph -- config.reuse_rollout_activations)
            self.bufs = dict(n=n, rows=rows, y1=f(rows, n, 32, 20, 20), y2=f(rows, n, 64, 9, 9), y3=f(rows, n, 64, 7, 7), slabs=f(28, n, 512),
                             phi=f(rows, n, 512))
            one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
            # (per-step pointer arguments built once: the rollout loop of PPO's 129 steps is host-side eager code)
            self.bufs['args'] = [(ctypes.c_void_p(self.bufs['y1'][t].data_ptr()), one(self.bufs['y1'][t]), one(self.bufs['y2'][t]),
                                  one(self.bufs['y3'][t]), ctypes.c_void_p(self.bufs['phi'][t].data_ptr())) for t in range(rows)]
        b = self.bufs
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        arr = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
        st = stream_ptr()
        w1, w2, w3 = body.conv1.weight, body.conv2.weight, body.conv3.weight       # [(c, kh, kw)][oc] storage (FlatParams)
        wa, ba, wv, bv = net.fc_action.weight, net.fc_action.bias, net.fc_critic.weight, net.fc_critic.bias
        n_act = int(wa.shape[0])
        coef = float(body.conv1.u8_coef)
        wt2, b2 = arr(w2), arr(body.conv2.bias)
        wt3, b3, w4 = arr(w3), arr(body.conv3.bias), arr(body.fc4.weight)
        b4, slabs = body.fc4.bias, b['slabs']
        for t in range(rows):
            prev = t > 0
            y1p, x2, y2, y3, _ = b['args'][t]
            # fc4 of step t - 1 left its 28 K-slice partial sums: the head that rides in this launch folds them (bias, ReLU) first
            # (and leaves the folded features of step t - 1 in phi[t - 1]: fc4's forward output, for A2C's update)
            lib.dra_rollout_conv1_heads_phi(p(frames[t]), p(w1), p(body.conv1.bias), y1p, n, coef,
                                            p(slabs) if prev else None, p(b4) if prev else None, p(wa), p(ba), p(wv), p(bv),
                                            p(slots.uniform[t - 1]) if prev else None, n_act,
                                            p(slots.action[t - 1]) if prev else None, p(slots.log_pi_a[t - 1]) if prev else None,
                                            p(slots.entropy[t - 1]) if prev else None, p(slots.v[t - 1]) if prev else None,
                                            b['args'][t - 1][4] if prev else None, st)
            lib.dra_conv_fwd_koc(2, 1, x2, wt2, b2, y2, n, 0, 1.0, ops.ACT["relu"], st)
            lib.dra_conv_fwd_koc(3, 1, y2, wt3, b3, y3, n, 0, 1.0, ops.ACT["relu"], st)
            lib.dra_linear_fwd_slabs_one(1, y3, w4, n, 3136, 512, 28, p(slabs), st)
        t = rows - 1
        lib.dra_policy_heads_sample_fold28(p(slabs), p(b4), p(wa), p(ba), p(wv), p(bv), p(slots.uniform[t]), n, n_act,
                                           p(slots.action[t]), p(slots.log_pi_a[t]), p(slots.entropy[t]), p(slots.v[t]), st)
        slots.end()


def _install_sampler(agent):
    """Data-parallel agents (and config.dp_invariant_sampling) sample actions from per-step noise that is the same
    however the environments are spread over ranks (dist.DataParallel.uniforms); everything else keeps the reference's
    dist.sample() on torch's global generator."""
    dp = agent.dp
    net = agent.network
    if not dp.invariant_sampling or not hasattr(net, 'fc_action') or not hasattr(agent.task, 'action_space'):
        return
    from .envs import Discrete
    if not isinstance(agent.task.action_space, Discrete):
        from .nets import GaussianActorCriticNet
        if not isinstance(net, GaussianActorCriticNet):
            raise NotImplementedError("rank-invariant sampling is implemented for categorical and Gaussian policies")
        from . import ppo_mlp

        def gauss(mean, scale):     # mean + scale * hashed normal of (noise seed, sampler step, GLOBAL environment, dimension)
            if dp.step_dev is None:
                dp.step_dev = torch.zeros(1, dtype=torch.int64, device=mean.device)
            return ppo_mlp.gauss_sample(mean, scale, dp.noise_seed, dp.step_dev, dp.global_workers, dp.lo)
        net.sampler = gauss
        return
    # one kernel per sample; the noise stream's position is a device counter the kernel advances: graph-capturable
    net.sampler = lambda logits: dp.sample(logits.detach())


class _OnPolicyGraph:
    """Replays an on-policy agent's device-side work of one rollout (A2CAgent._rollout_compute: T forwards with action
    sampling, the return scan, loss, backward, clip, optimizer) from ONE captured hipGraph after two eager rollouts.  The
    rollout plan lives in persistent device buffers (DeviceAtariVec.plan), so every kernel argument is constant across
    rollouts; torch's graph-safe Philox bookkeeping continues the sampling generator exactly as the eager kernels would.
    Data parallel (dist.DataParallel): the captured region ends BEFORE the gradient exchange -- run(plan, compute, tail)
    replays [rollout + loss + backward] and then calls tail() eagerly (all-reduce of the flat gradient over RCCL / gloo,
    clip + optimizer step: three launches), so that ranks x environments-per-rank runs at the single-process graph speed.
    Not used with grad hooks or config.graph_update off."""
    WARMUP = 2

    def __init__(self, agent, optimizer_inside=True):
        self.agent = agent
        self.optimizer_inside = optimizer_inside      # the captured region ends with agent._fused.step()
        self.calls = 0
        self.graph = None
        self.out = None
        self.key = None
        self.failed = False

    def usable(self):
        a = self.agent
        cfg = a.config
        return not (self.failed or getattr(cfg, 'graph_update', True) is False or a.grad_hook is not None)

    def run(self, plan, compute, tail=None):
        """compute(plan): the capturable device work; tail(out) (optional): what must stay eager after it (the data-parallel
        exchange and the optimizer step behind it) -- called after the eager compute as well as after a replay."""
        a = self.agent
        self.calls += 1
        if not self.usable() or self.calls <= self.WARMUP:
            out = compute(plan)
            return out if tail is None else tail(out)
        opt = a._fused if (self.optimizer_inside and tail is None) else None
        key = (plan.counters.data_ptr(), plan.t_len, opt.hyper_signature() if opt is not None else None)
        if self.graph is not None and key != self.key:
            self.graph = None
        if self.graph is None:
            validate = torch.distributions.Distribution._validate_args
            try:
                torch.distributions.Distribution.set_default_validate_args(False)   # its checks synchronise with the host
                if opt is not None:
                    opt.enable_graph_mode()
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                steps = a._rollout_step
                with _capture(g):
                    self.out = compute(plan)
                a._rollout_step = steps       # the capture pass only recorded the work
                self.graph, self.key = g, key
            except Exception as e:
                _capture_failed(a.config, "the rollout", e)
                self.failed = True
                self.graph = None
                if opt is not None:
                    opt.graph_mode = False
                out = compute(plan)
                return out if tail is None else tail(out)
            finally:
                torch.distributions.Distribution.set_default_validate_args(validate)
        if opt is not None:
            opt.prepare_step()
        self.graph.replay()
        a._rollout_step += plan.t_len + (1 if hasattr(a, '_act') else 0)
        return self.out if tail is None else tail(self.out)


def _dp_sync_start(dp, network, *fused):
    """Data parallel: every rank starts from rank 0's parameters, module buffers and optimizer state (only gradients are
    exchanged afterwards, so equal starts stay equal)."""
    if not dp.active:
        return
    tensors = []
    for f in fused:
        tensors += [f.flat.flat, f.state1, f.state2]
    seen = {t.data_ptr() for t in tensors}
    for t in list(network.parameters()) + list(network.buffers()):
        # parameters re-homed in a flat buffer are views of it; anything else (buffers, stray parameters) goes by itself
        if not any(f.flat.flat.data_ptr() <= t.data_ptr() < f.flat.flat.data_ptr() + 4 * f.flat.flat.numel() for f in fused) \
                and t.data_ptr() not in seen:
            tensors.append(t)
            seen.add(t.data_ptr())
    dp.sync_state(*tensors)


def _dp_plan_exchange(dp, network, fused):
    """Data parallel: the gradient exchange in two segments, [fc4 + heads] -- complete as soon as fc4's backward has run -- going
    out while the convolutions are still being differentiated (dist.DataParallel.plan_split).  NatureConvBody networks only
    (6.4 of the 6.75 MB sit behind the split); anything else keeps the single exchange."""
    if not dp.active:
        return
    fc4 = getattr(getattr(network, 'phi_body', None), 'fc4', None)
    w = getattr(fc4, 'weight', None)
    if w is None or not any(p is w for p in fused.flat.params):
        return
    start = fused.flat.offset_of(w)
    dp.plan_split(fused.flat, [p for p, o in zip(fused.flat.params, fused.flat.offsets) if o >= start])


def _device_state_fn(agent):
    """How a device-resident rollout hands observations to the network.  ImageNormalizer over NatureConvBody (a2c_pixel /
    ppo_pixel, examples.py:361-381, 525-550): conv1's kernels take the uint8 frames themselves and normalise while staging --
    f32(f64(v) * coef), the normaliser's own table values, bit for bit (tests/test_gpu_kernels.py: u8 and f32 inputs agree to the
    bit) -- so the normaliser's launch per rollout step disappears and the rollout is stored, gathered and re-read as uint8 (a
    quarter of the bytes).  Anything else goes through config.state_normalizer as the reference does."""
    from .nets import Conv2d, NatureConvBody
    from .normalizers import RescaleNormalizer
    cfg = agent.config
    norm = cfg.state_normalizer
    body = getattr(agent.network, 'phi_body', None)
    conv1 = getattr(body, 'conv1', None)
    if (Config.DEVICE.type == 'cuda' and isinstance(norm, RescaleNormalizer) and type(body) is NatureConvBody
            and isinstance(conv1, Conv2d) and conv1.weight.permute(1, 2, 3, 0).is_contiguous()
            and getattr(cfg, 'device_u8_states', True)):
        conv1.u8_coef = float(norm.coef)

        def passthrough(x):
            if isinstance(x, torch.Tensor) and x.dtype == torch.uint8 and x.is_cuda:
                return x
            return norm(x)
        return passthrough
    return norm


class A2CAgent(BaseAgent):
    """A2C_agent.py:12-64."""

    def __init__(self, config):
        BaseAgent.__init__(self, config)
        self.config = config
        from .dist import DataParallel
        self.dp = DataParallel(config)           # shards config.num_workers over the ranks BEFORE task_fn builds the envs
        self.task = config.task_fn()
        self.network = config.network_fn()
        self.optimizer = config.optimizer_fn(self.network.parameters())
        self._fused = FusedOptimizer.adopt(self.optimizer)
        _dp_sync_start(self.dp, self.network, self._fused)
        _dp_plan_exchange(self.dp, self.network, self._fused)
        if self.dp.active:
            self.dp.set_weight(1.0 / self.dp.world)
        self.total_steps = 0
        from .device_env import DeviceAtariVec
        if DeviceAtariVec.eligible(self.task, config):      # synthetic Atari emulators: the environments live on the device
            self.task = DeviceAtariVec(self.task)
        self.states = self.task.reset()
        self.grad_hook = None  # optional extra hook on the flat gradient before the optimiser step
        self._rollout_step = 0
        self._dev_graph = _OnPolicyGraph(self)
        self._dev_state = _device_state_fn(self)
        _install_sampler(self)
        self._pixel_rollout = _PixelRollout(self)
        self.network.fuse_fc4_head = bool(getattr(config, 'fuse_fc4_head', True))
        self.network.rollout_fc4_slices = bool(getattr(config, 'rollout_fc4_slices', True))

    def close(self):
        close_obj(self.task)
        self.dp.close()

    def _step_device(self):
        """step() over device-resident environments: the host lays the rollout out (counters, rewards, terminals of every
        environment: the synthetic environment is a pure function of its frame counter), uploads it with one copy, and
        everything else -- T x [observations, forward, action sample], bootstrap forward, return scan, loss, backward, clip,
        optimizer -- is enqueued without a host round trip and, after two eager rollouts, replayed as ONE captured graph."""
        config = self.config
        plan = self.task.plan(config.rollout_length, config.reward_normalizer)
        for t in range(config.rollout_length):
            self.record_online_return(plan.infos[t])
            self.total_steps += self.dp.global_workers
        if self.dp.active:      # graph = [rollout + loss + backward]; the exchange and the step behind it stay eager
            out4 = self._dev_graph.run(plan, lambda pl: self._rollout_compute(pl, apply=False), tail=self._learn_apply)
        else:
            out4 = self._dev_graph.run(plan, self._rollout_compute)
        self.last_loss = out4

    def _rollout_compute(self, plan, apply=True):
        config = self.config
        t_len = config.rollout_length
        slots = _begin_rollout(self, self.task.num_envs)
        frames = self.task.states_all(plan)        # every observation of the planned rollout, one launch
        if slots is not None and self._pixel_rollout.eligible():
            self._pixel_rollout.run(frames, slots)     # three launches per rollout step (csrc/conv_v2.hip)
            self._rollout_step += t_len + 1
            n = self.task.num_envs
            if getattr(config, 'reuse_rollout_activations', True):
                # A2C_agent.py:29-64 backpropagates through the rollout's own forwards; so does this update: the conv layers of its
                # one batched forward take the rollout's stored outputs (same inputs, same parameters) instead of recomputing them
                # -- 31 us of a 315 us agent step at 16 x 5.  (The recomputed forward runs the four-wave kernel shape, the rollout
                # the eight-wave one: the two differ in fp32 summation order, parameters agree to ~1e-6 per update, not bit for bit.)
                acts, body = self._pixel_rollout.bufs, self.network.phi_body
                for conv, key in ((body.conv1, 'y1'), (body.conv2, 'y2'), (body.conv3, 'y3')):
                    y = acts[key][:t_len]
                    conv._y_pre = y.reshape((t_len * n,) + tuple(y.shape[2:]))
                # (and fc4's output: the folded features the rollout's head launches left -- the update's fc4 + head node runs the
                # head launch alone)
                self.network._phi_pre = acts['phi'][:t_len].reshape(t_len * n, 512)
            return self._learn_stacked(frames[:t_len].reshape((t_len * n,) + tuple(frames.shape[2:])),
                                       slots.action[:t_len].reshape(-1), slots.v[:t_len + 1].unsqueeze(-1), plan.reward,
                                       plan.mask, apply=apply)
        states, actions, values = [], [], []
        with torch.no_grad():
            for t in range(t_len):
                state_t = self._dev_state(frames[t])
                prediction = self.network(state_t)
                self._rollout_step += 1
                states.append(state_t)
                actions.append(prediction['action'])
                values.append(prediction['v'])
            values.append(self.network(self._dev_state(frames[t_len]))['v'])
        if slots is not None:
            slots.end()
            if (values[-1].data_ptr() == slots.v[t_len].data_ptr() and states[-1].data_ptr() == frames[t_len - 1].data_ptr()):
                # the forwards wrote into the slots' rows and read the frames in place: the rollout IS these buffers (no stack / cat)
                n = self.task.num_envs
                return self._learn_stacked(frames[:t_len].reshape((t_len * n,) + tuple(frames.shape[2:])),
                                           slots.action[:t_len].reshape(-1), slots.v[:t_len + 1].unsqueeze(-1), plan.reward,
                                           plan.mask, apply=apply)
        return self._learn(states, actions, values, [plan.reward[t] for t in range(t_len)],
                           [plan.mask[t] for t in range(t_len)], apply=apply)

    def _learn_apply(self, out4=None):
        """The part of an update behind the backward pass: gradient exchange (data parallel: the loss is the mean over this
        rank's T x N/G rows; equal shards -> the mean of the G gradients is the gradient of the global mean loss; ONE
        all-reduce), optional hook, clip + optimizer step identically on every rank."""
        self.dp.sum_grads(self._fused.flat.grad, 1.0 / self.dp.world if self.dp.active else 1.0)
        if self.grad_hook is not None:
            self.grad_hook(self._fused.flat.grad)
        self._fused.step(self.config.gradient_clip)
        return out4

    def _learn(self, states, actions, values, rewards, masks, apply=True):
        """A2C_agent.py:43-64 on a finished rollout.  The reference keeps the autograd graph of every rollout forward and
        backpropagates through all T of them; the parameters do not change inside a rollout, so ONE forward over the T x N
        stored observations with the stored actions gives the same log-probabilities, entropies and values and ONE
        backward the same gradient (up to fp32 summation order) -- T-fold fewer backward launches and no gradient
        accumulation passes (rocprofv3, profiles/r02z9_*: 490 launches per A2C step, 86 of them `grad += ...`)."""
        # (stored actions: [N] per step for Categorical policies, [N, action_dim] for Gaussian ones -- a2c_continuous,
        # examples.py:384-404 -- so the environment axis is concatenated and the action axis kept)
        return self._learn_stacked(torch.cat(states, dim=0), torch.cat(actions, dim=0), torch.stack(values).contiguous(),
                                   torch.stack(rewards).contiguous(), torch.stack(masks).contiguous(), apply=apply)

    def _learn_stacked(self, states, actions, value, reward, mask, apply=True):
        """_learn on a rollout that is already laid out: states [T * N, ...], actions [T * N(, A)], value [T + 1, N, 1],
        reward / mask [T, N, 1]."""
        config = self.config
        adv, ret = ops.gae(reward, mask, value, config.discount, config.gae_tau, config.use_gae)
        prediction = self.network(states, actions)
        out4, (g_lp, g_ent, g_v) = ops.a2c_loss(prediction['log_pi_a'].detach(), prediction['entropy'].detach(),
                                                prediction['v'].detach(), adv.reshape(-1, 1), ret.reshape(-1, 1),
                                                config.entropy_weight, config.value_loss_weight)
        self._fused.zero_grad(direct=not self.dp.active)
        # (one batched forward: every parameter is used once -> the layers write their gradients in place, nets.direct_param_grads;
        # data parallel keeps autograd's accumulation: its post-accumulate hooks start the early gradient exchange)
        from .nets import direct_param_grads
        # (the optimizer step follows with nothing reading the gradient in between -> the conv layers' slab folds ride in its
        # norm launch; config.defer_conv_folds = False restores one fold launch per layer)
        defer = self._fused if (apply and not self.dp.active and self.grad_hook is None
                                and getattr(config, 'defer_conv_folds', True)) else None
        with direct_param_grads(not self.dp.active, defer_folds_to=defer, covers=[self._fused]):
            torch.autograd.backward([prediction['log_pi_a'], prediction['entropy'], prediction['v']], [g_lp, g_ent, g_v])
        return self._learn_apply(out4) if apply else out4

    def step(self):
        if getattr(self.task, 'on_device', False):
            return self._step_device()
        config = self.config
        states = self.states
        seen, actions, values, rewards_l, masks_l = [], [], [], [], []
        slots = _begin_rollout(self, len(states))       # the same single draw per rollout as the device path
        for _ in range(config.rollout_length):
            with torch.no_grad():
                state_t = tensor(config.state_normalizer(states))
                prediction = self.network(state_t)
            self._rollout_step += 1
            next_states, rewards, terminals, info = self.task.step(to_np(prediction['action']))
            self.record_online_return(info)
            rewards = config.reward_normalizer(rewards)
            seen.append(state_t)
            actions.append(prediction['action'])
            values.append(prediction['v'])
            rewards_l.append(tensor(rewards).unsqueeze(-1))
            masks_l.append(tensor(1 - terminals).unsqueeze(-1))
            states = next_states
            self.total_steps += self.dp.global_workers
        self.states = states
        with torch.no_grad():
            values.append(self.network(config.state_normalizer(states))['v'])
        if slots is not None:
            slots.end()
        self.last_loss = self._learn(seen, actions, values, rewards_l, masks_l)


class NStepDQNAgent(BaseAgent):
    """NStepDQN_agent.py:12-67: on-policy n-step Q-learning; returns via the scan kernel (use_gae off)."""

    def __init__(self, config):
        BaseAgent.__init__(self, config)
        self.config = config
        self.task = config.task_fn()
        self.network = config.network_fn()
        self.target_network = config.network_fn()
        self.optimizer = config.optimizer_fn(self.network.parameters())
        self._fused = FusedOptimizer.adopt(self.optimizer)
        self._target_flat = FlatParams(list(self.target_network.parameters()),
                                       koc=nature_conv_weights(list(self.target_network.parameters())))
        ops.copy_f32(self._target_flat.flat, self._fused.flat.flat)
        self.total_steps = 0
        self.states = self.task.reset()

    def step(self):
        config = self.config
        storage = Storage(config.rollout_length)
        states = self.states
        for _ in range(config.rollout_length):
            q = self.network(config.state_normalizer(states))['q']
            epsilon = config.random_action_prob(config.num_workers)
            actions = epsilon_greedy(epsilon, to_np(q))
            next_states, rewards, terminals, info = self.task.step(actions)
            self.record_online_return(info)
            rewards = config.reward_normalizer(rewards)
            storage.feed({'q': q, 'action': tensor(actions).unsqueeze(-1).long(),
                          'reward': tensor(rewards).unsqueeze(-1), 'mask': tensor(1 - terminals).unsqueeze(-1)})
            states = next_states
            self.total_steps += config.num_workers
            if self.total_steps // config.num_workers % config.target_network_update_freq == 0:
                ops.copy_f32(self._target_flat.flat, self._fused.flat.flat)
        self.states = states
        storage.placeholder()
        with torch.no_grad():
            ret = self.target_network(config.state_normalizer(states))['q']
            ret = torch.max(ret, dim=1, keepdim=True)[0]
        t_len = config.rollout_length
        reward = torch.stack(storage.reward[:t_len]).contiguous()
        mask = torch.stack(storage.mask[:t_len]).contiguous()
        zeros = torch.zeros((t_len + 1,) + tuple(ret.shape), device=ret.device)
        zeros[t_len] = ret
        _, rets = ops.gae(reward, mask, zeros, config.discount, 1.0, False)
        for i in range(t_len):
            storage.ret[i] = rets[i]
        entries = storage.extract(['q', 'action', 'ret'])
        q_a = entries.q.gather(1, entries.action)
        n = q_a.numel()
        grad = (q_a.detach() - entries.ret) / n  # d/dq of 0.5 * mean((q - ret)^2)
        self._fused.zero_grad()
        q_a.backward(grad)
        self._fused.step(config.gradient_clip)


class PPOAgent(BaseAgent):
    """PPO_agent.py:12-100."""

    def __init__(self, config):
        BaseAgent.__init__(self, config)
        self.config = config
        from .dist import DataParallel
        self.dp = DataParallel(config)
        self.task = config.task_fn()
        self.network = config.network_fn()
        if config.shared_repr:
            self.opt = config.optimizer_fn(self.network.parameters())
            self._fused = FusedOptimizer.adopt(self.opt)
        else:
            self.actor_opt = config.actor_opt_fn(self.network.actor_params)
            self.critic_opt = config.critic_opt_fn(self.network.critic_params)
            actor_ids = {id(p) for p in self.network.actor_params}
            if any(id(p) in actor_ids for p in self.network.critic_params):
                # (the reference cannot run this configuration either: PPO_agent.py:89-96 calls policy_loss.backward() and then
                # value_loss.backward() on ONE forward -- with a parameterised phi_body the first call frees the shared part of the
                # graph and steps its weights in place, and the second raises inside autograd; tests/test_nets_host_logic.py)
                raise NotImplementedError("PPO with a parameterised phi_body shared by separate actor / critic optimisers: the "
                                          "reference's own update (PPO_agent.py:89-96) raises inside autograd for it; use "
                                          "shared_repr=True (one optimiser, examples.py:525-550)")
            self._fused_actor = FusedOptimizer.adopt(self.actor_opt)
            self._fused_critic = FusedOptimizer.adopt(self.critic_opt)
        _dp_sync_start(self.dp, self.network, *([self._fused] if config.shared_repr else [self._fused_actor, self._fused_critic]))
        if config.shared_repr:
            _dp_plan_exchange(self.dp, self.network, self._fused)
        self.total_steps = 0
        from .device_env import DeviceAtariVec, DeviceContinuousVec
        from .ppo_mlp import MlpPPO
        self.grad_hook = None
        self._mlp = MlpPPO(self)        # csrc/ppo_mlp.hip: two small tanh MLPs + two Adam optimisers as persistent kernels
        if DeviceAtariVec.eligible(self.task, config):      # synthetic Atari emulators: the environments live on the device
            self.task = DeviceAtariVec(self.task)
            self.states = None
        else:
            self.states = self.task.reset()
            self.states = config.state_normalizer(self.states)
            if DeviceContinuousVec.eligible(self.task, config) and self._mlp.usable():
                # synthetic continuous-control environments (BASELINE configs[2]): observations, counters and the observation
                # statistics move to the device; rollouts are one launch each (_step_device_mlp)
                self.task = DeviceContinuousVec(self.task, self.states, config.state_normalizer)
                self.states = None
        self._dev_graph = _OnPolicyGraph(self, optimizer_inside=False)
        self._dev_state = _device_state_fn(self)
        if config.shared_repr:
            self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda step: 1 - step / config.max_steps)
        self._graphed = _GraphedPPO(self)
        self._rollout_graph = dict(calls=0, graph=None, failed=False, k=0)
        self._rollout_step = 0
        _install_sampler(self)
        self._pixel_rollout = _PixelRollout(self)
        self.network.fuse_fc4_head = bool(getattr(config, 'fuse_fc4_head', True))
        self.network.rollout_fc4_slices = bool(getattr(config, 'rollout_fc4_slices', True))
        # action noise of the device rollout: the rank-invariant stream when one is configured, else a seed of its own
        self._noise_seed = self.dp.noise_seed

    def close(self):
        close_obj(self.task)
        self.dp.close()

    def save(self, filename):
        """The reference's two files (BaseAgent.py:24-27) plus `<filename>.sampler`: seed and position of the counter-hash action
        noise of the device rollout, without which a restored agent would replay the noise of its first rollouts."""
        BaseAgent.save(self, filename)
        if self.dp.is_main:
            with open(filename + '.sampler', 'wb') as f:
                pickle.dump(self.dp.sampler_state(), f)

    def load(self, filename):
        BaseAgent.load(self, filename)
        if os.path.isfile(filename + '.sampler'):      # (checkpoints of the reference itself have no such file)
            with open(filename + '.sampler', 'rb') as f:
                self.dp.load_sampler_state(pickle.load(f), Config.DEVICE)
            self._noise_seed = self.dp.noise_seed

    def _step_device(self):
        """step() over device-resident environments (see A2CAgent._step_device): the rollout -- T x [observations, image
        normalisation, no-grad forward, action sample], bootstrap forward, GAE scan, advantage normalisation -- is one
        captured graph whose outputs are the rollout entries; the optimisation phase is unchanged."""
        config = self.config
        plan = self.task.plan(config.rollout_length, config.reward_normalizer)
        for t in range(config.rollout_length):
            self.record_online_return(plan.infos[t])
            self.total_steps += self.dp.global_workers
        if self.dp.active:      # PPO_agent.py:66 over the GLOBAL rollout: three all-reduced scalars, outside the captured region
            from .dist import global_advantage_normalize_

            def normalise(entries):
                global_advantage_normalize_(entries.advantage)
                return entries
            entries = self._dev_graph.run(plan, lambda pl: self._rollout_compute(pl, normalise=False), tail=normalise)
        else:
            entries = self._dev_graph.run(plan, self._rollout_compute)
        if config.shared_repr:
            self.lr_scheduler.step(self.total_steps)
        self.optimize(entries)

    def _rollout_compute(self, plan, normalise=True):
        config = self.config
        t_len = config.rollout_length
        slots = _begin_rollout(self, self.task.num_envs)
        frames = self.task.states_all(plan)        # every observation of the planned rollout, one launch
        storage = Storage(t_len)
        fused = slots is not None and self._pixel_rollout.eligible()
        if fused:
            self._pixel_rollout.run(frames, slots)     # three launches per rollout step (csrc/conv_v2.hip)
            self._rollout_step += t_len + 1
        else:
            with torch.no_grad():
                for t in range(t_len):
                    state_t = self._dev_state(frames[t])
                    prediction = self.network(state_t)
                    self._rollout_step += 1
                    storage.feed(prediction)
                    storage.feed({'reward': plan.reward[t], 'mask': plan.mask[t], 'state': state_t})
                prediction = self.network(self._dev_state(frames[t_len]))
                self._rollout_step += 1
            if slots is not None:
                slots.end()
        if fused or (slots is not None and prediction['v'].data_ptr() == slots.v[t_len].data_ptr()
                     and state_t.data_ptr() == frames[t_len - 1].data_ptr()):
            # the forwards wrote into the slots' rows and read the frames in place: the rollout IS these buffers (Storage's
            # per-key torch.cat launches and the reward / mask / value stacks of the scan disappear)
            from collections import namedtuple
            rows = t_len * self.task.num_envs
            adv, ret = ops.gae(plan.reward, plan.mask, slots.v[:t_len + 1].unsqueeze(-1), config.discount, config.gae_tau,
                               config.use_gae)
            entry_cls = namedtuple('Entry', ['state', 'action', 'log_pi_a', 'ret', 'advantage'])
            entries = entry_cls(frames[:t_len].reshape((rows,) + tuple(frames.shape[2:])), slots.action[:t_len].reshape(rows),
                                slots.log_pi_a[:t_len].reshape(rows, 1), ret.reshape(rows, 1), adv.reshape(rows, 1))
        else:
            storage.feed(prediction)
            storage.placeholder()
            _rollout_scan(storage, config, prediction['v'])
            entries = storage.extract(['state', 'action', 'log_pi_a', 'ret', 'advantage'])
            entry_cls = entries.__class__
            entries = entry_cls(*[x.detach() for x in entries])
        if not normalise:
            return entries
        if self.dp.active:
            from .dist import global_advantage_normalize_
            global_advantage_normalize_(entries.advantage)
        else:
            ops.adv_normalize_(entries.advantage)
        return entries

    def _step_device_mlp(self):
        """step() over device_env.DeviceContinuousVec: PPO_agent.py:32-49 is ONE launch (dra_ppo_mlp_rollout: per step the
        normalised observation, both forwards, action = mean + softplus(std) * noise, log-probability, environment step,
        running observation statistics), then the return scan, the advantage normalisation and optimize()."""
        import ctypes
        from collections import namedtuple
        from ._lib import lib, stream_ptr
        from .ppo_mlp import RolloutIO
        config, task, dp = self.config, self.task, self.dp
        t_len, n = int(config.rollout_length), task.num_envs
        b = task.buffers(t_len)
        events = task.shadow(t_len)
        if dp.step_dev is None:
            dp.step_dev = torch.zeros(1, dtype=torch.int64, device=Config.DEVICE)
        cfg, actor, critic = self._mlp.structs()
        self._mlp.sync_counts()
        io = RolloutIO()
        io.env_state, io.env_counter, io.env_seed = task.env_state.data_ptr(), task.env_counter.data_ptr(), task.env_seed.data_ptr()
        io.rms, io.cur_state, io.sampler_step = task.rms.data_ptr(), task.cur_state.data_ptr(), dp.step_dev.data_ptr()
        io.out_state, io.out_action, io.out_log_pi_a = b['state'].data_ptr(), b['action'].data_ptr(), b['log_pi_a'].data_ptr()
        io.out_v, io.out_reward, io.out_mask = b['v'].data_ptr(), b['reward'].data_ptr(), b['mask'].data_ptr()
        io.env0, io.n_global, io.noise_seed, io.horizon = dp.lo, dp.global_workers, self._noise_seed, task.horizon
        io.reward_coef, io.rms_epsilon, io.rms_clip = float(config.reward_normalizer.coef), task.rms_epsilon, task.rms_clip
        io.rms_update = 1 if (task.rms_kind == 'meanstd' and not config.state_normalizer.read_only) else 0
        io.t_len, io.n_env = t_len, n
        lib.dra_ppo_mlp_rollout(ctypes.byref(cfg), ctypes.byref(actor), ctypes.byref(critic), ctypes.byref(io), stream_ptr())
        self._rollout_step += t_len + 1
        for t, i, ret in events:        # BaseAgent.record_online_return at the step the episode ended
            at = self.total_steps + t * dp.global_workers + i
            self.logger.add_scalar('episodic_return_train', ret, at)
            self.logger.info('steps %d, episodic_return_train %s' % (at, ret))
        self.total_steps += t_len * dp.global_workers
        adv, ret = ops.gae(b['reward'], b['mask'], b['v'], config.discount, config.gae_tau, config.use_gae)
        entry_cls = namedtuple('Entry', ['state', 'action', 'log_pi_a', 'ret', 'advantage'])
        rows = t_len * n
        entries = entry_cls(b['state'].view(rows, -1), b['action'].view(rows, -1), b['log_pi_a'].view(rows, 1),
                            ret.view(rows, 1), adv.view(rows, 1))
        ops.adv_normalize_(entries.advantage)
        self.optimize(entries)

    def step(self):
        if getattr(self.task, 'on_device', False):
            from .device_env import DeviceContinuousVec
            if isinstance(self.task, DeviceContinuousVec):
                return self._step_device_mlp()
            return self._step_device()
        config = self.config
        storage = Storage(config.rollout_length)
        states = self.states
        slots = _begin_rollout(self, len(states))       # the same single draw per rollout as the device path
        for _ in range(config.rollout_length):
            prediction, state_t = self._act(states)
            self._rollout_step += 1
            next_states, rewards, terminals, info = self.task.step(to_np(prediction['action']))
            self.record_online_return(info)
            rewards = config.reward_normalizer(rewards)
            next_states = config.state_normalizer(next_states)
            storage.feed(prediction)
            storage.feed({'reward': tensor(rewards).unsqueeze(-1), 'mask': tensor(1 - terminals).unsqueeze(-1),
                          'state': state_t})
            states = next_states
            self.total_steps += self.dp.global_workers
        self.states = states
        prediction, _ = self._act(states)
        self._rollout_step += 1
        if slots is not None:
            slots.end()
        storage.feed(prediction)
        storage.placeholder()
        _rollout_scan(storage, config, prediction['v'])

        entries = storage.extract(['state', 'action', 'log_pi_a', 'ret', 'advantage'])
        entry_cls = entries.__class__
        entries = entry_cls(*[x.detach() for x in entries])
        if self.dp.active:      # PPO_agent.py:66 over the GLOBAL rollout: three scalars all-reduced
            from .dist import global_advantage_normalize_
            global_advantage_normalize_(entries.advantage)
        else:
            ops.adv_normalize_(entries.advantage)  # PPO_agent.py:66 in place

        if config.shared_repr:
            self.lr_scheduler.step(self.total_steps)
        self.optimize(entries)

    def _act(self, states):
        """The rollout's no-grad forward (PPO_agent.py:35-36, incl. the action sample) -> (prediction dict, the
        f32 device copy of `states` that PPO_agent.py:41 stores).  After two eager calls it is replayed from a
        captured hipGraph over a static input buffer: same kernels, same arguments, and torch's graph-safe Philox
        bookkeeping continues the generator's sequence exactly as the eager sampling kernels would."""
        g = self._rollout_graph
        cfg = self.config
        usable = (g is not None and not g['failed'] and getattr(cfg, 'graph_rollout', getattr(cfg, 'graph_update', True))
                  and Config.DEVICE.type == 'cuda' and not isinstance(states, torch.Tensor))
        if usable:
            x = np.ascontiguousarray(np.asarray(states, dtype=np.float32))    # what tensor() uploads (torch_utils.py:23)
            g['calls'] += 1
            if g['calls'] > 2:
                if g['graph'] is None or g['in'].shape != x.shape:
                    validate = torch.distributions.Distribution._validate_args
                    try:
                        g['in'] = torch.zeros(x.shape, dtype=torch.float32, device=Config.DEVICE)
                        g['stage'] = [torch.empty(x.shape, dtype=torch.float32).pin_memory() for _ in range(4)]
                        g['events'] = [None] * 4
                        torch.distributions.Distribution.set_default_validate_args(False)
                        graph = torch.cuda.CUDAGraph()
                        # (a categorical net reads its uniforms from the rollout's one draw: the captured forward reads a static
                        # row that every replay fills from that draw first)
                        slots = getattr(self.network, 'rollout_slots', None)
                        g['u'] = None
                        if slots is not None and getattr(self.network, 'sampler', None) is None:
                            g['u'] = torch.zeros(x.shape[0], dtype=torch.float32, device=Config.DEVICE)
                            slots.pin(g['u'])
                        torch.cuda.synchronize()
                        with _capture(graph):
                            with torch.no_grad():
                                g['out'] = self.network(g['in'])
                        g['graph'] = graph
                    except Exception as e:
                        _capture_failed(cfg, "the rollout forward", e)
                        g['failed'] = True
                    finally:
                        torch.distributions.Distribution.set_default_validate_args(validate)
                        if getattr(self.network, 'rollout_slots', None) is not None:
                            self.network.rollout_slots.pin(None)
                if not g['failed']:
                    k = g['k']
                    g['k'] = (k + 1) % 4
                    if g['events'][k] is not None:
                        g['events'][k].synchronize()
                    g['stage'][k].numpy()[...] = x
                    g['in'].copy_(g['stage'][k], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    g['events'][k] = ev
                    if g['u'] is not None:
                        u = self.network.rollout_slots.next_uniform()
                        if u is not None and u.numel() == g['u'].numel():
                            g['u'].copy_(u)
                        else:
                            g['u'].uniform_()
                    g['graph'].replay()
                    return {key: v.clone() for key, v in g['out'].items()}, g['in'].clone()
        with torch.no_grad():
            prediction = self.network(states)
        return prediction, tensor(states)

    def _minibatch(self, entry, prepared=False, weight=1.0):
        """One minibatch update (PPO_agent.py:77-99) on already-gathered rows.  `prepared`: the optimisers are in
        graph mode and the caller has advanced their step scalars (FusedOptimizer.prepare_step).  `weight` (data
        parallel): rows here / rows of the GLOBAL minibatch -- the kernels' means are over the local rows, so
        weight * local gradient summed over ranks is the gradient of the global mean."""
        config = self.config
        dp = self.dp
        n_local = entry.state.size(0)
        if n_local > 0:
            prediction = self.network(entry.state, entry.action)
            out3, (g_lp, g_ent, g_v) = ops.ppo_loss(
                prediction['log_pi_a'].detach(), prediction['entropy'].detach(), prediction['v'].detach(),
                entry.log_pi_a, entry.advantage, entry.ret, config.ppo_ratio_clip, config.entropy_weight)
        else:       # this rank holds no row of the global minibatch: it still takes part in the exchange
            prediction, out3, g_lp, g_ent, g_v = None, torch.zeros(3, device=Config.DEVICE), None, None, None
        if config.shared_repr:
            self._fused.zero_grad(direct=not dp.active and prediction is not None)
            dp.set_weight(weight)       # (the fc4 segment of the exchange goes out from inside the backward pass: dist.py)
            if prediction is not None:
                from .nets import direct_param_grads
                defer = self._fused if (not dp.active and self.grad_hook is None
                                        and getattr(config, 'defer_conv_folds', True)) else None
                with direct_param_grads(not dp.active, defer_folds_to=defer, covers=[self._fused]):     # (one forward per minibatch: see A2CAgent._learn)
                    torch.autograd.backward([prediction['log_pi_a'], prediction['entropy'], prediction['v']],
                                            [g_lp, g_ent, g_v])
            dp.sum_grads(self._fused.flat.grad, weight)
            if self.grad_hook is not None:
                self.grad_hook(self._fused.flat.grad)
            self._fused.step(config.gradient_clip)
        else:
            approx_kl = out3[2].item()
            if dp.active:   # the KL gate (PPO_agent.py:88) must fall the same way on every rank: global mean KL
                approx_kl = dp.sum_scalars([approx_kl * weight])[0]
            if approx_kl <= 1.5 * config.target_kl:
                self._fused_actor.zero_grad()
                if prediction is not None:
                    torch.autograd.backward([prediction['log_pi_a'], prediction['entropy']], [g_lp, g_ent], retain_graph=True)
                dp.sum_grads(self._fused_actor.flat.grad, weight)
                self._fused_actor.step(None)
            self._fused_critic.zero_grad()
            if prediction is not None:
                prediction['v'].backward(g_v)
            dp.sum_grads(self._fused_critic.flat.grad, weight)
            self._fused_critic.step(None)
        return out3

    def optimize(self, entries):
        """PPO_agent.py:71-99: epochs of shuffled minibatches over the (detached) rollout entries."""
        config = self.config
        dp = self.dp
        if self._mlp.usable() and self._mlp.optimize(entries):      # all epochs x minibatches in one launch (csrc/ppo_mlp.hip)
            return
        self._mlp.sync_counts()
        if dp.invariant_sampling:        # G ranks, or one process asked to behave exactly like G ranks would
            return self._optimize_data_parallel(entries)
        if self._graphed.usable() and self._graphed.optimize(entries):
            return
        entry_cls = entries.__class__
        for _ in range(config.optimization_epochs):
            sampler = random_sample(np.arange(entries.state.size(0)), config.mini_batch_size)
            for batch_indices in sampler:
                batch_indices = tensor(batch_indices).long()
                entry = entry_cls(*[x[batch_indices] for x in entries])
                out3 = self._minibatch(entry)
        self.last_loss = out3

    def _optimize_data_parallel(self, entries):
        """The same epochs x minibatches over the GLOBAL rollout (T x N rows, time-major like Storage.extract): every rank
        draws the same permutation (dist.DataParallel.permutation), forms the same global minibatches as a single
        process would, and contributes the rows that belong to its environments [lo, hi)."""
        config, dp = self.config, self.dp
        entry_cls = entries.__class__
        n_glob, n_loc = dp.global_workers, dp.hi - dp.lo
        t_len = entries.state.size(0) // n_loc
        total = t_len * n_glob
        mb = config.mini_batch_size
        for _ in range(config.optimization_epochs):
            perm = np.asarray(dp.permutation(total))
            full = total // mb * mb
            batches = list(perm[:full].reshape(-1, mb)) + ([perm[full:]] if total % mb else [])
            for g_idx in batches:
                t, n = g_idx // n_glob, g_idx % n_glob
                mine = (n >= dp.lo) & (n < dp.hi)
                local = t[mine] * n_loc + (n[mine] - dp.lo)
                rows = tensor(local).long()
                entry = entry_cls(*[x[rows] for x in entries])
                out3 = self._minibatch(entry, weight=float(len(local)) / float(len(g_idx)))
        self.last_loss = out3


class _GraphedPPO:
    """PPO's optimisation phase (PPO_agent.py:71-99: epochs x minibatches, each a forward, the clip loss, a
    backward and one or two Adam steps -- 320 tiny updates per rollout at ppo_continuous's sizes) with every
    full-size minibatch replayed from captured hipGraphs: the rollout's rows live in static buffers, a static
    index tensor selects the minibatch inside the graph, Adam's step-dependent scalars (and a scheduled lr) reach
    the kernels through device memory (FusedOptimizer.prepare_step).  shared_repr: one graph (forward, loss,
    backward, clip, step).  Separate actor / critic optimisers: the KL gate of PPO_agent.py:88-93 needs the
    forward's result on the host first, so a forward+loss graph runs, the host decides, and one of two update
    graphs (actor+critic, critic only) -- each recomputing the same forward -- runs.  Same kernels and arguments
    as the eager path: bit-identical parameters.  First rollout and odd-sized remainder minibatches run eagerly."""
    WARMUP = 1

    def __init__(self, agent):
        self.agent = agent
        self.rollouts = 0
        self.static = None
        self.idx = None
        self.graphs = None
        self.out3 = None
        self.failed = False
        self._up = None

    def usable(self):
        a = self.agent
        cfg = a.config
        if self.failed or getattr(cfg, 'graph_update', True) is False or a.grad_hook is not None or a.dp.active:
            return False
        if Config.DEVICE.type != 'cuda':
            return False
        opts = [a._fused] if cfg.shared_repr else [a._fused_actor, a._fused_critic]
        return all(o.kind == 'adam' for o in opts)      # scheduled / stepped scalars travel through device memory

    def _capture(self, entry_cls):
        a = self.agent
        cfg = a.config
        rows = lambda: entry_cls(*ops.gather_rows(list(self.static), self.idx))     # all five fields, one launch
        opts = [a._fused] if cfg.shared_repr else [a._fused_actor, a._fused_critic]
        for o in opts:
            o.enable_graph_mode()
        validate = torch.distributions.Distribution._validate_args
        torch.distributions.Distribution.set_default_validate_args(False)   # argument checks synchronise with the host
        try:
            torch.cuda.synchronize()
            if cfg.shared_repr:
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    self.out3 = a._minibatch(rows(), prepared=True)
                self.graphs = dict(all=g)
            else:
                net = a.network
                g_fwd, g_both, g_critic = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()

                def forward_loss():
                    e = rows()
                    p = net(e.state, e.action)
                    out3, grads = ops.ppo_loss(p['log_pi_a'].detach(), p['entropy'].detach(), p['v'].detach(), e.log_pi_a,
                                               e.advantage, e.ret, cfg.ppo_ratio_clip, cfg.entropy_weight)
                    return p, out3, grads

                with _capture(g_fwd):
                    with torch.no_grad():
                        _, self.out3, _ = forward_loss()
                with _capture(g_both):
                    p, _, (g_lp, g_ent, g_v) = forward_loss()
                    a._fused_actor.zero_grad()
                    torch.autograd.backward([p['log_pi_a'], p['entropy']], [g_lp, g_ent])
                    a._fused_actor.step(None)
                    a._fused_critic.zero_grad()
                    p['v'].backward(g_v)
                    a._fused_critic.step(None)
                with _capture(g_critic):
                    p, _, (g_lp, g_ent, g_v) = forward_loss()
                    a._fused_critic.zero_grad()
                    p['v'].backward(g_v)
                    a._fused_critic.step(None)
                self.graphs = dict(fwd=g_fwd, both=g_both, critic=g_critic)
        finally:
            torch.distributions.Distribution.set_default_validate_args(validate)

    def optimize(self, entries):
        a = self.agent
        cfg = a.config
        self.rollouts += 1
        if self.rollouts <= self.WARMUP:
            return False
        entry_cls = entries.__class__
        mb = cfg.mini_batch_size
        n = entries.state.size(0)
        try:
            if self.static is None or any(s.shape != x.shape for s, x in zip(self.static, entries)):
                self.static = entry_cls(*[x.clone() for x in entries])
                self.idx = torch.zeros(mb, dtype=torch.int64, device=entries.state.device)
                self.graphs = None
                from .replay import _PinnedUploader
                self._up = _PinnedUploader(torch.int64, mb, entries.state.device)
            else:
                for st, x in zip(self.static, entries):
                    st.copy_(x)
            if self.graphs is None:
                self._capture(entry_cls)
        except Exception as e:
            _capture_failed(cfg, "the PPO minibatch update", e)
            self.failed = True
            for o in ([a._fused] if cfg.shared_repr else [a._fused_actor, a._fused_critic]):
                o.graph_mode = False
            return False
        if cfg.shared_repr and n % mb == 0 and getattr(cfg, 'graph_ppo_block', True):
            return self._optimize_block(entry_cls, n, mb)
        opts = [a._fused] if cfg.shared_repr else None
        for _ in range(cfg.optimization_epochs):
            for batch_indices in random_sample(np.arange(n), mb):
                if len(batch_indices) != mb:          # remainder minibatch: eager, optimisers stay in graph mode
                    bi = tensor(batch_indices).long()
                    if cfg.shared_repr:
                        a._fused.prepare_step()
                        out3 = a._minibatch(entry_cls(*[x[bi] for x in self.static]), prepared=True)
                    else:
                        out3 = self._eager_split(entry_cls(*[x[bi] for x in self.static]))
                    continue
                self._up.upload_into(self.idx, np.asarray(batch_indices, dtype=np.int64))
                if cfg.shared_repr:
                    a._fused.prepare_step()
                    self.graphs['all'].replay()
                else:
                    self.graphs['fwd'].replay()
                    if self.out3[2].item() <= 1.5 * cfg.target_kl:
                        a._fused_actor.prepare_step()
                        a._fused_critic.prepare_step()
                        self.graphs['both'].replay()
                    else:
                        a._fused_critic.prepare_step()
                        self.graphs['critic'].replay()
                out3 = self.out3
        a.last_loss = out3
        return True

    def _optimize_block(self, entry_cls, n, mb):
        """shared_repr with full minibatches only (ppo_pixel: 4 epochs x 4 minibatches of 256): the WHOLE optimisation phase is one
        captured graph -- minibatch j gathers its rows by row j of one [epochs x n / mb, mb] index block and reads its Adam scalars
        from row j of one [., 2] block, both uploaded once per rollout -- instead of a graph launch and two uploads per minibatch
        (32 copies of ~4 us in the stream + 15 launch gaps per agent step: profiles/r05t_kernel_stats_ppo_pixel_8.txt).  The
        permutations are drawn exactly as the per-minibatch loop draws them; same kernels, same arguments: same parameters."""
        a = self.agent
        cfg = a.config
        k = cfg.optimization_epochs * (n // mb)
        opt = a._fused
        try:
            if getattr(self, 'block', None) is None or self.block['k'] != k or self.block['static'] is not self.static:
                from .replay import _PinnedUploader
                idx_all = torch.zeros((k, mb), dtype=torch.int64, device=self.static.state.device)
                block = opt.enable_block_mode(k)
                validate = torch.distributions.Distribution._validate_args
                torch.distributions.Distribution.set_default_validate_args(False)
                keep = opt._hyper_dev
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with _capture(g):
                        for j in range(k):
                            opt._hyper_dev = block[j]
                            out3 = a._minibatch(entry_cls(*ops.gather_rows(list(self.static), idx_all[j])), prepared=True)
                finally:
                    opt._hyper_dev = keep
                    torch.distributions.Distribution.set_default_validate_args(validate)
                self.block = dict(k=k, static=self.static, idx=idx_all, graph=g, out3=out3,
                                  up=_PinnedUploader(torch.int64, k * mb, idx_all.device))
        except Exception as e:
            _capture_failed(cfg, "the PPO optimisation phase", e)
            self.failed = True
            opt.graph_mode = False
            return False
        b = self.block
        batches = []
        for _ in range(cfg.optimization_epochs):
            batches += [np.asarray(bi, dtype=np.int64) for bi in random_sample(np.arange(n), mb)]
        b['up'].upload_into(b['idx'].view(-1), np.concatenate(batches))
        opt.prepare_steps(k)
        b['graph'].replay()
        a.last_loss = b['out3']
        return True

    def _eager_split(self, entry):
        """remainder minibatch with separate optimisers in graph mode: same arithmetic as PPOAgent._minibatch."""
        a = self.agent
        cfg = a.config
        p = a.network(entry.state, entry.action)
        out3, (g_lp, g_ent, g_v) = ops.ppo_loss(p['log_pi_a'].detach(), p['entropy'].detach(), p['v'].detach(), entry.log_pi_a,
                                                entry.advantage, entry.ret, cfg.ppo_ratio_clip, cfg.entropy_weight)
        if out3[2].item() <= 1.5 * cfg.target_kl:
            a._fused_actor.prepare_step()
            a._fused_actor.zero_grad()
            torch.autograd.backward([p['log_pi_a'], p['entropy']], [g_lp, g_ent])
            a._fused_actor.step(None)
        a._fused_critic.prepare_step()
        a._fused_critic.zero_grad()
        p['v'].backward(g_v)
        a._fused_critic.step(None)
        return out3


# ==================================================================================================== replay-based actor-critic
def _f32(x):
    """A sampled replay field as a float32 device tensor (the reference's tensor() narrows its f64 numpy batches to
    f32 BEFORE any arithmetic, torch_utils.py:20-25; the HBM ring hands back the stored dtype)."""
    return x.float() if isinstance(x, torch.Tensor) else tensor(x)


class _DeterministicPolicyAgent(BaseAgent):
    """What DDPG_agent.py:13-100 and TD3_agent.py:13-108 share: one environment, a deterministic policy perturbed by a
    random process (uniform actions during warm-up), every transition fed to the uniform replay -- here the HBM ring:
    17-float states and 6-float actions are just small "frames" -- one sampled minibatch per step once warm, and a
    polyak-averaged target network.  Sub-classes provide `warm()` and `learn(batch)`."""

    def __init__(self, config):
        BaseAgent.__init__(self, config)
        self.config = config
        self.task = config.task_fn()
        self.network = config.network_fn()
        self.target_network = config.network_fn()
        self.target_network.load_state_dict(self.network.state_dict())
        self.replay = config.replay_fn()
        self.random_process = config.random_process_fn()
        self.total_steps = 0
        self.state = None

    def close(self):
        close_obj(self.replay)
        close_obj(self.task)

    def _flat_pair(self, target, src):
        """Both networks' parameters re-homed (once) into one flat fp32 buffer each, same tensor order and offsets; the
        modules and their torch optimizers keep seeing the same Parameter objects (views into the buffers)."""
        pair = getattr(self, '_soft_flat', None)
        if pair is None or pair[2] is not target or pair[3] is not src:
            tf, sf = FlatParams(list(target.parameters())), FlatParams(list(src.parameters()))
            if tf.offsets != sf.offsets or tf.numel != sf.numel:
                raise DraError("soft_update: target and source networks differ in parameter layout")
            for p in list(target.parameters()) + list(src.parameters()):
                p.grad = None        # FlatParams' gradient views are not used here (torch optimizers own the gradients)
            pair = self._soft_flat = (tf, sf, target, src)
        return pair[0].flat, pair[1].flat

    def soft_update(self, target, src):
        """target <- target * (1 - mix) + src * mix over every parameter (DDPG_agent.py:26-30): ONE launch of
        `dra_soft_update` over the two networks' flat parameter buffers."""
        t_flat, s_flat = self._flat_pair(target, src)
        ops.soft_update(t_flat, s_flat, self.config.target_network_mix)

    def eval_step(self, state):
        norm = self.config.state_normalizer
        norm.set_read_only()
        with torch.no_grad():
            action = self.network(norm(state))
        norm.unset_read_only()
        return to_np(action)

    def _behaviour_action(self):
        space = self.task.action_space
        if self.total_steps < self.config.warm_up:
            action = [space.sample()]
        else:
            with torch.no_grad():
                action = to_np(self.network(self.state))
            action = action + self.random_process.sample()
        return np.clip(action, space.low, space.high)

    def step(self):
        config = self.config
        if self.state is None:
            self.random_process.reset_states()
            self.state = config.state_normalizer(self.task.reset())
        action = self._behaviour_action()
        next_state, reward, done, info = self.task.step(action)
        next_state = config.state_normalizer(next_state)
        self.record_online_return(info)
        reward = config.reward_normalizer(reward)
        self.replay.feed(dict(state=self.state, action=action, reward=reward, next_state=next_state,
                              mask=1 - np.asarray(done, dtype=np.int32)))
        if done[0]:
            self.random_process.reset_states()
        self.state = next_state
        self.total_steps += 1
        if self.warm():
            tr = self.replay.sample()
            self.learn(_f32(tr.state), _f32(tr.action), _f32(tr.reward).unsqueeze(-1), _f32(tr.next_state),
                       _f32(tr.mask).unsqueeze(-1))


class DDPGAgent(_DeterministicPolicyAgent):
    """DDPG_agent.py:13-100."""

    def warm(self):
        return self.replay.size() >= self.config.warm_up

    def learn(self, states, actions, rewards, next_states, mask):
        net, tgt, gamma = self.network, self.target_network, self.config.discount
        with torch.no_grad():
            phi_next = tgt.feature(next_states)
            q_next = tgt.critic(phi_next, tgt.actor(phi_next))
            y = (gamma * mask * q_next).add_(rewards)
        critic_loss = (net.critic(net.feature(states), actions) - y).pow(2).mul(0.5).sum(-1).mean()
        net.zero_grad()
        critic_loss.backward()
        net.critic_opt.step()
        phi = net.feature(states)
        policy_loss = -net.critic(phi.detach(), net.actor(phi)).mean()
        net.zero_grad()
        policy_loss.backward()
        net.actor_opt.step()
        self.soft_update(tgt, net)


class TD3Agent(_DeterministicPolicyAgent):
    """TD3_agent.py:13-108: clipped double Q, target-policy smoothing, delayed policy / target updates (on the steps
    where total_steps % td3_delay is non-zero, as written at TD3_agent.py:100)."""

    def warm(self):
        return self.total_steps >= self.config.warm_up

    def learn(self, states, actions, rewards, next_states, mask):
        config, net, tgt = self.config, self.network, self.target_network
        space = self.task.action_space
        with torch.no_grad():
            a_next = tgt(next_states)
            noise = torch.randn_like(a_next).mul(config.td3_noise).clamp(-config.td3_noise_clip, config.td3_noise_clip)
            a_next = (a_next + noise).clamp(float(space.low[0]), float(space.high[0]))
            y = rewards + config.discount * mask * torch.min(*tgt.q(next_states, a_next))
        q_1, q_2 = net.q(states, actions)
        critic_loss = F.mse_loss(q_1, y) + F.mse_loss(q_2, y)
        net.zero_grad()
        critic_loss.backward()
        net.critic_opt.step()
        if self.total_steps % config.td3_delay:
            policy_loss = -net.q(states, net(states))[0].mean()
            net.zero_grad()
            policy_loss.backward()
            net.actor_opt.step()
            self.soft_update(tgt, net)


# ==================================================================================================== option-critic
class OptionCriticAgent(BaseAgent):
    """OptionCritic_agent.py:11-119: n-step option-critic over `num_workers` environments.  Per rollout step: one forward
    (q over options, termination beta, intra-option policies), an epsilon-soft option choice that keeps the previous
    option unless it terminates, an action from the chosen option's policy; per rollout: returns bootstrapped from the
    target network, three losses (option values, intra-option policy, termination) through ONE backward and the fused
    clip + optimizer step.  Draw order of the three Categorical samples per step is the reference's."""

    def __init__(self, config):
        BaseAgent.__init__(self, config)
        self.config = config
        self.task = config.task_fn()
        self.network = config.network_fn()
        self.target_network = config.network_fn()
        self.optimizer = config.optimizer_fn(self.network.parameters())
        self._fused = FusedOptimizer.adopt(self.optimizer)
        self._target_flat = FlatParams(list(self.target_network.parameters()),
                                       koc=nature_conv_weights(list(self.target_network.parameters())))
        self._sync_target()
        self.total_steps = 0
        self.worker_index = range_tensor(config.num_workers)
        self.states = config.state_normalizer(self.task.reset())
        self.is_initial_states = torch.ones(config.num_workers, dtype=torch.bool, device=Config.DEVICE)
        self.prev_options = torch.ones(config.num_workers, dtype=torch.int64, device=Config.DEVICE)

    def _sync_target(self):
        ops.copy_f32(self._target_flat.flat, self._fused.flat.flat)

    def sample_option(self, prediction, epsilon, prev_option, is_initial):
        with torch.no_grad():
            q = prediction['q']
            n_opt = q.size(1)
            pi = torch.full_like(q, epsilon / n_opt)                       # epsilon-soft over options ...
            pi.scatter_(1, q.argmax(dim=-1, keepdim=True), 1 - epsilon + epsilon / n_opt)
            keep = torch.zeros_like(q)
            keep[self.worker_index, prev_option] = 1
            beta = prediction['beta']
            pi_hat = (1 - beta) * keep + beta * pi                         # ... unless the previous option continues
            fresh = torch.distributions.Categorical(probs=pi).sample()
            continued = torch.distributions.Categorical(probs=pi_hat).sample()
            return torch.where(is_initial, fresh, continued)

    def step(self):
        config = self.config
        n, w = config.rollout_length, self.worker_index
        storage = Storage(n, ['beta', 'option', 'beta_advantage', 'prev_option', 'init_state', 'eps'])
        for _ in range(n):
            prediction = self.network(self.states)
            epsilon = config.random_option_prob(config.num_workers)
            options = self.sample_option(prediction, epsilon, self.prev_options, self.is_initial_states)
            prediction['pi'] = prediction['pi'][w, options]
            prediction['log_pi'] = prediction['log_pi'][w, options]
            policy = torch.distributions.Categorical(probs=prediction['pi'])
            actions = policy.sample()
            next_states, rewards, terminals, info = self.task.step(to_np(actions))
            self.record_online_return(info)
            storage.feed(prediction)
            storage.feed({'reward': tensor(config.reward_normalizer(rewards)).unsqueeze(-1),
                          'mask': tensor(1 - terminals).unsqueeze(-1), 'option': options.unsqueeze(-1),
                          'prev_option': self.prev_options.unsqueeze(-1), 'entropy': policy.entropy().unsqueeze(-1),
                          'action': actions.unsqueeze(-1), 'init_state': self.is_initial_states.unsqueeze(-1).float(),
                          'eps': epsilon})
            self.is_initial_states = torch.as_tensor(np.asarray(terminals), device=Config.DEVICE).bool()
            self.prev_options = options
            self.states = config.state_normalizer(next_states)
            self.total_steps += config.num_workers
            if self.total_steps // config.num_workers % config.target_network_update_freq == 0:
                self._sync_target()
        with torch.no_grad():
            boot = self.target_network(self.states)
            storage.placeholder()
            beta = boot['beta'][w, self.prev_options]
            ret = ((1 - beta) * boot['q'][w, self.prev_options] + beta * boot['q'].max(dim=-1)[0]).unsqueeze(-1)
            for i in reversed(range(n)):
                q_i = storage.q[i].detach()
                ret = storage.reward[i] + config.discount * storage.mask[i] * ret
                storage.ret[i] = ret
                storage.advantage[i] = ret - q_i.gather(1, storage.option[i])
                v = q_i.max(dim=-1, keepdim=True)[0] * (1 - storage.eps[i]) + q_i.mean(-1).unsqueeze(-1) * storage.eps[i]
                storage.beta_advantage[i] = q_i.gather(1, storage.prev_option[i]) - v + config.termination_regularizer
        e = storage.extract(['q', 'beta', 'log_pi', 'ret', 'advantage', 'beta_advantage', 'entropy', 'option', 'action',
                             'init_state', 'prev_option'])
        q_loss = (e.q.gather(1, e.option) - e.ret).pow(2).mul(0.5).mean()
        pi_loss = (-(e.log_pi.gather(1, e.action) * e.advantage) - config.entropy_weight * e.entropy).mean()
        beta_loss = (e.beta.gather(1, e.prev_option) * e.beta_advantage * (1 - e.init_state)).mean()
        self._fused.zero_grad()
        (pi_loss + q_loss + beta_loss).backward()
        self._fused.step(config.gradient_clip)
