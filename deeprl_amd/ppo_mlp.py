"""Host side of csrc/ppo_mlp.hip: PPO over GaussianActorCriticNet with two small tanh MLPs (BASELINE configs[2],
examples.py:497-523) as persistent kernels.

`MlpPPO(agent)` is what PPOAgent.optimize() hands a rollout's entries to when the configuration is one the kernels
implement (`MlpPPO.eligible`): it draws the epochs' np.random permutations exactly as PPO_agent.py:72 / misc.py:55-62 do
(one permutation per epoch, nothing else consumes np.random in between), gathers every minibatch once (dra_ppo_mlp_pack) and
runs ALL epochs x minibatches in one launch (dra_ppo_mlp_update) on the optimisers' own flat buffers, so that
state_dict()s, checkpoints and the generic path see the same parameters and Adam state afterwards.

`rollout(...)` is the device-environment rollout (device_env.DeviceContinuousVec): one launch per PPO_agent.py:32-49.
There is no CPU / eager implementation here: without the HIP library every call raises.
"""
import ctypes

import numpy as np
import torch

from ._lib import DraError, lib, ptr, stream_ptr
from .support import Config

DBG_FLOATS = 65536          # DRA_PPO_MLP_DBG_FLOATS


class Net(ctypes.Structure):
    """Mirror of dra_ppo_mlp_net (include/deeprl_amd.h)."""
    _fields_ = [("param", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("step_dev", ctypes.c_void_p),
                ("off_w1", ctypes.c_int32), ("off_b1", ctypes.c_int32), ("off_w2", ctypes.c_int32), ("off_b2", ctypes.c_int32),
                ("off_w3", ctypes.c_int32), ("off_b3", ctypes.c_int32), ("off_std", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float)]


class Cfg(ctypes.Structure):
    """Mirror of dra_ppo_mlp_cfg."""
    _fields_ = [("state_dim", ctypes.c_int32), ("action_dim", ctypes.c_int32), ("hidden", ctypes.c_int32),
                ("mini_batch", ctypes.c_int32), ("ratio_clip", ctypes.c_float), ("entropy_weight", ctypes.c_float),
                ("kl_limit", ctypes.c_double)]


class RolloutIO(ctypes.Structure):
    """Mirror of dra_ppo_mlp_rollout_io."""
    _fields_ = [("env_state", ctypes.c_void_p), ("env_counter", ctypes.c_void_p), ("env_seed", ctypes.c_void_p),
                ("rms", ctypes.c_void_p), ("cur_state", ctypes.c_void_p), ("sampler_step", ctypes.c_void_p),
                ("out_state", ctypes.c_void_p), ("out_action", ctypes.c_void_p), ("out_log_pi_a", ctypes.c_void_p),
                ("out_v", ctypes.c_void_p), ("out_reward", ctypes.c_void_p), ("out_mask", ctypes.c_void_p),
                ("env0", ctypes.c_int64), ("n_global", ctypes.c_int64), ("noise_seed", ctypes.c_uint64),
                ("horizon", ctypes.c_int64), ("reward_coef", ctypes.c_double), ("rms_epsilon", ctypes.c_double),
                ("rms_clip", ctypes.c_double), ("rms_update", ctypes.c_int32), ("t_len", ctypes.c_int32),
                ("n_env", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def _mlp_shape(network):
    """(state_dim, action_dim, hidden) when `network` is a GaussianActorCriticNet whose phi_body is the identity and whose
    actor / critic bodies are two-layer tanh FCBody stacks of one width; else None."""
    from .nets import DummyBody, FCBody, GaussianActorCriticNet, Linear
    if type(network) is not GaussianActorCriticNet or type(network.phi_body) is not DummyBody:
        return None
    bodies = (network.actor_body, network.critic_body)
    for b in bodies:
        if type(b) is not FCBody or b.noisy_linear or b.gate is not torch.tanh or len(b.layers) != 2:
            return None
        if any(type(layer) is not Linear or layer.bias is None for layer in b.layers):
            return None
    a, c = bodies
    s_dim = a.layers[0].weight.shape[1]
    hidden = a.layers[0].weight.shape[0]
    shapes_ok = all(tuple(b.layers[0].weight.shape) == (hidden, s_dim) and tuple(b.layers[1].weight.shape) == (hidden, hidden)
                    for b in bodies)
    if not shapes_ok or network.fc_action.weight.shape[1] != hidden or tuple(network.fc_critic.weight.shape) != (1, hidden):
        return None
    if network.fc_action.bias is None or network.fc_critic.bias is None:
        return None
    return int(s_dim), int(network.fc_action.weight.shape[0]), int(hidden)


def _net_struct(fused, body, head, std, step_dev):
    """dra_ppo_mlp_net over a FusedOptimizer's flat buffers."""
    flat = fused.flat
    b1, b2 = fused.hyper['betas']
    n = Net()
    n.param, n.exp_avg, n.exp_avg_sq = flat.flat.data_ptr(), fused.state1.data_ptr(), fused.state2.data_ptr()
    n.step_dev = step_dev.data_ptr()
    n.off_w1, n.off_b1 = flat.offset_of(body.layers[0].weight), flat.offset_of(body.layers[0].bias)
    n.off_w2, n.off_b2 = flat.offset_of(body.layers[1].weight), flat.offset_of(body.layers[1].bias)
    n.off_w3, n.off_b3 = flat.offset_of(head.weight), flat.offset_of(head.bias)
    n.off_std = flat.offset_of(std) if std is not None else -1
    n.lr, n.beta1, n.beta2, n.eps = float(fused.hyper['lr']), float(b1), float(b2), float(fused.hyper['eps'])
    return n


class MlpPPO:
    """The persistent-kernel form of PPOAgent.optimize() (PPO_agent.py:71-99, shared_repr False)."""

    def __init__(self, agent):
        self.agent = agent
        self.shape = None
        self.failed = False
        self._steps = None          # device int64 [2]: Adam step counts of the two optimisers
        self._packed = None
        self._perm = None
        self._perm_up = None
        self.out3 = None
        self.counts = None
        self.launches = 0

    @staticmethod
    def eligible(agent):
        cfg = agent.config
        if cfg.shared_repr or Config.DEVICE.type != 'cuda' or getattr(cfg, 'fused_ppo_mlp', True) is False:
            return None
        if agent.dp.active:
            return None
        shape = _mlp_shape(agent.network)
        if shape is None:
            return None
        if lib.dra_ppo_mlp_supported.raw(shape[0], shape[1], shape[2], shape[2], int(cfg.mini_batch_size)) != 0:
            return None
        opts = (agent._fused_actor, agent._fused_critic)
        if any(o.kind != 'adam' for o in opts):
            return None
        net = agent.network
        expect_a = [net.actor_body.layers[0].weight, net.actor_body.layers[0].bias, net.actor_body.layers[1].weight,
                    net.actor_body.layers[1].bias, net.fc_action.weight, net.fc_action.bias, net.std]
        expect_c = [net.critic_body.layers[0].weight, net.critic_body.layers[0].bias, net.critic_body.layers[1].weight,
                    net.critic_body.layers[1].bias, net.fc_critic.weight, net.fc_critic.bias]
        for o, expect in zip(opts, (expect_a, expect_c)):
            if len(o.flat.params) != len(expect) or any(all(p is not q for q in o.flat.params) for p in expect):
                return None
        return shape

    def usable(self):
        a = self.agent
        if self.failed or a.grad_hook is not None:
            return False
        if self.shape is None:
            self.shape = self.eligible(a) or False
        return bool(self.shape)

    def structs(self):
        a = self.agent
        net = a.network
        dev = Config.DEVICE
        if self._steps is None:
            self._steps = torch.zeros(2, dtype=torch.int64, device=dev)
            self.out3 = torch.zeros(3, dtype=torch.float32, device=dev)
            self.counts = torch.zeros(2, dtype=torch.int64, device=dev)
        s_dim, a_dim, hidden = self.shape
        cfg = Cfg()
        cfg.state_dim, cfg.action_dim, cfg.hidden, cfg.mini_batch = s_dim, a_dim, hidden, int(a.config.mini_batch_size)
        cfg.ratio_clip, cfg.entropy_weight = float(a.config.ppo_ratio_clip), float(a.config.entropy_weight)
        cfg.kl_limit = 1.5 * float(a.config.target_kl)
        actor = _net_struct(a._fused_actor, net.actor_body, net.fc_action, net.std, self._steps[0:1])
        critic = _net_struct(a._fused_critic, net.critic_body, net.fc_critic, None, self._steps[1:2])
        return cfg, actor, critic

    def optimize(self, entries, dbg=None):
        """All epochs x minibatches of one rollout.  Returns False (nothing done) when the entries are not what the kernels
        take; the caller then runs the generic path."""
        a = self.agent
        cfg_a = a.config
        n = int(entries.state.size(0))
        s_dim, a_dim, hidden = self.shape
        if tuple(entries.state.shape) != (n, s_dim) or tuple(entries.action.shape) != (n, a_dim) or n < 1:
            return False
        if any(x.dtype != torch.float32 or not x.is_cuda for x in entries):
            return False
        epochs, mb = int(cfg_a.optimization_epochs), int(cfg_a.mini_batch_size)
        dev = entries.state.device
        # PPO_agent.py:72 -> misc.py:55-62: one np.random.permutation per epoch (random_sample draws it when the epoch starts;
        # no other consumer of np.random sits between the epochs, so drawing them together leaves the stream where it was)
        if self._perm is None or self._perm.numel() != epochs * n:
            self._perm = torch.zeros(epochs * n, dtype=torch.int64, device=dev)
            self._perm_up = [torch.zeros(epochs * n, dtype=torch.int64).pin_memory() for _ in range(2)]
            self._perm_ev = [None, None]
            self._perm_k = 0
        k = self._perm_k
        self._perm_k = 1 - k
        if self._perm_ev[k] is not None:
            self._perm_ev[k].synchronize()
        stage = self._perm_up[k].numpy()
        for e in range(epochs):
            stage[e * n:(e + 1) * n] = a.dp.permutation(n)
        self._perm.copy_(self._perm_up[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._perm_ev[k] = ev
        floats = ctypes.c_int64()
        lib.dra_ppo_mlp_packed_floats(n, epochs, mb, s_dim, ctypes.byref(floats))
        if self._packed is None or self._packed.numel() != floats.value:
            self._packed = torch.empty(floats.value, dtype=torch.float32, device=dev)
        st = stream_ptr()
        cont = [x if x.is_contiguous() else x.contiguous() for x in entries]
        state, action, log_pi_a, ret, adv = cont      # Storage.extract order: state, action, log_pi_a, ret, advantage
        lib.dra_ppo_mlp_pack(ptr(state), ptr(action), ptr(log_pi_a), ptr(adv), ptr(ret), ptr(self._perm), n, epochs, mb, s_dim,
                             a_dim, ptr(self._packed), st)
        cfg, actor, critic = self.structs()
        # Adam step counts: the device pair is authoritative while launches follow each other (the actor's count depends on
        # the KL gate); the optimisers' host counts are uploaded when the generic path (or a fresh agent) had the last word
        if not getattr(a, '_pending_counts', False):
            self._steps.copy_(torch.tensor([a._fused_actor.steps, a._fused_critic.steps], dtype=torch.int64))
        lib.dra_ppo_mlp_update(ctypes.byref(cfg), ctypes.byref(actor), ctypes.byref(critic), ptr(self._packed), n, epochs,
                               ptr(self.out3), ptr(self.counts), ptr(dbg), st)
        self.launches += 1
        a._pending_counts = True
        a.last_loss = self.out3
        return True

    def sync_counts(self):
        """Adam step counts back to the host optimisers (one small D2H; called lazily -- before the generic path, a checkpoint
        or the next launch needs them)."""
        a = self.agent
        if getattr(a, '_pending_counts', False):
            s = self._steps.cpu()
            a._fused_actor.steps, a._fused_critic.steps = int(s[0]), int(s[1])
            a._pending_counts = False


# ------------------------------------------------------------------------------------------ stand-alone pieces
def rms_normalize(x, mean, var, count, update=True, epsilon=1e-8, clip=10.0, out_f32=True, out_f64=False):
    """normalizer.py:28-51 on the device: x f64 [n, d]; mean / var f64 [d] and count f64 [1] are updated in place when
    `update`.  Returns (f32 result or None, f64 result or None)."""
    if x.dtype != torch.float64 or not x.is_cuda or x.dim() != 2:
        raise DraError("rms_normalize takes a float64 device tensor [n, d]")
    x = x if x.is_contiguous() else x.contiguous()
    n, d = x.shape
    o32 = torch.empty((n, d), dtype=torch.float32, device=x.device) if out_f32 else None
    o64 = torch.empty((n, d), dtype=torch.float64, device=x.device) if out_f64 else None
    lib.dra_rms_normalize(ptr(x), n, d, ptr(mean), ptr(var), ptr(count), 1 if update else 0, float(epsilon), float(clip),
                          ptr(o32), ptr(o64), stream_ptr())
    return o32, o64


def gauss_sample(mean, scale, noise_seed, step_dev, n_global=None, env0=0):
    """mean [n, A] + scale [A] * hashed standard normals (csrc/cont_env.h gauss_noise); advances step_dev by one."""
    mean = mean if mean.is_contiguous() else mean.contiguous()
    scale = scale if scale.is_contiguous() else scale.contiguous()
    n, a_dim = mean.shape
    out = torch.empty_like(mean)
    lib.dra_gauss_sample(ptr(mean), ptr(scale), n, a_dim, int(noise_seed), ptr(step_dev), int(n_global if n_global else n),
                         int(env0), ptr(out), stream_ptr())
    return out


def cont_env_step(state, counter, seed, action, horizon):
    """One step of n device-resident SyntheticContinuous environments; returns (reward f64 [n], done i32 [n])."""
    n, s_dim = state.shape
    action = action if action.is_contiguous() else action.contiguous()
    reward = torch.empty(n, dtype=torch.float64, device=state.device)
    done = torch.empty(n, dtype=torch.int32, device=state.device)
    lib.dra_cont_env_step(ptr(state), ptr(counter), ptr(seed), ptr(action), n, s_dim, action.shape[1], int(horizon), ptr(reward),
                          ptr(done), stream_ptr())
    return reward, done
