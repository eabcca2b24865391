"""Oracle: losses and return/advantage recurrences.  TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 CPU restatement of the reference's op chains (gradients come
from autograd on these chains).  All inputs are torch tensors on CPU.
"""
import torch


def dqn_td_error(q, q_next_target, actions, rewards, masks, gamma_n, q_next_online=None):
    """deep_rl/agent/DQN_agent.py:85-99.  `q` is the online net's [B,A] output for
    `states`; `q_next_target` the target net's output for `next_states` (no grad);
    double-Q gathers the target at the online argmax (:87-89), else max (:91).
    Returns delta = (r + gamma^n * q_next * mask) - q[a]   (shape [B])."""
    q_next = q_next_target.detach()
    if q_next_online is not None:
        best = torch.argmax(q_next_online.detach(), dim=-1)
        q_next = q_next.gather(1, best.unsqueeze(-1)).squeeze(1)
    else:
        q_next = q_next.max(1)[0]
    q_target = rewards + gamma_n * q_next * masks
    q_a = q.gather(1, actions.long().unsqueeze(-1)).squeeze(-1)
    return q_target - q_a


def dqn_reduce(delta):
    """DQN_agent.py:78-79: mean(0.5 * delta^2) -- plain MSE, not Huber."""
    return delta.pow(2).mul(0.5).mean()


def per_priorities_and_weights(loss_vec, sampling_prob, replay_eps, replay_alpha, beta):
    """DQN_agent.py:120-127: priorities from the PRE-weight loss vector; importance
    weights use the BATCH size; returns (priorities, weights, weighted loss_vec)."""
    prio = loss_vec.abs().add(replay_eps).pow(replay_alpha)
    w = sampling_prob.mul(sampling_prob.size(0)).add(1e-6).pow(-beta)
    w = w / w.max()
    return prio, w, loss_vec.mul(w)


def c51_kl(log_prob, prob_next_target, actions, rewards, masks, gamma_n, atoms, v_min, v_max,
           prob_next_online=None):
    """deep_rl/agent/CategoricalDQN_agent.py:60-86.  log_prob [B,A,N] (online,
    states), prob_next_target [B,A,N]; returns KL [B]; reduce = mean (:88-89)."""
    n_atoms = atoms.numel()
    delta_atom = (v_max - v_min) / float(n_atoms - 1)
    b = torch.arange(log_prob.size(0))
    with torch.no_grad():
        prob_next = prob_next_target
        q_next = (prob_next * atoms).sum(-1)
        if prob_next_online is not None:
            a_next = torch.argmax((prob_next_online * atoms).sum(-1), dim=-1)
        else:
            a_next = torch.argmax(q_next, dim=-1)
        prob_next = prob_next[b, a_next, :]
    r = rewards.unsqueeze(-1)
    m = masks.unsqueeze(-1)
    tz = r + gamma_n * m * atoms.view(1, -1)
    tz = tz.clamp(v_min, v_max).unsqueeze(1)
    target = (1 - (tz - atoms.view(1, -1, 1)).abs() / delta_atom).clamp(0, 1) * prob_next.unsqueeze(1)
    target = target.sum(-1)
    lp = log_prob[b, actions.long(), :]
    return (target * target.add(1e-5).log() - target * lp).sum(-1)


def huber(x, k=1.0):
    """deep_rl/utils/torch_utils.py:47-48."""
    return torch.where(x.abs() < k, 0.5 * x.pow(2), k * (x.abs() - 0.5 * k))


def qr_loss(quantiles, quantiles_next_target, actions, rewards, masks, gamma_n):
    """deep_rl/agent/QuantileRegressionDQN_agent.py:55-74.  quantiles [B,A,N]
    (online, states), quantiles_next_target [B,A,N].  Returns the [N]-vector
    loss.sum(-1).mean(1) (index = TARGET quantile j); reduce = mean (:76-77)."""
    n = quantiles.size(-1)
    b = torch.arange(quantiles.size(0))
    tau = ((2 * torch.arange(n, dtype=torch.float64) + 1) / (2.0 * n)).float().view(1, -1)
    qn = quantiles_next_target.detach()
    a_next = torch.argmax(qn.sum(-1), dim=-1)
    qn = qn[b, a_next, :]
    qn = rewards.unsqueeze(-1) + gamma_n * masks.unsqueeze(-1) * qn
    theta = quantiles[b, actions.long(), :]
    diff = qn.t().unsqueeze(-1) - theta  # [N_j, B, N_i]
    loss = huber(diff) * (tau - (diff.detach() < 0).float()).abs()
    return loss.sum(-1).mean(1)


def gae_reverse(rewards, masks, values, gamma, tau, use_gae):
    """deep_rl/agent/PPO_agent.py:51-61 == A2C_agent.py:43-53 (NStepDQN_agent.py:56-60
    is the use_gae=False / returns-only case).  rewards, masks: [T,N,1]; values:
    [T+1,N,1] (values[T] bootstraps).  Returns (advantages, returns) [T,N,1]."""
    t_len = rewards.size(0)
    adv = torch.zeros_like(rewards[0])
    ret = values[t_len].detach()
    advs, rets = [None] * t_len, [None] * t_len
    for i in range(t_len - 1, -1, -1):
        ret = rewards[i] + gamma * masks[i] * ret
        if not use_gae:
            adv = ret - values[i].detach()
        else:
            td = rewards[i] + gamma * masks[i] * values[i + 1] - values[i]
            adv = adv * tau * gamma * masks[i] + td
        advs[i] = adv.detach()
        rets[i] = ret.detach()
    return torch.stack(advs), torch.stack(rets)


def normalize_advantage(adv):
    """PPO_agent.py:66: (a - mean) / std with torch's default unbiased std."""
    return (adv - adv.mean()) / adv.std()


def ppo_losses(log_pi_a, entropy, v, old_log_pi_a, adv, ret, clip, entropy_weight):
    """PPO_agent.py:77-86.  Returns (policy_loss, value_loss, approx_kl)."""
    ratio = (log_pi_a - old_log_pi_a).exp()
    obj = ratio * adv
    obj_clipped = ratio.clamp(1.0 - clip, 1.0 + clip) * adv
    policy_loss = -torch.min(obj, obj_clipped).mean() - entropy_weight * entropy.mean()
    value_loss = 0.5 * (ret - v).pow(2).mean()
    approx_kl = (old_log_pi_a - log_pi_a).mean()
    return policy_loss, value_loss, approx_kl


def a2c_loss(log_pi_a, entropy, v, adv, ret, entropy_weight, value_loss_weight):
    """A2C_agent.py:55-62."""
    policy_loss = -(log_pi_a * adv).mean()
    value_loss = 0.5 * (ret - v).pow(2).mean()
    entropy_loss = entropy.mean()
    return policy_loss - entropy_weight * entropy_loss + value_loss_weight * value_loss
