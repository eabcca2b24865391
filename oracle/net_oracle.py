"""Oracle: network forward passes and the optimiser step.  TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 CPU ops on explicit parameter dicts whose keys are the
reference modules' state_dict keys, so fixtures interchange with the reference.
"""
import math

import torch
import torch.nn.functional as F


def orthogonal_init_(weight, w_scale=1.0):
    """deep_rl/network/network_utils.py:23-27 (weight part; bias is zeroed)."""
    torch.nn.init.orthogonal_(weight)
    weight.mul_(w_scale)
    return weight


def nature_conv_body(p, x, prefix="body."):
    """deep_rl/network/network_bodies.py:27-33: conv(4->32,k8,s4) relu conv(32->64,k4,s2)
    relu conv(64->64,k3,s1) relu flatten fc(3136->512) relu."""
    y = F.relu(F.conv2d(x, p[prefix + "conv1.weight"], p[prefix + "conv1.bias"], stride=4))
    y = F.relu(F.conv2d(y, p[prefix + "conv2.weight"], p[prefix + "conv2.bias"], stride=2))
    y = F.relu(F.conv2d(y, p[prefix + "conv3.weight"], p[prefix + "conv3.bias"], stride=1))
    y = y.reshape(y.size(0), -1)
    return F.relu(F.linear(y, p[prefix + "fc4.weight"], p[prefix + "fc4.bias"]))


def nature_conv_body_margin(p, x, prefix="body."):
    """nature_conv_body plus the smallest |pre-activation| over its four ReLU layers.  A ReLU whose input lies within
    fp32 summation noise of zero (~1e-7 of the operand scale) can be gated differently by two correct fp32
    implementations; its whole backward contribution then differs.  Parity tests use the margin to tell such a step
    (expected about once per 3-4 batch-32 updates: 7e5 gates, density ~2 per unit at zero) from a real mismatch."""
    margin = float("inf")
    y = x
    for name, stride in (("conv1", 4), ("conv2", 2), ("conv3", 1)):
        pre = F.conv2d(y, p[prefix + name + ".weight"], p[prefix + name + ".bias"], stride=stride)
        margin = min(margin, float(pre.detach().abs().min()))
        y = F.relu(pre)
    y = y.reshape(y.size(0), -1)
    pre = F.linear(y, p[prefix + "fc4.weight"], p[prefix + "fc4.bias"])
    margin = min(margin, float(pre.detach().abs().min()))
    return F.relu(pre), margin


def fc_body(p, x, n_layers, gate=F.relu, prefix="body."):
    """deep_rl/network/network_bodies.py:70-73."""
    for i in range(n_layers):
        x = gate(F.linear(x, p["%slayers.%d.weight" % (prefix, i)], p["%slayers.%d.bias" % (prefix, i)]))
    return x


def vanilla_head(p, phi):
    """deep_rl/network/network_heads.py:18-21."""
    return F.linear(phi, p["fc_head.weight"], p["fc_head.bias"])


def categorical_head(p, phi, action_dim, num_atoms):
    """deep_rl/network/network_heads.py:49-54: returns (prob, log_prob) [B,A,N]."""
    pre = F.linear(phi, p["fc_categorical.weight"], p["fc_categorical.bias"]).view(-1, action_dim, num_atoms)
    return F.softmax(pre, dim=-1), F.log_softmax(pre, dim=-1)


def quantile_head(p, phi, action_dim, num_quantiles):
    """deep_rl/network/network_heads.py:98-102."""
    return F.linear(phi, p["fc_quantiles.weight"], p["fc_quantiles.bias"]).view(-1, action_dim, num_quantiles)


def dueling_head(p, phi):
    """deep_rl/network/network_heads.py:32-37: q = value + (advantage - mean_a advantage)."""
    value = F.linear(phi, p["fc_value.weight"], p["fc_value.bias"])
    adv = F.linear(phi, p["fc_advantage.weight"], p["fc_advantage.bias"])
    return value.expand_as(adv) + (adv - adv.mean(1, keepdim=True).expand_as(adv))


def transform_noise(x):
    """deep_rl/network/network_utils.py:82-83: f(e) = sign(e) sqrt|e|."""
    return x.sign().mul(x.abs().sqrt())


def noisy_epsilon(noise_in, noise_out_weight, noise_out_bias):
    """deep_rl/network/network_utils.py:78-80: (weight_epsilon, bias_epsilon) from the three drawn vectors."""
    return transform_noise(noise_out_weight).ger(transform_noise(noise_in)), transform_noise(noise_out_bias)


def noisy_linear(p, x, prefix, training=True):
    """deep_rl/network/network_utils.py:54-62: W = mu + sigma * eps (training) or mu (evaluation)."""
    if training:
        w = p[prefix + "weight_mu"] + p[prefix + "weight_sigma"].mul(p[prefix + "weight_epsilon"])
        b = p[prefix + "bias_mu"] + p[prefix + "bias_sigma"].mul(p[prefix + "bias_epsilon"])
    else:
        w, b = p[prefix + "weight_mu"], p[prefix + "bias_mu"]
    return F.linear(x, w, b)


def nature_conv_body_noisy(p, x, training=True, prefix="body."):
    """deep_rl/network/network_bodies.py:27-33 with fc4 = NoisyLinear (noisy_linear=True, :21-24)."""
    y = F.relu(F.conv2d(x, p[prefix + "conv1.weight"], p[prefix + "conv1.bias"], stride=4))
    y = F.relu(F.conv2d(y, p[prefix + "conv2.weight"], p[prefix + "conv2.bias"], stride=2))
    y = F.relu(F.conv2d(y, p[prefix + "conv3.weight"], p[prefix + "conv3.bias"], stride=1))
    return F.relu(noisy_linear(p, y.reshape(y.size(0), -1), prefix + "fc4.", training))


def rainbow_head(p, phi, action_dim, num_atoms, training=True):
    """deep_rl/network/network_heads.py:78-86 with NoisyLinear heads: (prob, log_prob) [B,A,N] of
    value + (advantage - mean_a advantage)."""
    value = noisy_linear(p, phi, "fc_value.", training).view(-1, 1, num_atoms)
    adv = noisy_linear(p, phi, "fc_advantage.", training).view(-1, action_dim, num_atoms)
    q = value + (adv - adv.mean(1, keepdim=True))
    return F.softmax(q, dim=-1), F.log_softmax(q, dim=-1)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (call site DQN_agent.py:132): global L2 norm
    over all grads (per-tensor 2-norms, then the 2-norm of those, as torch does);
    scale by max_norm/(norm+1e-6) when that is < 1.  Returns the norm and the scaled
    grads.  fp32 summation order differs between torch versions / thread counts, so
    callers compare the norm at 5e-5 relative."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.detach().float()) for g in grads]))
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        grads = [g * coef for g in grads]
    return total, grads


def rmsprop_step(param, grad, square_avg, grad_avg, lr, alpha, eps, centered):
    """torch.optim.RMSprop (torch==1.5.1 per requirements.txt:1; call sites
    examples.py:67-68,370): sq <- a*sq + (1-a)*g^2;  centered: ga <- a*ga + (1-a)*g,
    avg = sqrt(sq - ga^2) + eps, else avg = sqrt(sq) + eps;  p <- p - lr * g/avg."""
    square_avg = alpha * square_avg + (1 - alpha) * grad * grad
    if centered:
        grad_avg = alpha * grad_avg + (1 - alpha) * grad
        avg = (square_avg - grad_avg * grad_avg).sqrt() + eps
    else:
        avg = square_avg.sqrt() + eps
    return param - lr * grad / avg, square_avg, grad_avg


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps):
    """torch.optim.Adam (call sites examples.py:139,204,508-509,534): `step` is the
    1-based step count AFTER increment."""
    exp_avg = beta1 * exp_avg + (1 - beta1) * grad
    exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = exp_avg_sq.sqrt() / math.sqrt(bc2) + eps
    return param - (lr / bc1) * exp_avg / denom, exp_avg, exp_avg_sq
