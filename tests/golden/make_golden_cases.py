"""Case tables + input streams shared by the golden generator and the tests
(kept free of any reference import so it loads on the GPU box)."""
import numpy as np


def stream(rs, t_len, state_shape, state_kind, action_dim, done_p):
    if state_kind == "u8":
        states = rs.randint(0, 256, size=(t_len,) + state_shape).astype(np.uint8)
    else:
        states = rs.uniform(-1, 1, size=(t_len,) + state_shape)
    actions = rs.randint(0, action_dim, size=t_len).astype(np.int64)
    rewards = np.sign(rs.choice([-2.0, 0.0, 3.0], size=t_len, p=[0.2, 0.6, 0.2]))
    masks = (1 - (rs.rand(t_len) < done_p)).astype(np.int32)
    return states, actions, rewards, masks


UNIFORM_CASES = [
    # name, memory, batch, H, n, discount, state_shape, kind, T, checkpoints
    ("a", 37, 8, 4, 1, 0.99, (6, 6), "u8", 100, [12, 36, 37, 41, 60, 74, 99]),
    ("b", 32, 6, 2, 3, 0.5, (3, 5), "u8", 90, [9, 31, 33, 40, 64, 89]),
    ("c", 20, 5, 1, 1, 0.99, (4,), "f64", 55, [3, 19, 20, 27, 54]),
    ("d", 64, 16, 4, 3, 0.99, (4, 4), "u8", 200, [20, 63, 64, 70, 128, 199]),
]

PER_CASES = [
    # name, memory, batch, H, n, discount, state_shape, kind, T, sample_every
    ("p", 50, 8, 4, 1, 0.99, (4, 4), "u8", 180, 7),
    ("q", 64, 16, 1, 3, 0.9, (4,), "f64", 200, 5),
    ("r", 33, 4, 2, 1, 0.99, (2, 2), "u8", 120, 3),
]

PIXEL_AGENT_CASES = [   # tag, agent, replay, n_step, done_period, agent steps
    ("dqn_b32", "dqn", "uniform", 1, 9, 24),
    ("c51_per", "c51", "per", 1, 7, 24),
    ("qr_n3", "qr", "uniform", 3, 11, 22),
    ("dqn_per", "dqn", "per", 1, 8, 24),        # round 4: the vanilla head's prioritized branch (DQN_agent.py:120-127)
]

# Dueling / Rainbow heads (network_heads.py:24-37,57-86; NoisyLinear network_utils.py:31-83): tensors the fixtures load from
# tests/fake_envs.numpy_params (strict=False for the noisy layers: their sigmas keep the constructor's constants)
_CONV = [("body.conv1.weight", (32, 4, 8, 8)), ("body.conv1.bias", (32,)), ("body.conv2.weight", (64, 32, 4, 4)),
         ("body.conv2.bias", (64,)), ("body.conv3.weight", (64, 64, 3, 3)), ("body.conv3.bias", (64,))]
DUELING_SHAPES = _CONV + [("body.fc4.weight", (512, 3136)), ("body.fc4.bias", (512,)), ("fc_value.weight", (1, 512)),
                          ("fc_value.bias", (1,)), ("fc_advantage.weight", (4, 512)), ("fc_advantage.bias", (4,))]
RAINBOW_SHAPES = _CONV + [("body.fc4.weight_mu", (512, 3136)), ("body.fc4.bias_mu", (512,)),
                          ("fc_value.weight_mu", (51, 512)), ("fc_value.bias_mu", (51,)),
                          ("fc_advantage.weight_mu", (4 * 51, 512)), ("fc_advantage.bias_mu", (4 * 51,))]
NOISY_LAYERS = ("body.fc4", "fc_value", "fc_advantage")
NOISE_BUFFERS = ("noise_in", "noise_out_weight", "noise_out_bias")


def head_inputs():
    """Seeded inputs of the module-level Dueling / Rainbow fixtures: uint8 batch [5,4,84,84] and the weights of the two
    linear functionals whose gradients are compared."""
    rs = np.random.RandomState(41)
    x = rs.randint(0, 256, size=(5, 4, 84, 84)).astype(np.uint8)
    return x, rs.standard_normal((5, 4)).astype(np.float32), rs.standard_normal((5, 4, 51)).astype(np.float32)


HEAD_AGENT_CASES = [   # tag, agent steps: rainbow_pixel's agent (examples.py:283-336) and DQNAgent over DuelingNet
    ("rainbow", 24),
    ("dueling", 22),
]


def trajectory_digest(state_dict, stride=4099):
    """Every `stride`-th element of every tensor of a state dict (fp32, in state-dict order): small enough to keep per
    update, so a run can be compared with the reference's after EVERY update and not only at the end."""
    return np.concatenate([np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32).reshape(-1)[::stride]
                           for v in state_dict.values()])


def digest(t, stride=1009):
    """Small stand-in for a multi-megabyte tensor: {fp64 sum, fp64 sum of squares} + every `stride`-th element."""
    a = np.asarray(t, dtype=np.float32).reshape(-1)
    return np.concatenate([[a.astype(np.float64).sum(), np.square(a.astype(np.float64)).sum()], a[::stride].astype(np.float64)])
