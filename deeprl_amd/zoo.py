"""Ready-made experiment configurations: the hyper-parameters of the reference's examples.py entry points, as DATA.

`ZOO[name]` holds what examples.py sets field by field (cited per entry); `config(name, **kw)` turns an entry into a
`Config`, `agent(name, **kw)` builds the agent, `run(name, **kw)` is `run_steps(agent)`.  Module-level functions with
the reference's names (`dqn_pixel(game=...)`, ...) make this file usable wherever examples.py is
(`python -m deeprl_amd.launch deeprl_amd/zoo.py dqn_pixel game=BreakoutNoFrameskip-v4`).

Keyword arguments are merged into the Config first, like examples.py does (`config.merge(kwargs)`); two extras are
applied LAST so that short runs are possible without editing a table: `max_steps=` and `overrides={field: value}`.
tests/test_zoo_vs_reference.py checks every entry against the Config the reference's own function builds.
"""
import torch
import torch.nn.functional as F

# absolute imports: this file is also executed as a plain examples file by deeprl_amd.launch
from deeprl_amd import agents as A, nets as N
from deeprl_amd.envs import Task
from deeprl_amd.normalizers import ImageNormalizer, MeanStdNormalizer, SignNormalizer
from deeprl_amd.replay import ReplayWrapper, UniformReplay
from deeprl_amd.support import Config, LinearSchedule, generate_tag, run_steps


def _rmsprop(**kw):
    return lambda params: torch.optim.RMSprop(params, **kw)


def _adam(**kw):
    return lambda params: torch.optim.Adam(params, **kw)


# name -> spec.  `kw`: kwargs.setdefault(...) of the entry point; `fields`: plain Config fields; the callables receive
# the Config under construction (they are evaluated lazily, exactly where examples.py uses a lambda).
ZOO = {
    # examples.py:11-52
    "dqn_feature": dict(
        agent="DQNAgent", kw=dict(log_level=0, n_step=1, replay_cls=UniformReplay, async_replay=True),
        task=lambda c: Task(c.game), eval_task="same", optimizer=_rmsprop(lr=0.001),
        network=lambda c: N.VanillaNet(c.action_dim, N.FCBody(c.state_dim)),
        fields=dict(history_length=1, batch_size=10, discount=0.99, max_steps=1e5, replay_eps=0.01, replay_alpha=0.5,
                    target_network_update_freq=200, exploration_steps=1000, double_q=False, sgd_update_frequency=4,
                    gradient_clip=5, eval_interval=int(5e3), async_actor=False),
        replay=dict(memory_size=int(1e4), with_n_step=True), eps=(1.0, 0.1, 1e4), beta=(0.4, 1.0)),
    # examples.py:55-97
    "dqn_pixel": dict(
        agent="DQNAgent", kw=dict(log_level=0, n_step=1, replay_cls=UniformReplay, async_replay=True),
        task=lambda c: Task(c.game), eval_task="same",
        optimizer=_rmsprop(lr=0.00025, alpha=0.95, eps=0.01, centered=True),
        network=lambda c: N.VanillaNet(c.action_dim, N.NatureConvBody(in_channels=c.history_length)),
        fields=dict(batch_size=32, discount=0.99, history_length=4, max_steps=int(2e7), replay_eps=0.01, replay_alpha=0.5,
                    target_network_update_freq=10000, exploration_steps=50000, sgd_update_frequency=4, gradient_clip=5,
                    double_q=False, async_actor=True),
        normalizers=(ImageNormalizer, SignNormalizer),
        replay=dict(memory_size=int(1e6), with_n_step=True), eps=(1.0, 0.01, 1e6), beta=(0.4, 1.0)),
    # examples.py:129-161
    "quantile_regression_dqn_pixel": dict(
        agent="QuantileRegressionDQNAgent", kw=dict(log_level=0), task=lambda c: Task(c.game), eval_task="same",
        optimizer=_adam(lr=0.00005, eps=0.01 / 32),
        network=lambda c: N.QuantileNet(c.action_dim, c.num_quantiles, N.NatureConvBody()),
        fields=dict(batch_size=32, discount=0.99, target_network_update_freq=10000, exploration_steps=50000,
                    sgd_update_frequency=4, gradient_clip=5, num_quantiles=200, max_steps=int(2e7)),
        normalizers=(ImageNormalizer, SignNormalizer),
        replay=dict(memory_size=int(1e6), history_length=4, fixed_cls=UniformReplay, fixed_async=True), eps=(1.0, 0.01, 1e6)),
    # examples.py:196-227
    "categorical_dqn_pixel": dict(
        agent="CategoricalDQNAgent", kw=dict(log_level=0), task=lambda c: Task(c.game), eval_task="same",
        optimizer=_adam(lr=0.00025, eps=0.01 / 32),
        network=lambda c: N.CategoricalNet(c.action_dim, c.categorical_n_atoms, N.NatureConvBody()),
        fields=dict(batch_size=32, discount=0.99, target_network_update_freq=10000, exploration_steps=50000,
                    categorical_v_max=10, categorical_v_min=-10, categorical_n_atoms=51, sgd_update_frequency=4,
                    gradient_clip=0.5, max_steps=int(2e7)),
        normalizers=(ImageNormalizer, SignNormalizer),
        replay=dict(memory_size=int(1e6), history_length=4, fixed_cls=UniformReplay, fixed_async=True), eps=(1.0, 0.01, 1e6)),
    # examples.py:361-381
    "a2c_pixel": dict(
        agent="A2CAgent", kw=dict(log_level=0), pre_fields=dict(num_workers=16),
        task=lambda c: Task(c.game, num_envs=c.num_workers), eval_task=lambda c: Task(c.game),
        optimizer=_rmsprop(lr=1e-4, alpha=0.99, eps=1e-5),
        network=lambda c: N.CategoricalActorCriticNet(c.state_dim, c.action_dim, N.NatureConvBody()),
        fields=dict(discount=0.99, use_gae=True, gae_tau=1.0, entropy_weight=0.01, rollout_length=5, gradient_clip=5,
                    max_steps=int(2e7)),
        normalizers=(ImageNormalizer, SignNormalizer)),
    # examples.py:525-550
    "ppo_pixel": dict(
        agent="PPOAgent", kw=dict(skip=False), pre_fields=dict(num_workers=8),
        task=lambda c: Task(c.game, num_envs=c.num_workers), eval_task=lambda c: Task(c.game),
        optimizer=_adam(lr=2.5e-4),
        network=lambda c: N.CategoricalActorCriticNet(c.state_dim, c.action_dim, N.NatureConvBody()),
        fields=dict(discount=0.99, use_gae=True, gae_tau=0.95, entropy_weight=0.01, gradient_clip=0.5, rollout_length=128,
                    optimization_epochs=4, ppo_ratio_clip=0.1, shared_repr=True, max_steps=int(2e7)),
        derived=lambda c: dict(mini_batch_size=c.rollout_length * c.num_workers // 4,
                               log_interval=c.rollout_length * c.num_workers),
        normalizers=(ImageNormalizer, SignNormalizer)),
    # examples.py:494-522
    "ppo_continuous": dict(
        agent="PPOAgent", kw=dict(log_level=0), task=lambda c: Task(c.game), eval_task="same",
        network=lambda c: N.GaussianActorCriticNet(c.state_dim, c.action_dim, actor_body=N.FCBody(c.state_dim, gate=torch.tanh),
                                                   critic_body=N.FCBody(c.state_dim, gate=torch.tanh)),
        actor_opt=_adam(lr=3e-4), critic_opt=_adam(lr=1e-3),
        fields=dict(discount=0.99, use_gae=True, gae_tau=0.95, gradient_clip=0.5, rollout_length=2048, optimization_epochs=10,
                    mini_batch_size=64, ppo_ratio_clip=0.2, log_interval=2048, max_steps=3e6, target_kl=0.01),
        normalizers=(MeanStdNormalizer, None)),
}


def config(name, **kwargs):
    spec = ZOO[name]
    max_steps = kwargs.pop("max_steps", None)
    overrides = kwargs.pop("overrides", None) or {}
    generate_tag(kwargs)
    for k, v in spec.get("kw", {}).items():
        kwargs.setdefault(k, v)
    c = Config()
    c.merge(kwargs)
    for k, v in spec.get("pre_fields", {}).items():
        setattr(c, k, v)
    c.task_fn = lambda: spec["task"](c)
    c.eval_env = c.task_fn() if spec.get("eval_task") == "same" else spec["eval_task"](c)
    if "optimizer" in spec:
        c.optimizer_fn = spec["optimizer"]
    if "actor_opt" in spec:
        c.actor_opt_fn, c.critic_opt_fn = spec["actor_opt"], spec["critic_opt"]
    c.network_fn = lambda: spec["network"](c)
    for k, v in spec.get("fields", {}).items():
        setattr(c, k, v)
    if "derived" in spec:
        for k, v in spec["derived"](c).items():
            setattr(c, k, v)
    if "eps" in spec:
        c.random_action_prob = LinearSchedule(*spec["eps"])
    if "beta" in spec:
        c.replay_beta = LinearSchedule(spec["beta"][0], spec["beta"][1], c.max_steps)
    norm = spec.get("normalizers")
    if norm:
        if norm[0] is not None:
            c.state_normalizer = norm[0]()
        if norm[1] is not None:
            c.reward_normalizer = norm[1]()
    rp = spec.get("replay")
    if rp:
        kw = dict(memory_size=rp["memory_size"], batch_size=c.batch_size)
        if rp.get("with_n_step"):
            kw.update(n_step=c.n_step, discount=c.discount, history_length=c.history_length)
        else:
            kw.update(history_length=rp["history_length"])
        cls = rp.get("fixed_cls") or c.replay_cls
        flag = rp["fixed_async"] if "fixed_async" in rp else c.async_replay
        c.replay_kwargs = kw
        c.replay_fn = lambda: ReplayWrapper(cls, kw, flag)
    if max_steps is not None:
        c.max_steps = max_steps
    for k, v in overrides.items():
        setattr(c, k, v)
    return c


def agent(name, **kwargs):
    return getattr(A, ZOO[name]["agent"])(config(name, **kwargs))


def run(name, **kwargs):
    ag = agent(name, **kwargs)
    run_steps(ag)
    return ag


def _entry(name):
    def fn(**kwargs):
        return run(name, **kwargs)
    fn.__name__ = name
    fn.__doc__ = "run_steps(%s(config)) with the hyper-parameters of the reference's examples.py::%s" % (ZOO[name]["agent"], name)
    return fn


for _n in ZOO:
    globals()[_n] = _entry(_n)
del _n
