"""Oracle: the deterministic schedule of the pipelined async DQN agent step.  TEST INFRASTRUCTURE ONLY.

The reference's async_actor=True (deep_rl/agent/BaseAgent.py:108-182, DQN_agent.py:24-45,101-138) runs the actor in
its own process, two agent steps ahead at most, on whatever parameters the shared-memory network holds at the
time -- a race the reference does not resolve.  deeprl_amd's two-stream pipeline (csrc/learner.hip step_pipelined,
DQNLearnerBench with the actor parameter ring) fixes ONE legal interleaving of that race, and this module restates
it on the CPU with the oracle's own update / forward arithmetic:

    prefill      ring <- synthetic transitions 0 .. cap-1 (counter hash), pos = 0
    actor(0)     4 transitions on theta_0
    for k = 0, 1, ...:
        idx_k  = UniformReplay.sample's rejection loop on the GLOBAL np.random stream, ring state after actor(k)
        batch_k = gather(idx_k)                       # BEFORE actor(k+1) overwrites the oldest slots
        actor(k+1) on theta_k                          # = parameters after updates 0 .. k-1
        theta_{k+1} = dqn_update(theta_k, batch_k)     # DQN_agent.py:114-134, centered RMSprop, clip 5
    actor randomness (torch_utils.py:51-58 order: randint(A) then rand()) comes from its OWN RandomState(seed + 977),
    like the reference's actor process has its own np.random state.

A transition of the synthetic environment: observation = counter-hash frame, reward / mask = hashes of the same
counter (synth_oracle.py); the action does not influence the next observation.
"""
import numpy as np
import torch

from . import loss_oracle as L, net_oracle as N, numerics_oracle as NUM
from .replay_oracle import UniformReplayOracle
from .synth_oracle import synth_transitions


def draw_uniform_indices(size, pos, batch, history, n_step):
    """replay.py:92-110 on the global np.random stream (block draws = the same stream as scalar draws)."""
    out = np.empty(batch, dtype=np.int64)
    have = 0
    while have < batch:
        cand = np.random.randint(0, size, size=batch - have)
        lo, hi = cand - history + 1, cand + n_step
        ok = ((lo >= 0) & (hi < pos)) | ((lo >= pos) & (hi < size))
        good = cand[ok]
        out[have:have + len(good)] = good
        have += len(good)
    return out


class AsyncDqnScheduleOracle:
    """head = "vanilla" (DQN_agent.py:81-99, centered RMSprop), "c51" (CategoricalDQN_agent.py:60-89, Adam) or "qr"
    (QuantileRegressionDQN_agent.py:55-77, Adam): the same schedule with the head's own loss, action values and optimizer
    (round 3: config 4's heads are pinned to the oracle in ASYNC mode too, not only through async == in-order)."""

    def __init__(self, params, target_params, cap, batch, seed, n_actions=4, epsilon=0.01, done_period=800, gamma=0.99,
                 clip=5.0, lr=0.00025, alpha=0.95, eps=0.01, double_q=False, head="vanilla", n_atoms=51, v_min=-10.0,
                 v_max=10.0, betas=(0.9, 0.999)):
        torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
        self.head, self.n_atoms, self.v_min, self.v_max, self.betas = head, int(n_atoms), float(v_min), float(v_max), betas
        self.opt_step = 0
        if head == "c51":     # np.linspace in fp64, then fp32 (CategoricalDQN_agent.py:33, tensor())
            self.atoms = torch.tensor(np.linspace(v_min, v_max, n_atoms), dtype=torch.float32)
        self.p = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
        self.pt = {k: torch.tensor(v) for k, v in target_params.items()}
        self.names = list(self.p)
        self.sq = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.ga = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.cap, self.batch, self.seed, self.A = cap, batch, seed, n_actions
        self.epsilon, self.done_period, self.gamma = epsilon, done_period, gamma
        self.clip, self.lr, self.alpha, self.eps, self.double_q = clip, lr, alpha, eps, double_q
        frames, act, rew, msk = synth_transitions(0, cap, 7056, seed=seed, n_actions=n_actions, done_period=done_period)
        self.rep = UniformReplayOracle(cap, batch, 1, gamma, 4)
        for t in range(cap):
            self.rep.feed_one(frames[t].reshape(84, 84), act[t], rew[t], msk[t])
        self.counter = cap
        self.actor_rs = np.random.RandomState(seed + 977)
        self.actions = []          # every action the actor stored, in order
        self.q_gaps = []           # top-2 gap of the actor's q per greedy decision (tie diagnostics)
        self.losses = []

    def load_state(self, state):
        """Adopts {'params', 'square_avg', 'grad_avg'} (per-tensor CPU tensors, module layout): the checker restarts
        from the implementation's own state after a step whose ReLU gates were ambiguous (net_oracle.nature_conv_body_margin)."""
        with torch.no_grad():
            for k in self.names:
                self.p[k].copy_(state["params"][k])
                self.sq[k] = state["square_avg"][k].clone()      # (Adam heads: exp_avg / exp_avg_sq under the same keys)
                self.ga[k] = state["grad_avg"][k].clone()

    def _snapshot(self):
        return {k: v.detach().clone() for k, v in self.p.items()}

    def actor_step(self, theta, override_actions=None):
        """4 env transitions on parameters `theta` (DQN_agent.py:24-45).  override_actions: actions to STORE instead of
        the oracle's own (keeps the two ring histories aligned after a numerical near-tie)."""
        rep = self.rep
        out = []
        for e in range(4):
            ra = int(self.actor_rs.randint(self.A, size=1)[0])
            dice = float(self.actor_rs.rand(1)[0])
            frame, _, rew, msk = synth_transitions(self.counter, 1, 7056, seed=self.seed, n_actions=self.A,
                                                   done_period=self.done_period)
            slot = rep.pos
            # the observation the actor acts on: the 3 newest ring frames + the new frame
            stack = np.stack([rep.state[(slot - 3 + j) % self.cap] for j in range(3)] + [frame[0].reshape(84, 84)])
            x = torch.from_numpy(NUM.image_normalize_sync(stack[None]))
            with torch.no_grad():
                q = self._action_values(theta, N.nature_conv_body(theta, x)).numpy()[0]
            greedy = int(np.argmax(q))
            srt = np.sort(q)
            action = ra if dice < self.epsilon else greedy
            if not (dice < self.epsilon):
                self.q_gaps.append(float(srt[-1] - srt[-2]))
            stored = action if override_actions is None else int(override_actions[e])
            rep.feed_one(frame[0].reshape(84, 84), np.int64(stored), rew[0], msk[0])
            self.counter += 1
            out.append((action, float(srt[-1] - srt[-2]), dice < self.epsilon))
            self.actions.append(action)
        return out

    def _action_values(self, theta, phi):
        """What the actor takes the argmax of: q (DQN_agent.py:29-33), sum_n prob * atoms (CategoricalDQN_agent.py:21-24),
        mean_n quantile (QuantileRegressionDQN_agent.py:17-20)."""
        if self.head == "c51":
            prob, _ = N.categorical_head(theta, phi, self.A, self.n_atoms)
            return (prob * self.atoms).sum(-1)
        if self.head == "qr":
            return N.quantile_head(theta, phi, self.A, self.n_atoms).mean(-1)
        return N.vanilla_head(theta, phi)

    def sample(self):
        idx = draw_uniform_indices(self.rep.size(), self.rep.pos, self.batch, 4, 1)
        return idx, self.rep.gather(idx)

    def _update_dist(self, batch, weights=None):
        """One update of a distributional head: per-sample loss vector, its (importance-weighted) mean, clipped gradients, Adam."""
        st, ac, rw, ns, mk = batch
        p, pt = self.p, self.pt
        x = torch.from_numpy(NUM.image_normalize_sync(st))
        xn = torch.from_numpy(NUM.image_normalize_sync(ns))
        a_t, r_t, m_t = torch.from_numpy(ac), torch.from_numpy(rw.astype(np.float32)), torch.from_numpy(mk.astype(np.float32))
        phi, self.relu_margin = N.nature_conv_body_margin(p, x)
        if self.head == "c51":
            with torch.no_grad():
                prob_t, _ = N.categorical_head(pt, N.nature_conv_body(pt, xn), self.A, self.n_atoms)
                prob_o = N.categorical_head(p, N.nature_conv_body(p, xn), self.A, self.n_atoms)[0] if self.double_q else None
            prob, log_prob = N.categorical_head(p, phi, self.A, self.n_atoms)
            vec = L.c51_kl(log_prob, prob_t, a_t, r_t, m_t, self.gamma, self.atoms, self.v_min, self.v_max, prob_next_online=prob_o)
            out = (prob * self.atoms).sum(-1)
        else:
            with torch.no_grad():
                qn = N.quantile_head(pt, N.nature_conv_body(pt, xn), self.A, self.n_atoms)
            quant = N.quantile_head(p, phi, self.A, self.n_atoms)
            vec = L.qr_loss(quant, qn, a_t, r_t, m_t, self.gamma)
            out = quant.mean(-1)
        loss = vec.mean() if weights is None else vec.mul(weights).mean()      # DQN_agent.py:126-128
        grads = torch.autograd.grad(loss, [p[k] for k in self.names])
        norm, grads = N.clip_grad_norm(list(grads), self.clip)
        self.opt_step += 1
        with torch.no_grad():
            for k, g in zip(self.names, grads):      # sq / ga hold Adam's exp_avg / exp_avg_sq here
                newp, self.sq[k], self.ga[k] = N.adam_step(p[k], g, self.sq[k], self.ga[k], self.opt_step, self.lr, self.betas[0],
                                                           self.betas[1], self.eps)
                p[k].copy_(newp)
        loss = float(loss.detach())
        self.losses.append(loss)
        return loss, vec.detach().numpy(), out.detach().numpy(), float(norm)

    def update(self, batch, weights=None):
        """weights: importance weights of a prioritized minibatch (DQN_agent.py:124-127), applied to the per-sample loss
        vector before the mean; the returned TD errors / loss vector are the PRE-weight ones (what the priorities use)."""
        if self.head != "vanilla":
            return self._update_dist(batch, weights)
        st, ac, rw, ns, mk = batch
        p, pt = self.p, self.pt
        x = torch.from_numpy(NUM.image_normalize_sync(st))
        xn = torch.from_numpy(NUM.image_normalize_sync(ns))
        with torch.no_grad():
            qn = N.vanilla_head(pt, N.nature_conv_body(pt, xn))
            qno = N.vanilla_head(p, N.nature_conv_body(p, xn)) if self.double_q else None
        phi, self.relu_margin = N.nature_conv_body_margin(p, x)    # smallest |ReLU input| of the differentiated forward
        q = N.vanilla_head(p, phi)
        delta = L.dqn_td_error(q, qn, torch.from_numpy(ac), torch.from_numpy(rw.astype(np.float32)),
                               torch.from_numpy(mk.astype(np.float32)), self.gamma, q_next_online=qno)
        # PER: the importance weights multiply the TD-error VECTOR compute_loss returns (DQN_agent.py:98-99,126), and
        # reduce_loss squares afterwards (:78-79): mean(0.5 * (delta * w)^2)
        loss = L.dqn_reduce(delta) if weights is None else L.dqn_reduce(delta.mul(weights))
        grads = torch.autograd.grad(loss, [p[k] for k in self.names])
        norm, grads = N.clip_grad_norm(list(grads), self.clip)
        with torch.no_grad():
            for k, g in zip(self.names, grads):
                newp, self.sq[k], self.ga[k] = N.rmsprop_step(p[k], g, self.sq[k], self.ga[k], self.lr, self.alpha,
                                                              self.eps, True)
                p[k].copy_(newp)
        loss = float(loss.detach())
        self.losses.append(loss)
        return loss, delta.detach().numpy(), q.detach().numpy(), float(norm)


class _SyntheticAtariOracle:
    """The synthetic Atari emulator behind DummyVecEnv's auto-reset (the stand-in for envs.py:126-150 + a real game): frame
    k = counter hash (synth_transitions), reset() = ONE new frame repeated `history` times, step() hashes (reward, done) from
    the counter of the frame it then generates; after a terminal step the post-step frame is dropped and the stack restarts."""

    def __init__(self, seed, done_period, history=4, n_actions=4):
        self.seed, self.done_period, self.history, self.A = seed, done_period, history, n_actions
        self.counter = 0
        self.frames = None

    def _frame(self):
        f = synth_transitions(self.counter, 1, 7056, seed=self.seed, n_actions=self.A, done_period=self.done_period)[0][0]
        self.counter += 1
        return f.reshape(84, 84)

    def reset(self):
        self.frames = [self._frame()] * self.history
        return np.stack(self.frames)

    def step(self):
        _, _, rew, msk = synth_transitions(self.counter, 1, 7056, seed=self.seed, n_actions=self.A, done_period=self.done_period)
        self.frames = self.frames[1:] + [self._frame()]
        done = msk[0] == 0
        if done:
            self.reset()
        return np.stack(self.frames), float(rew[0]), bool(done)


class AsyncPerAgentScheduleOracle(AsyncDqnScheduleOracle):
    """The async pipeline WITH PrioritizedReplay, at agent level (DQN_agent.py:101-138 with replay.py:152-196 and
    sum_tree.py; round 4 -- until then async + PER was pinned to the in-order path with epsilon = 1 only).  The schedule
    csrc/learner.hip + replay.DeviceDraw fix, per agent step k (the actor runs one agent step ahead):

        report       the 4 transitions actor(k) produced: ring slots, then tree.add(max_priority) each   (replay.py:160-162)
        if total_steps > exploration_steps:
            draw_k   stratified random.uniform descent, valid_index filter, random.choice padding     (replay.py:164-186)
            batch_k  gathered BEFORE actor(k+1) overwrites the oldest slots
        actor(k+1)   on theta_k (= after updates .. k-1): stale by one update, q-dependent actions when dice >= epsilon
        if updating: theta_{k+1} = update(theta_k, batch_k) with importance weights (beta schedule, one inc per update);
                     update_priorities(tree_idx_k, (|loss_vec| + eps)^alpha)  -- max_priority first, first writer wins
        target sync  when total_steps / sgd_update_frequency % target_network_update_freq == 0         (DQN_agent.py:136-138)

    i.e. tree order per update: priorities k -> adds of actor(k+1) -> draw k+1, exactly the in-order order; what differs from
    the in-order run is WHICH parameters the actor saw.  The device computes priorities from ITS fp32 loss vector; a checker
    that wants the tree bit for bit passes those back through `override_priorities` (and compares them with the oracle's own
    at the loss tolerance)."""

    def __init__(self, params, cap, batch, env_seed, done_period, actor_rs, epsilon_fn, beta_fn, exploration_steps,
                 target_freq, sgd_update_frequency=4, replay_eps=0.01, replay_alpha=0.5, n_actions=4, **kw):
        from .replay_oracle import PrioritizedReplayOracle
        kw.setdefault("seed", env_seed)
        AsyncDqnScheduleOracle.__init__(self, params, params, 8, batch, n_actions=n_actions, done_period=done_period, **kw)
        self.cap = cap
        self.rep = PrioritizedReplayOracle(cap, batch, 1, self.gamma, 4)
        self.env = _SyntheticAtariOracle(env_seed, done_period, 4, n_actions)
        self.obs = None
        self.actor_rs, self.epsilon_fn, self.beta_fn = actor_rs, epsilon_fn, beta_fn
        self.exploration_steps, self.target_freq, self.freq = exploration_steps, target_freq, sgd_update_frequency
        self.replay_eps, self.replay_alpha = replay_eps, replay_alpha
        self.total_steps = 0
        self.ahead = None              # transitions of the actor launch not yet reported
        self.actions, self.q_gaps, self.losses = [], [], []

    def actor_step(self, theta, override_actions=None):
        """4 transitions on `theta` (DQN_agent.py:24-45): held back until report()."""
        out, held = [], []
        for e in range(self.freq):
            if self.obs is None:
                self.obs = self.env.reset()
            eps = self.epsilon_fn()
            ra = int(self.actor_rs.randint(self.A, size=1)[0])
            dice = float(self.actor_rs.rand(1)[0])
            x = torch.from_numpy(NUM.image_normalize_sync(self.obs[None]))
            with torch.no_grad():
                q = self._action_values(theta, N.nature_conv_body(theta, x)).numpy()[0]
            srt = np.sort(q)
            rnd = dice < eps
            action = ra if rnd else int(np.argmax(q))
            if not rnd:
                self.q_gaps.append(float(srt[-1] - srt[-2]))
            stored = action if override_actions is None else int(override_actions[e])
            nxt, reward, done = self.env.step()
            held.append((self.obs[-1].copy(), np.int64(stored), float(np.sign(reward)), np.int32(0 if done else 1)))
            self.obs = nxt
            out.append((action, float(srt[-1] - srt[-2]), rnd))
            self.actions.append(action)
        self.ahead = held
        return out

    def report(self):
        """The agent step's accounting of the transitions produced one call earlier: replay feed (ring + tree add)."""
        for frame, action, reward, mask in self.ahead:
            self.rep.feed_one(frame, action, reward, mask)
            self.total_steps += 1
        self.ahead = None
        return self.total_steps > self.exploration_steps

    def draw(self):
        tree_idx, prob, data_idx = self.rep.draw()
        return tree_idx, prob, data_idx, self.rep.gather(data_idx)

    def weights(self, prob, beta):
        """DQN_agent.py:124-126 in fp32: (p * B + 1e-6)^-beta / max."""
        sp = torch.from_numpy(np.asarray(prob, dtype=np.float32))
        w = sp.mul(sp.size(0)).add(1e-6).pow(-beta)
        return w / w.max()

    def priorities(self, loss_vec):
        """DQN_agent.py:121: (|loss| + replay_eps)^replay_alpha on the PRE-weight per-sample loss, fp32."""
        return torch.from_numpy(np.asarray(loss_vec, dtype=np.float32)).abs().add(self.replay_eps).pow(self.replay_alpha).numpy()

    def learn(self, tree_idx, prob, batch, override_priorities=None):
        """update + update_priorities; returns (loss, the pre-weight vector compute_loss returns -- TD errors for the vanilla
        head, KL per sample for C51 --, own priorities, weights)."""
        w = self.weights(prob, self.beta_fn())
        loss, vec, out, norm = self.update(batch, weights=w)     # vec: TD errors (DQN_agent.py:98-99) / KL (CategoricalDQN :85-86)
        prio = self.priorities(vec)
        use = prio if override_priorities is None else np.asarray(override_priorities, dtype=np.float32)
        self.rep.update_priorities(zip(tree_idx.tolist(), [float(v) for v in use]))
        return loss, vec, prio, w.numpy()

    def maybe_sync_target(self):
        if self.total_steps / self.freq % self.target_freq == 0:
            self.pt = {k: v.detach().clone() for k, v in self.p.items()}
            return True
        return False
