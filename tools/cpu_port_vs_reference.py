#!/usr/bin/env python
"""The reference's own CPU timing next to the oracle port's, on a box that has BOTH (the authoring container: the GPU box has
no /root/reference, so bench.py's `cpu_baseline.kind` is "port" there).  VERDICT r3 #3 / SURVEY.md 8(d) "Reference CPU timing":

    loop  = replay.sample() -> compute_loss -> reduce_loss -> zero_grad / backward / clip_grad_norm_ / optimizer.step()
            (deep_rl/agent/DQN_agent.py:114-134), DQN VanillaNet(NatureConvBody), batch 32, history 4, sync UniformReplay
            of 20 000 synthetic 84x84 frames (counter hash: oracle/synth_oracle.py), centered RMSprop (examples.py:67-68),
            clip 5, ImageNormalizer; torch.set_num_threads(1) = the reference's set_one_thread() (examples.py:623).
    ref   = the reference's modules, untouched, under tests/ref_shim.py (UniformReplay, DQNAgent.compute_loss /
            reduce_loss on a stand-in with the attributes those methods read, torch.optim.RMSprop)
    port  = bench.py's cpu_baseline() loop (oracle/: numpy ring, torch-CPU fp32 restatement)

Same seeded inputs, same initial weights; interleaved repetitions.  Writes profiles/<ROUND_TAG, default r06>_cpu_port_vs_reference.json:
updates/s of both and the ratio port / reference that bench.py quotes in `cpu_baseline.sample`.

    python tools/cpu_port_vs_reference.py [seconds_per_repetition] [repetitions]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

B, H, A, RING = 32, 4, 4, 20_000


def shapes():
    return [("body.conv1.weight", (32, 4, 8, 8)), ("body.conv1.bias", (32,)), ("body.conv2.weight", (64, 32, 4, 4)),
            ("body.conv2.bias", (64,)), ("body.conv3.weight", (64, 64, 3, 3)), ("body.conv3.bias", (64,)),
            ("body.fc4.weight", (512, 3136)), ("body.fc4.bias", (512,)), ("fc_head.weight", (A, 512)), ("fc_head.bias", (A,))]


def init_params():
    rs = np.random.RandomState(0)
    return {k: (rs.standard_normal(s) / np.sqrt(max(1, int(np.prod(s[1:]))))).astype(np.float32) for k, s in shapes()}


def make_reference():
    import ref_shim
    ref = ref_shim.load()
    from oracle.synth_oracle import synth_transitions
    net = ref.VanillaNet(A, ref.NatureConvBody())
    tgt = ref.VanillaNet(A, ref.NatureConvBody())
    p0 = {k: torch.from_numpy(v) for k, v in init_params().items()}
    net.load_state_dict(p0)
    tgt.load_state_dict(p0)
    rep = ref.UniformReplay(memory_size=RING, batch_size=B, n_step=1, discount=0.99, history_length=H)
    frames, act, rew, msk = synth_transitions(0, RING, 7056, seed=0)
    for t in range(RING):
        rep.feed(dict(state=[frames[t].reshape(84, 84)], action=[act[t]], reward=[rew[t]], mask=[msk[t]]))

    class Obj:
        pass
    agent, cfg = Obj(), ref.Config()
    cfg.discount, cfg.n_step, cfg.double_q = 0.99, 1, False
    cfg.state_normalizer = ref.ImageNormalizer()
    agent.config, agent.network, agent.target_network = cfg, net, tgt
    opt = torch.optim.RMSprop(net.parameters(), lr=0.00025, alpha=0.95, eps=0.01, centered=True)

    def one():
        tr = rep.sample()
        loss = ref.DQNAgent.reduce_loss(agent, ref.DQNAgent.compute_loss(agent, tr))
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 5)
        opt.step()
        return float(loss.detach())
    return one


def make_port():
    from oracle import loss_oracle as L, net_oracle as N, numerics_oracle as NUM
    from oracle.replay_oracle import UniformReplayOracle
    from oracle.synth_oracle import synth_transitions
    p = {k: torch.nn.Parameter(torch.tensor(v)) for k, v in init_params().items()}
    pt = {k: v.detach().clone() for k, v in p.items()}
    rep = UniformReplayOracle(RING, B, 1, 0.99, H)
    frames, act, rew, msk = synth_transitions(0, RING, 7056, seed=0)
    for t in range(RING):
        rep.feed_one(frames[t].reshape(84, 84), act[t], rew[t], msk[t])
    # clip + optimizer: torch's own clip_grad_norm_ / RMSprop, the library calls the reference makes (DQN_agent.py:130-134,
    # examples.py:67-68) -- the oracle's per-tensor restatement of them (net_oracle.rmsprop_step, used by the parity
    # tests) is 30 % slower than the library's fused loops and would understate the CPU path
    opt = torch.optim.RMSprop(list(p.values()), lr=0.00025, alpha=0.95, eps=0.01, centered=True)

    def one():                                   # = bench.py cpu_baseline()'s loop
        st, ac, rw, ns, mk, _ = rep.sample()
        x = torch.from_numpy(NUM.image_normalize_sync(st))
        xn = torch.from_numpy(NUM.image_normalize_sync(ns))
        with torch.no_grad():
            qn = N.vanilla_head(pt, N.nature_conv_body(pt, xn))
        q = N.vanilla_head(p, N.nature_conv_body(p, x))
        delta = L.dqn_td_error(q, qn, torch.from_numpy(ac), torch.from_numpy(rw.astype(np.float32)),
                               torch.from_numpy(mk.astype(np.float32)), 0.99)
        loss = L.dqn_reduce(delta)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(p.values()), 5)
        opt.step()
        return float(loss.detach())
    return one


def timed(fn, seconds):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        n += 1
    return n / (time.perf_counter() - t0)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.set_num_threads(1)
    np.random.seed(0)
    ref_one, port_one = make_reference(), make_port()
    # same minibatches: both draw from the global np.random stream, reseeded identically before each side's first step
    np.random.seed(7)
    ref_losses = [ref_one() for _ in range(3)]
    np.random.seed(7)
    port_losses = [port_one() for _ in range(3)]
    rates = {"reference": [], "port": []}
    for _ in range(reps):                           # interleaved
        rates["reference"].append(timed(ref_one, seconds))
        rates["port"].append(timed(port_one, seconds))
    r, p = float(np.median(rates["reference"])), float(np.median(rates["port"]))
    out = {"what": "DQN update loop (DQN_agent.py:114-134) on one CPU thread: the reference's own modules under tests/ref_shim.py vs "
                   "the oracle port bench.py times as cpu_baseline (kind 'port') on the GPU box",
           "config": "VanillaNet(NatureConvBody), batch 32, history 4, %d-frame sync UniformReplay, centered RMSprop, clip 5" % RING,
           "threads": 1, "cores_on_this_box": os.cpu_count(), "seconds_per_repetition": seconds, "repetitions": reps,
           "reference_updates_per_s": r, "port_updates_per_s": p, "port_over_reference": p / r,
           "all_repetitions": rates,
           "first_losses": {"reference": ref_losses, "port": port_losses,
                            "note": "same seeded minibatches and initial weights: the two loops are the same computation"},
           "torch": torch.__version__}
    path = os.path.join(ROOT, "profiles", "%s_cpu_port_vs_reference.json" % os.environ.get("ROUND_TAG", "r06"))
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
