#!/usr/bin/env python
"""bench.py -- gradient-updates/sec (+ env-steps/sec) of the DQN Breakout hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.
  * workload = BASELINE.json configs[1]: DQN, NatureConvBody, 84x84x4 uint8 frames, batch 32,
    1M-frame HBM replay ring, centered RMSprop, clip 5 (examples.py:55-97), synthetic frames
    (counter hash, data: "synthetic"), random-init weights.
  * a "step" = one full agent step of that config: 4 environment transitions (synthetic frame
    source + batch-1 greedy/epsilon action selection on device) fed to the ring, one uniform
    minibatch draw (host np.random, reference-exact), gather, target + online forward, TD loss,
    backward, global-norm clip, RMSprop.  Inputs are resident in HBM before the timed region.
  * N > 1: one process per GPU, independent replicas (the reference's only multi-GPU mode for
    off-policy agents, docker_batch.sh:2-8) -> "scaling": "weak", no data-path
    collective; a barrier + max-over-ranks brackets the timed region.
  * roofline: the dominant kernel of the update, timed live with HIP events on the launch stream
    ("source": "hip_events"; the rocprofv3 summary of the same command is committed under profiles/).
  * cpu_baseline: the CPU oracle (port of the reference path on torch-CPU fp32) on a bounded
    sample, rank 0 / N=1 only: `value` with ONE thread (the reference's set_one_thread(), examples.py:623),
    `multi_thread` = the best of an 8 / 16 / 32-thread sweep of the same learner, `all_cores_replicas` = 64 single-thread
    learners side by side (how the CPU path scales on a host).
  * parity_check: after the timed region, SIX more agent steps of the SAME pipelined configuration are replayed through
    the CPU oracle, chained (the oracle carries its own parameters / optimizer state from step to step; the minibatch each
    update consumed comes from the learner): loss and every parameter at 1e-5 per step, the worst is reported.  Outside the
    timing.
  * `--gpus N` without a torchrun environment launches the N ranks itself (torch.distributed.run, 127.0.0.1).
  * a run of fewer than 500 timed steps (the driver's default is 20) is a few milliseconds: `value` is still exactly
    those K steps (the contract), and `long_run` carries the same measurement over 2000 steps.  With a warm-up of fewer than
    100 steps, 300 untimed `setup_steps` run in front of it (graph capture on first use, first touch of the ring, clocks).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, H, A, F = 32, 4, 4, 84 * 84
CONV_FLOPS_FWD = 209_715_200 + 169_869_312 + 115_605_504
UPDATE_FLOPS = 2_182_600_000
FP32_MFMA_PEAK = 157.3e12
HBM_PEAK = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--ring", type=int, default=1_000_000, help="replay capacity in frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-actor", action="store_true", help="skip actor inference (then env_steps is null)")
    ap.add_argument("--variant", type=int, default=-1, help="DRA_VAR_* kernel-variant mask (-1: library default)")
    ap.add_argument("--sync-actor", action="store_true",
                    help="in-order actor (async_actor=False); default is the reference's dqn_pixel setting async_actor=True")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the post-run oracle replay of one agent step")
    ap.add_argument("--no-long-run", action="store_true", help="short runs (< 500 steps): skip the extra 2000-step measurement")
    ap.add_argument("--cpu-replica-worker", type=float, default=0.0,
                    help="internal: run the single-thread CPU oracle loop for this many seconds and print its count")
    ap.add_argument("--master-port", type=int, default=29517, help="rendezvous port when bench.py launches the ranks itself")
    ap.add_argument("--comm-check", action="store_true",
                    help="multi-GPU self-diagnosis instead of a benchmark: device count vs --gpus, what RCCL reports for the "
                         "communicator (ranks, algorithm / protocol lines of NCCL_DEBUG=INFO), and the 6.75 MB gradient all-reduce "
                         "timed alone (us, GB/s)")
    ap.add_argument("--workload", default="dqn_pixel", choices=["dqn_pixel", "a2c_pixel", "ppo_pixel", "ppo_continuous"],
                    help="dqn_pixel = BASELINE configs[1] (the headline); a2c_pixel / ppo_pixel = configs[4]: the on-policy agents, "
                         "environments sharded over the ranks, one gradient all-reduce per optimizer step (SURVEY.md 8e); "
                         "ppo_continuous = configs[2]: PPO on HalfCheetah shapes, 16 workers per GPU (replicas over the ranks)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: launch N ranks of this script (one process per GPU, rendezvous on
    127.0.0.1) and pass rank 0's JSON line through."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def pin_this_rank():
    """One rank per GPU: pin this process to cores of its GPU's NUMA node, disjoint from the node's other ranks
    (deeprl_amd.dist.pin_rank; SURVEY.md 8e: eight Python drivers must not share cores).  Returns the mapping for the line."""
    from deeprl_amd.dist import pin_rank
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", 1)))
    return pin_rank(local_rank, local_world)


def gather_affinity(mine, world):
    """Every rank's placement on rank 0 (all ranks must call)."""
    if world <= 1:
        return [mine]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out


def cpu_baseline(seconds=15.0, ring=20_000, worker=False):
    """The CPU oracle of the same update (port of the reference path): numpy ring gather ->
    f64*(1/255)->f32 -> target fwd, online fwd, TD loss, backward, clip, centered RMSprop on
    torch-CPU fp32, single thread like the reference's set_one_thread() (examples.py:623)."""
    from oracle import loss_oracle as L, net_oracle as N, numerics_oracle as NUM
    from oracle.replay_oracle import UniformReplayOracle
    from oracle.synth_oracle import synth_transitions
    threads_before = torch.get_num_threads()
    torch.set_num_threads(1)
    rs = np.random.RandomState(0)
    shapes = [("body.conv1.weight", (32, 4, 8, 8)), ("body.conv1.bias", (32,)), ("body.conv2.weight", (64, 32, 4, 4)),
              ("body.conv2.bias", (64,)), ("body.conv3.weight", (64, 64, 3, 3)), ("body.conv3.bias", (64,)),
              ("body.fc4.weight", (512, 3136)), ("body.fc4.bias", (512,)), ("fc_head.weight", (A, 512)),
              ("fc_head.bias", (A,))]
    p = {k: torch.nn.Parameter(torch.tensor((rs.standard_normal(s) / np.sqrt(max(1, int(np.prod(s[1:]))))).astype(np.float32)))
         for k, s in shapes}
    pt = {k: v.detach().clone() for k, v in p.items()}
    rep = UniformReplayOracle(ring, B, 1, 0.99, H)
    frames, act, rew, msk = synth_transitions(0, ring, F, seed=0)
    for t in range(ring):
        rep.feed_one(frames[t].reshape(84, 84), act[t], rew[t], msk[t])
    # clip + optimizer: torch's own clip_grad_norm_ / RMSprop, the library calls the reference makes (DQN_agent.py:130-134,
    # examples.py:67-68) -- the oracle's per-tensor restatement of them (net_oracle.rmsprop_step, used by the parity
    # tests) is 30 % slower than the library's fused loops and would understate the CPU path
    opt = torch.optim.RMSprop(list(p.values()), lr=0.00025, alpha=0.95, eps=0.01, centered=True)
    np.random.seed(0)

    def one():
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

    def timed(limit):
        for _ in range(3):
            one()
        n, t0 = 0, time.time()
        while time.time() - t0 < limit:
            one()
            n += 1
        return n, time.time() - t0

    n, dt = timed(seconds)
    if worker:
        torch.set_num_threads(threads_before)
        return {"updates": n, "seconds": dt}
    nproc = os.cpu_count() or 1
    # one learner on several threads: the best of a small sweep.  (torch.set_num_threads(nproc) on this host's 256 hardware
    # threads is an oversubscription artefact, not a baseline: 0.08 updates/s in round 2 -- a batch-32 NatureConv update has no
    # work for that many threads.)
    sweep = {}
    for nt in (8, 16, 32):
        if nt > nproc:
            continue
        torch.set_num_threads(nt)
        one()
        k, t1 = 0, time.time()
        while k < 1 or time.time() - t1 < 2.0:
            one()
            k += 1
        sweep[nt] = k / (time.time() - t1)
    best_nt = max(sweep, key=sweep.get) if sweep else 1
    n_all, dt_all = (sweep[best_nt], 1.0) if sweep else (n / dt, 1.0)
    torch.set_num_threads(threads_before)
    # the port against the reference ITSELF, measured where both exist (the authoring container; tools/cpu_port_vs_reference.py):
    # read from the committed file, so that the "port" number can be converted
    pvr = None
    try:
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_port_vs_reference.json")))[-1]
        rec = json.load(open(newest))
        pvr = {"port_over_reference": rec["port_over_reference"], "reference_updates_per_s": rec["reference_updates_per_s"],
               "port_updates_per_s": rec["port_updates_per_s"], "cores_on_that_box": rec["cores_on_this_box"],
               "source": "committed file profiles/%s (the reference's own modules under tests/ref_shim.py "
                         "vs this loop, same inputs, 1 thread, authoring container)" % os.path.basename(newest)}
    except Exception:
        pass
    return {"value": n / dt, "unit": "gradient-updates/sec", "cores": 1, "kind": "port",
            "sample": "%d DQN updates (B=32, 84x84x4, %d-frame ring) in %.1f s, torch-CPU fp32 oracle + torch's own clip_grad_norm_ / "
                      "RMSprop, 1 thread (the reference's set_one_thread()); kind 'port': the reference tree is not on the GPU box%s"
                      % (n, ring, dt, "; where both exist the port runs at %.2fx the reference's own loop" % pvr["port_over_reference"]
                         if pvr else ""),
            "port_vs_reference": pvr,
            "multi_thread": {"value": n_all / dt_all, "cores": best_nt,
                             "sample": "ONE learner, best of torch.set_num_threads(8 / 16 / 32), 2 s each: %s updates/s"
                                       % {k: round(v, 1) for k, v in sweep.items()}},
            "all_cores_replicas": cpu_replicas(nproc)}


def cpu_replicas(nproc, seconds=6.0, ring=5_000, limit=64):
    """The CPU path as it scales on a host: R independent single-thread learners (own ring, own parameters: what the DQN
    family is on several GPUs too), one process each, all running for the same few seconds; the
    aggregate updates/s.  R = min(cores, 64).  None when the workers cannot be run (never takes the bench line down)."""
    import subprocess
    r = max(1, min(int(nproc), limit))
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""              # the workers are CPU-only: they must not open the GPU
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["OMP_NUM_THREADS"] = env["MKL_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-replica-worker", str(seconds), "--ring", str(ring)]
    try:
        procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(r)]
        total, done = 0.0, 0
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=120 + 4 * seconds)
                rec = json.loads(out.strip().splitlines()[-1])
                total += rec["updates"] / rec["seconds"]
                done += 1
            except Exception:
                pr.kill()
        if done == 0:
            return None
        return {"value": total, "cores": done, "sample": "%d single-thread learners side by side, %.0f s each, %d-frame rings"
                                                          % (done, seconds, ring)}
    except Exception:
        return None


def agent_api(seconds=2.0):
    """The same configuration through the drop-in surface: zoo.agent('dqn_pixel') = the reference's examples.py::dqn_pixel
    (1M-frame replay, async_actor=True; examples.py:55-97) stepped exactly as run_steps does (agent.step() in a loop).
    Five variants: the reference's own setting (async_actor=True: device-resident environment + two-stream pipeline),
    async_actor=False (same kernels in order), device_env=False (HOST emulator: every observation uploaded, every action
    crossing back, as in the reference) in order, the host emulator with async_actor=True (the forward passes of agent
    step t+1 on the actor stream while update t trains), and dqn_pixel(replay_cls=PrioritizedReplay) with the async actor
    (PrioritizedReplay.sample() on the device).  Reported next to `value`, never as `value`."""
    import deeprl_amd as d
    import deeprl_amd.agents as agents_mod
    from deeprl_amd import zoo

    class _Quiet:
        def info(self, *a, **k):
            pass
        add_scalar = add_histogram = info

    agents_mod.get_logger = lambda *a, **k: _Quiet()
    out = {}
    for name, over, kw in (("async_actor", dict(async_actor=True), {}), ("sync_actor", dict(async_actor=False), {}),
                           ("host_emulator", dict(async_actor=False, device_env=False), {}),
                           ("host_emulator_async_actor", dict(async_actor=True, device_env=False), {}),
                           ("async_actor_prioritized_replay", dict(async_actor=True), dict(replay_cls=d.PrioritizedReplay))):
        d.random_seed(1)
        over.update(exploration_steps=200, save_interval=0)
        agent = zoo.agent("dqn_pixel", game="synthetic-atari", overrides=over, **kw)
        for _ in range(300):
            agent.step()
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            agent.step()
            n += 1
        if agent._learner is not None:
            agent._learner.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"updates_per_s": n / dt, "env_steps_per_s": 4 * n / dt, "device_pipeline": agent._pipe is not None,
                     "fused_learner": agent._learner is not None}
        agent.close()
    out["note"] = "DQNAgent.step() of zoo dqn_pixel (= examples.py::dqn_pixel, 1M-frame ring) as run_steps drives it"
    return out


def _continuous_agent(d, zoo, workers, **over):
    """examples.py::ppo_continuous (497-523) on HalfCheetah shapes with `workers` vectorised environments (BASELINE configs[2]
    names 16; the reference's literal task_fn builds one)."""
    over.setdefault("save_interval", 0)
    c = zoo.config("ppo_continuous", game="synthetic-continuous-HalfCheetah", overrides=dict(num_workers=workers, **over))
    c.task_fn = lambda: d.Task(c.game, num_envs=c.num_workers, seed=1)
    return d.PPOAgent(c)


def other_config_lines(seconds=3.0):
    """BASELINE configs[2..4] through the drop-in surface (zoo = the reference's examples.py entries, stepped as run_steps does),
    a few seconds each, so that the driver's own record carries them: DQN + PrioritizedReplay is in agent_api above; here
    C51, C51 + PER, QR-DQN (configs[3]), A2C / PPO on pixels (configs[4], one GPU's share of the environments) and PPO on
    HalfCheetah shapes with 16 workers (configs[2]).  Never `value`."""
    import deeprl_amd as d
    import deeprl_amd.agents as agents_mod
    from deeprl_amd import zoo

    class _Quiet:
        def info(self, *a, **k):
            pass
        add_scalar = add_histogram = info

    agents_mod.get_logger = lambda *a, **k: _Quiet()
    dqn = dict(exploration_steps=200, save_interval=0)
    cases = [
        ("categorical_dqn_pixel", lambda: zoo.agent("categorical_dqn_pixel", game="synthetic-atari", overrides=dict(dqn)), 300, 4, 1),
        ("categorical_dqn_pixel_prioritized_replay",
         lambda: zoo.agent("categorical_dqn_pixel", game="synthetic-atari", replay_cls=d.PrioritizedReplay, overrides=dict(dqn)), 300, 4, 1),
        ("quantile_regression_dqn_pixel",
         lambda: zoo.agent("quantile_regression_dqn_pixel", game="synthetic-atari", overrides=dict(dqn)), 300, 4, 1),
        ("a2c_pixel_16", lambda: zoo.agent("a2c_pixel", game="synthetic-atari", overrides=dict(num_workers=16, save_interval=0)), 6, 80, 1),
        ("ppo_pixel_8", lambda: zoo.agent("ppo_pixel", game="synthetic-atari", overrides=dict(num_workers=8, save_interval=0)), 4, 1024, 16),
        ("ppo_continuous_16", lambda: _continuous_agent(d, zoo, 16), 3, 2048 * 16, 5120),
    ]
    out = {}
    for name, build, warm, env_per_step, upd_per_step in cases:
        try:
            d.random_seed(1)
            agent = build()
            for _ in range(warm):
                agent.step()
            torch.cuda.synchronize()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds or n < 2:
                agent.step()
                n += 1
            learner = getattr(agent, "_learner", None)
            if learner is not None:
                learner.synchronize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = {"env_steps_per_s": n * env_per_step / dt, "updates_per_s": n * upd_per_step / dt, "agent_steps": n,
                         "seconds": dt}
            agent.close()
        except Exception as e:
            out[name] = {"error": repr(e)}
    out["note"] = ("Agent.step() of the zoo entries (= examples.py functions) on synthetic environments; updates_per_s counts "
                   "optimizer steps (PPO: minibatch updates; ppo_continuous: the critic's 5120 per rollout)")
    return out


def parity_check(bench, n_steps=6, gate_margin=5e-7):
    """n_steps more agent steps of the configuration that was just timed (same learner, same pipeline, same graphs), replayed
    through the CPU oracle and CHAINED: the oracle starts from the learner's parameters / RMSprop state before the first of
    them and then carries ITS OWN state from step to step, redoing every update (DQN_agent.py:114-134) on the minibatch
    the learner's gather produced.  Bar per step: loss 1e-5 relative, every parameter within atol 2e-6 + rtol 1e-5 (weights
    are O(0.05)) of the oracle's.  A step in which some ReLU input of the differentiated forward lies within fp32 summation
    noise of zero (|pre-activation| < gate_margin: two correct fp32 implementations may gate it differently, which changes
    that unit's whole backward contribution) is still JUDGED when it is within tolerance (it almost always is); only a step
    that is outside the tolerance AND has such an input is excused -- reported, not judged -- and the oracle re-adopts the
    learner's state after it, which starts a new chain.  Reports the worst errors over the judged steps, the longest chain,
    and every step."""
    from oracle import loss_oracle as L, net_oracle as N, numerics_oracle as NUM
    lr = bench.learner
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    lr.keep_minibatch(True)     # ring-direct update: also gather what it reads, for the oracle (the update keeps reading the ring)
    lr.synchronize()
    snap = lr.export_state()          # parameters / target / RMSprop state before the first step, module layout, CPU
    names = list(snap["params"])
    p = {k: v.clone() for k, v in snap["params"].items()}
    sq = {k: v.clone() for k, v in snap["square_avg"].items()}
    ga = {k: v.clone() for k, v in snap["grad_avg"].items()}
    steps, worst_loss, worst_param, judged, chain, best_chain, all_ok = [], 0.0, 0.0, 0, 0, 0, True
    for it in range(n_steps):
        bench.step()
        lr.synchronize()
        torch.cuda.synchronize()
        now = lr.export_state()
        target = now["target"]
        st, ns, ac, rw, mk = [t.cpu() for t in lr.last_minibatch()]
        gpu_loss = float(lr.delta.double().pow(2).mul(0.5).mean().item())
        pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        x = torch.from_numpy(NUM.image_normalize_sync(st.numpy()))
        xn = torch.from_numpy(NUM.image_normalize_sync(ns.numpy()))
        with torch.no_grad():
            qn = N.vanilla_head(target, N.nature_conv_body(target, xn))
        phi, margin = N.nature_conv_body_margin(pr, x)
        q = N.vanilla_head(pr, phi)
        delta = L.dqn_td_error(q, qn, ac, rw, mk, 0.99)
        loss = L.dqn_reduce(delta)
        grads = torch.autograd.grad(loss, [pr[k] for k in names])
        _, grads = N.clip_grad_norm(list(grads), 5)
        step_abs, ok = 0.0, True
        for k, g in zip(names, grads):
            want, sq[k], ga[k] = N.rmsprop_step(pr[k].detach(), g, sq[k], ga[k], 0.00025, 0.95, 0.01, True)
            err = (now["params"][k] - want).abs()
            step_abs = max(step_abs, float(err.max()))
            ok = ok and bool(torch.all(err <= 2e-6 + 1e-5 * want.abs()))
            p[k] = want
        loss_f = float(loss.detach())
        rel_loss = abs(gpu_loss - loss_f) / max(abs(loss_f), 1e-12)
        ok = ok and rel_loss <= 1e-5
        ambiguous = bool(margin < gate_margin)
        # every step within tolerance is JUDGED (and counts); a step outside it is excused -- reported, not counted -- only when
        # some ReLU input of the differentiated forward lay within fp32 summation noise of zero, i.e. when a gate flip between
        # two correct fp32 implementations explains it; anything else is a failure
        excused = (not ok) and ambiguous
        steps.append({"rel_loss_err": rel_loss, "max_abs_param_err": step_abs, "min_relu_input_abs": margin,
                      "gate_ambiguous": ambiguous, "within_tolerance": bool(ok), "excused": bool(excused),
                      "chained_from_step": it - chain})
        if ok:
            judged += 1
            chain += 1
            best_chain = max(best_chain, chain)
            worst_loss, worst_param = max(worst_loss, rel_loss), max(worst_param, step_abs)
        else:
            if not excused:
                all_ok = False
                judged += 1
                worst_loss, worst_param = max(worst_loss, rel_loss), max(worst_param, step_abs)
            chain = 0       # new chain from the learner's own state
            p = {k: v.clone() for k, v in now["params"].items()}
            sq = {k: v.clone() for k, v in now["square_avg"].items()}
            ga = {k: v.clone() for k, v in now["grad_avg"].items()}
    lr.keep_minibatch(False)
    return {"ok": bool(all_ok and judged > 0), "steps_checked": n_steps, "steps_judged": judged, "longest_chain": best_chain,
            "worst_rel_loss_err": worst_loss, "worst_abs_param_err": worst_param, "steps": steps,
            "tolerance": "per step: loss 1e-5 rel; params atol 2e-6 + rtol 1e-5 against the oracle's CHAINED state; every step "
                         "within tolerance is judged; a step OUTSIDE it is excused (reported, not judged, chain restarted) only if a "
                         "ReLU input lay within %g of zero" % gate_margin,
            "what": "%d more agent steps of the timed configuration (same learner, pipeline and graphs) replayed through the CPU "
                    "oracle on the minibatches its gather produced; outside the timed region" % n_steps}


def replay_gather_line(bench, minibatches=1024, reps=10):
    """The replay ring's minibatch gather (what UniformReplay.sample() / PrioritizedReplay.sample() launch on the generic
    paths; the timed update reads the ring from conv1 and does not gather), live on the bench's own ring: `minibatches` x 32
    samples per launch, block form (5 frames of 7056 B read and 5 written per sample).  north_star's target is quoted on the
    READ roofline; one byte is written per byte read, so read + write is the kernel's position and the read share is half."""
    import deeprl_amd as d
    ring, cap = bench.ring, bench.capacity
    rs = np.random.RandomState(7)
    b = 32 * minibatches
    idx = torch.from_numpy(rs.randint(3, min(cap, bench.size) - 2, size=b).astype(np.int64)).to(d.Config.DEVICE)
    out = ring.gather(idx, (84, 84), torch.uint8, torch.int64, want_f32=True, block=True)
    for _ in range(3):
        ring.gather(idx, (84, 84), torch.uint8, torch.int64, want_f32=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ring.gather(idx, (84, 84), torch.uint8, torch.int64, want_f32=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    rd = wr = b * 5 * 7056
    del out
    return {"kernel": "ring_gather_kernel (block form)", "minibatches_per_launch": minibatches, "samples": b,
            "algorithmic_bytes": {"read": rd, "written": wr}, "us_per_launch": t * 1e6,
            "read_GBps": rd / t / 1e9, "read_plus_write_GBps": (rd + wr) / t / 1e9,
            "frac_read_of_8TBps": rd / t / 8e12, "frac_read_plus_write_of_8TBps": (rd + wr) / t / 8e12,
            "target": "north_star: replay gather >= 70 % of the HBM-read roofline", "target_met": bool(rd / t / 8e12 >= 0.7),
            "note": "every byte read is written once, so the read stream can use at most half of the fabric: the literal read "
                    "fraction is reported, the kernel's roofline position is read + write; the timed update does not launch "
                    "this kernel (conv1 reads the ring in place)"}


def pmc_traffic(kernel_group):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json, made by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of
    tools/pmc_workload.py, calibrated on the gather launch whose byte count is known); null when the
    kernel was not profiled.  Counters cannot be collected inside this process."""
    import glob
    # (newest round's table for the DEFAULT kernels: rNN_pmc_traffic.json; round 3's was rNN_pmc_traffic_layers4.json)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_layers4.json")), key=os.path.basename)
    if not files:
        return None
    try:
        table = json.load(open(files[-1]))["kernels"]
    except Exception:
        return None
    alias = {"conv3_bwd_x": "conv3_bwd", "conv3_bwd_w": "conv3_bwd", "conv2_bwd_x": "conv2_bwd", "conv2_bwd_w": "conv2_bwd",
             "fc4_bwd_x": "fc4_bwd"}
    rec = table.get(alias.get(kernel_group, kernel_group))
    return None if rec is None else rec.get("hbm_bytes")


def rocprof_kernel(kernel_group, flops):
    """The same kernel in the committed `rocprofv3 --kernel-trace --stats -- python bench.py` summary (profiles/
    rNN_rocprofv3_kernel_stats.txt, newest): its average duration without the event pair's launch boundary, and the roofline
    fraction that gives.  A profiler cannot run inside this process; the live number above is the conservative one."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats.txt")))
    # (kernel names as rocprofv3 prints them; the one-pass input-gradient role was ConvDgradOne until round 3)
    # (anchored at multi_kernel<: the chained backward launch's name carries the same role types)
    pats = {"conv2_bwd_x": ("multi_kernel<ConvDgradLin<ConvGeom<32, 20, 64, 4, 2>", "multi_kernel<ConvDgradOne<ConvGeom<32, 20, 64, 4, 2>"),
            "conv3_bwd_x": ("multi_kernel<ConvDgradLin<ConvGeom<64, 9, 64, 3, 1>", "multi_kernel<ConvDgradOne<ConvGeom<64, 9, 64, 3, 1>")}
    pat = {"conv2_bwd_x": None, "conv3_bwd_x": None, "conv2_fwd": "conv_fwd_v2_kernel<V2Geom<32, 20, 64, 4, 2>, false, 1, 4>",
           "conv3_fwd": "conv_fwd_v2_kernel<V2Geom<64, 9, 64, 3, 1>, false, 1, 4>", "fc4_fwd": "LinFwdSlabsOne<3136",
           "fc4_bwd_x": "multi_kernel<LinDgradOne<512>", "conv1_bwd_w": "multi_kernel<ConvWgradOne<ConvGeom<4, 84, 32, 8, 4>",
           "conv1_fwd": "conv_fwd_v2_kernel<V2Geom<4, 84, 32, 8, 4>, true, 1, 4>", "rmsprop_step": "late_step_kernel",
           "grad_norm": "clip_step_kernel<0>", "conv_fwd_chain": "conv_fwd_chain_kernel(", "conv_bwd_chain": "bwd_chain_kernel"}.get(kernel_group)     # (`bwd_chain_kernel<false>(` since DRA_VAR_BWD_CHAIN_FC made it a template)
    if not files or (not pat and kernel_group not in pats):
        return None
    try:
        for line in open(files[-1]):
            if (pat and pat in line) or any(q in line for q in pats.get(kernel_group, ())):
                cols = line.split()                      # ... calls avg_us min_us max_us share
                calls, avg_us = int(cols[-5]), float(cols[-4])
                out = {"avg_ms": avg_us * 1e-3, "calls": calls, "file": os.path.relpath(files[-1], ROOT)}
                if flops:
                    out["achieved"] = flops / (avg_us * 1e-6) / 1e12
                    out["frac"] = out["achieved"] / 157.3
                return out
    except Exception:
        return None
    return None


def on_policy_main(args):
    """BASELINE configs[4]: A2C / PPO on Atari shapes, config.num_workers environments PER GPU (weak scaling: 16 resp. 8,
    examples.py:361-381,525-550), sharded over the ranks with one gradient all-reduce per optimizer step.  A step = one
    agent.step() = one rollout (+ its optimisation phase); value = environment steps per second over all ranks."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    import deeprl_amd as d
    import deeprl_amd.agents as agents_mod
    import deeprl_amd.dist as dd
    from deeprl_amd import zoo
    n_dev = torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        torch.cuda.set_device(local_rank % n_dev)
        dd.init("nccl" if world <= n_dev else "gloo")
    d.select_device(local_rank % n_dev)
    affinity = pin_this_rank()

    class Quiet:
        def info(self, *a, **k):
            pass
        add_scalar = add_histogram = info

    agents_mod.get_logger = lambda *a, **k: Quiet()
    if args.workload == "ppo_continuous":
        return ppo_continuous_main(args, d, zoo, rank, world)
    per_gpu = 16 if args.workload == "a2c_pixel" else 8
    torch.manual_seed(0)
    np.random.seed(0)
    agent = zoo.agent(args.workload, game="synthetic-atari", overrides=dict(num_workers=per_gpu * world, save_interval=0))
    steps_per_call = agent.config.rollout_length * per_gpu * world
    # a step = one agent.step() = one rollout (+ its optimisation phase); at least 4 warm-up steps: the rollout is captured
    # as a graph after two eager calls (agents._OnPolicyGraph), the capture itself is the third
    n_warm = max(4, args.warmup if args.warmup <= 50 else args.warmup // 20)
    for _ in range(n_warm):
        agent.step()
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    k = max(1, args.steps if args.steps <= 400 else args.steps // 20)
    t0 = time.perf_counter()
    for _ in range(k):
        agent.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # what the collective itself saw, and every rank's own rate (its shard's environment steps over ITS wall time): a SCALE
    # record then explains itself -- ranks RCCL spanned vs WORLD_SIZE, stragglers, whether the fc4 segment overlapped
    rccl = agent.dp.comm.info() if getattr(agent.dp, "comm", None) is not None else None
    mine = {"rank": rank, "seconds": dt, "env_steps_per_s": k * agent.config.rollout_length * per_gpu / dt, "cpu_affinity": affinity,
            "rccl_ranks": rccl[0] if rccl else None, "rccl_rank": rccl[1] if rccl else None,
            "early_fc4_exchanges": int(getattr(agent.dp, "early_exchanges", 0)), "device": torch.cuda.current_device()}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "n_ranks": {"world_size": world, "rccl": rccl[0] if rccl else None,
                        "backend": dist.get_backend() if world > 1 else "none", "devices_visible": n_dev},
            "per_rank": per_rank,
            "metric": "env-steps/sec", "value": k * steps_per_call / dt, "unit": "env-steps/s", "n_gpus": world, "steps": k,
            "warmup": n_warm, "ms_per_step": 1e3 * dt / k, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s (BASELINE configs[4]): %d environments per GPU, rollout %d, %s, "
                                   "gradient all-reduce per optimizer step" % (
                                       args.workload, per_gpu, agent.config.rollout_length,
                                       "device-resident synthetic environments" if getattr(agent.task, "on_device", False)
                                       else "host-side synthetic emulators"),
                       "parallelism": "dp%d" % world, "collective": ("dra_allreduce_grads (RCCL): each rank's gradient scaled by ITS weight, then "
                       "summed; PPO (eager backward): the [fc4 + heads] segment goes out first on a communication stream and is joined "
                       "before the clip + optimizer launch; A2C (captured rollout graph): one exchange at the graph's tail -- "
                       "early_fc4_exchanges in per_rank counts the early segments that really went out") if agent.dp.comm else
                       ("torch.distributed " + (dist.get_backend() if world > 1 else "none"))},
            "updates_per_sec": k * (1 if args.workload == "a2c_pixel" else 16) / dt}), flush=True)
    agent.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def ppo_cpu_baseline(seconds=10.0):
    """The CPU oracle of the same update (oracle.ppo_mlp_oracle.ppo_update: the reference's PPO_agent.py:71-99 loop on
    torch-CPU fp32 autograd + torch.optim.Adam, one thread as the reference's set_one_thread()) on a bounded sample:
    minibatch updates of 64 rows over 2048 x 16 rollout rows until `seconds` have passed."""
    from oracle import ppo_mlp_oracle as O
    torch.set_num_threads(1)
    rs = np.random.RandomState(0)
    s_dim, a_dim, n = 17, 6, 2048 * 16
    actor, critic = O.init_params(s_dim, a_dim, 64, seed=1)
    state = torch.from_numpy(rs.randn(n, s_dim).astype(np.float32))
    with torch.no_grad():
        pred = O.gaussian_forward(actor, critic, state, noise=torch.from_numpy(rs.randn(n, a_dim).astype(np.float32)))
    entries = [state, pred['action'], pred['log_pi_a'], pred['v'] + torch.from_numpy(rs.randn(n, 1).astype(np.float32)),
               torch.from_numpy(rs.randn(n, 1).astype(np.float32))]
    done, t0, state_opt = 0, time.perf_counter(), None
    while time.perf_counter() - t0 < seconds:
        rows = rs.permutation(n)[:64 * 64]          # 64 minibatches per call
        sub = [e[rows] for e in entries]
        a_opt, c_opt, _, _ = O.ppo_update(actor, critic, sub, [np.arange(len(rows))], 64, 0.2, 0.0, 0.01, opt_state=state_opt)
        state_opt = (a_opt.state_dict(), c_opt.state_dict())
        done += 64
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": "%d minibatch updates (64 rows, both networks, Adam) of the CPU oracle in %.1f s, one thread" % (done, dt)}


def ppo_continuous_main(args, d, zoo, rank, world):
    """BASELINE configs[2]: PPO, GaussianActorCriticNet over two 17 -> 64 -> 64 tanh MLPs, 16 vectorised HalfCheetah-shaped
    workers per GPU, rollout 2048, 10 epochs x 512 minibatches of 64 (examples.py:497-523).  A step = one agent.step() = one
    rollout + its optimisation phase: three persistent launches (rollout, value, update) + scan / normalise / pack.  N > 1:
    independent replicas (the persistent update kernel keeps its weights on one CU; sharding 16 environments would put a
    collective between 64-row minibatches 10 us apart): value = sum over ranks."""
    import torch.distributed as dist
    torch.manual_seed(rank)
    np.random.seed(rank)
    agent = _continuous_agent(d, zoo, 16, data_parallel=False)
    per_step = agent.config.rollout_length * 16
    n_warm = max(3, args.warmup if args.warmup <= 20 else args.warmup // 20)
    for _ in range(n_warm):
        agent.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    k = max(1, args.steps if args.steps <= 100 else args.steps // 20)
    t0 = time.perf_counter()
    for _ in range(k):
        agent.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        agent._mlp.sync_counts()
        n_mb = agent.config.optimization_epochs * (per_step // agent.config.mini_batch_size)
        # the dominant kernel, timed live with HIP events on its stream: the persistent update launch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            ev0.record()
            agent.step()
            ev1.record()
            torch.cuda.synchronize()
            times.append(ev0.elapsed_time(ev1))
        # algorithmic flops of one minibatch update (both networks, forward + input / weight gradients), 64 rows
        mac_a = 17 * 64 + 64 * 64 + 64 * 6
        mac_c = 17 * 64 + 64 * 64 + 64 * 1
        flops_mb = 2 * 64 * ((mac_a + mac_c) + (2 * (mac_a + mac_c) - 2 * 17 * 64))
        ms_step = float(np.median(times))
        ach = n_mb * flops_mb / (ms_step * 1e-3) / 1e12
        out = {"metric": "env-steps/sec", "value": world * k * per_step / dt, "unit": "env-steps/s", "n_gpus": world, "steps": k,
               "warmup": n_warm, "ms_per_step": 1e3 * dt / k, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "PPO HalfCheetah shapes (17 obs, 6 actions), 16 vectorised workers per GPU, rollout 2048, "
                                      "10 epochs x 512 minibatches of 64, two Adam optimisers (BASELINE configs[2]); device-resident "
                                      "synthetic environments" + ("" if getattr(agent.task, "on_device", False) else " NOT ACTIVE"),
                          "parallelism": "replicas x%d" % world},
               "updates_per_sec": world * k * n_mb / dt,
               "roofline": {"kernel": "whole agent step (rollout + value + scan + pack + ppo_mlp_update_kernel)", "bound": "mfma",
                            "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None,
                            "avg_ms": ms_step, "algorithmic_flops": n_mb * flops_mb,
                            "note": "a chain of 5120 dependent 64-row updates on TWO workgroups (one network each): latency-bound by "
                                    "construction (SURVEY.md 8d) -- the fraction of the chip's MFMA peak says how little of the "
                                    "chip a sequential 5.7 k-parameter update can use, not how well the kernel is scheduled; "
                                    "profiles/r05*_prof_ppo_mlp.json has cycles per phase"},
               "fused_update_launches": agent._mlp.launches}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = ppo_cpu_baseline()
        print(json.dumps(out), flush=True)
    agent.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def comm_check(args):
    """`bench.py --gpus N --comm-check`: one SCALE run's worth of self-diagnosis for the data-parallel path.
    Every rank: is there a GPU for it, what does RCCL say about the communicator, how long does the all-reduce of the actor-critic
    net's flat fp32 gradient (1 686 693 floats = 6.75 MB, examples.py:361-381 / 525-550 with 4 actions) take on its own.
    Rank 0 prints ONE JSON line.  Algorithm / protocol: RCCL only reports its choice through NCCL_DEBUG=INFO, so the ranks run
    with NCCL_DEBUG=INFO, NCCL_DEBUG_SUBSYS=INIT,COLL,TUNING into per-rank files and rank 0 returns the lines that name them."""
    import glob
    import tempfile
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    n_dev = torch.cuda.device_count()
    log_dir = os.environ.get("DRA_COMM_CHECK_DIR") or tempfile.mkdtemp(prefix="dra_rccl_")
    os.environ.setdefault("NCCL_DEBUG", "INFO")
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL,TUNING")
    os.environ.setdefault("NCCL_DEBUG_FILE", os.path.join(log_dir, "rccl_rank%d_%%p.log" % rank))
    out = {"comm_check": True, "n_gpus_requested": args.gpus, "world_size": world, "devices_visible": n_dev,
           "enough_devices": n_dev >= world}
    if n_dev < 1:
        out["error"] = "no GPU visible"
        print(json.dumps(out), flush=True)
        return 3
    import deeprl_amd as d
    import deeprl_amd.dist as dd
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        torch.cuda.set_device(local_rank % n_dev)
        dd.init("nccl" if world <= n_dev else "gloo")
    d.select_device(local_rank % n_dev)
    n = 1_686_693
    grad = torch.randn(n + (-n) % 4, dtype=torch.float32, device=d.Config.DEVICE)
    comm = dd.RcclComm() if (world == 1 or world <= n_dev) else None
    rec = {"rank": rank, "device": torch.cuda.current_device(), "cpu_affinity": pin_this_rank()}
    if comm is not None:
        info = comm.info()
        rec.update(rccl_ranks=info[0], rccl_rank=info[1])
        for _ in range(10):
            comm.allreduce_grads(grad, 1.0 / world)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            comm.allreduce_grads(grad, 1.0 / world)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        nbytes = 4 * n
        rec.update(allreduce_us=us, alg_GBps=nbytes / us / 1e3,
                   bus_GBps=(2.0 * (world - 1) / world) * nbytes / us / 1e3 if world > 1 else 0.0)
        comm.close()
    else:
        rec["note"] = "more ranks than GPUs: the ranks share devices over gloo, RCCL is not exercised"
    recs = [rec]
    if world > 1:
        recs = [None] * world
        dist.all_gather_object(recs, rec)
    if rank == 0:
        lines = []
        for f in sorted(glob.glob(os.path.join(log_dir, "rccl_rank*.log")))[:2]:
            try:
                for ln in open(f, errors="replace"):
                    low = ln.lower()
                    if any(k in low for k in ("algo", "proto", "ring", "tree", "xgmi", "channel", "nranks", "rccl version", "nccl version")):
                        lines.append(ln.strip()[:240])
            except Exception:
                pass
        out["per_rank"] = recs
        slow = max((r.get("allreduce_us") or 0.0) for r in recs)
        out["allreduce_us_max_over_ranks"] = slow
        if world > 1 and slow > 0:
            bus = (2.0 * (world - 1) / world) * 4 * n / slow / 1e3
            links = min(world - 1, 7)
            out["bus_GBps"] = bus
            out["per_link_GBps_if_spread_over_%d_links" % links] = bus / links
            out["xgmi_link_peak_GBps"] = 153.0
            out["reading"] = ("a ring keeps ONE link per direction busy per step: if bus_GBps is near one link's 153 GB/s the ring is "
                              "link-bound and a direct reduce-scatter + all-gather over all %d links (SURVEY.md 5) would pay; if it is "
                              "far below, the 6.75 MB exchange is latency-bound and overlap (dist.DataParallel.plan_split) matters more" % links)
        out["rccl_debug_lines"] = lines[:60]
        out["rccl_debug_dir"] = log_dir
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if args.cpu_replica_worker > 0:
        print(json.dumps(cpu_baseline(args.cpu_replica_worker, min(args.ring, 20_000), worker=True)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.comm_check and torch.cuda.device_count() < args.gpus:
            print(json.dumps({"comm_check": True, "n_gpus_requested": args.gpus, "devices_visible": torch.cuda.device_count(),
                              "enough_devices": False, "error": "fewer GPUs than --gpus: nothing launched"}), flush=True)
            sys.exit(3)
        sys.exit(spawn_ranks(args))
    if args.comm_check:
        sys.exit(comm_check(args))
    if args.workload != "dqn_pixel":
        return on_policy_main(args)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        n_dev = torch.cuda.device_count()
        backend = "nccl"
        if world > n_dev:
            # more ranks than GPUs (only when exercising the launch contract on a 1-GPU box): EVERY rank shares
            # devices and uses gloo for the barrier / max-reduce; the replicas themselves never communicate
            local_rank, backend = local_rank % n_dev, "gloo"
        if os.environ.get("DRA_BENCH_BACKEND"):
            backend = os.environ["DRA_BENCH_BACKEND"]
        import datetime
        torch.cuda.set_device(local_rank)
        limit = datetime.timedelta(seconds=600)     # a rendezvous problem must fail, not hang the node
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
        else:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # the container hostname may not resolve
            dist.init_process_group(backend, timeout=limit)
    import deeprl_amd as d
    from deeprl_amd.learner import DQNLearnerBench
    d.select_device(local_rank)
    affinity = gather_affinity(pin_this_rank(), world)
    torch.manual_seed(1234 + rank)
    np.random.seed(rank)
    variant = args.variant
    shared_device = distributed and world > torch.cuda.device_count()
    if shared_device and variant < 0:
        # several ranks on ONE GPU (the launch contract exercised on a 1-GPU box): DRA_VAR_ACTOR_PERSIST's launch needs its 32
        # workgroups co-resident on the actor partition -- two processes' launches on the same CUs can each hold a part of it and
        # wait for the rest until their bounded waits give up.  A process that shares its GPU runs the multi-launch actor and
        # the event path (one rank per GPU, the deployment the variants are for, is unaffected)
        from deeprl_amd import ops as _ops
        variant = _ops.get_tuning() & ~(_ops.VAR_ACTOR_PERSIST | _ops.VAR_FLAG_SYNC | _ops.VAR_LANE_EAGER)
    bench = DQNLearnerBench(ring_capacity=args.ring, batch=B, seed=rank, actor=not args.no_actor,
                            async_actor=not args.sync_actor, variant=variant)
    # Setup before the contract's W warm-up steps when W is tiny (the driver's W = 5 is 0.6 ms): the four rotation slots'
    # graphs are captured on first use, the 1M-frame ring's pages are touched for the first time and the clocks ramp.  These
    # steps are untimed like the warm-up, reported as `setup_steps`; the K timed steps are exactly K steps either way.
    setup_steps = 300 if args.warmup < 100 else 0
    for _ in range(setup_steps + args.warmup):
        bench.step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    bench.learner.host_stats(reset=True)
    lane0 = bench.learner.lane_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bench.step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    host = bench.learner.host_stats(reset=True)
    host["python_loop_us_per_step"] = 1e6 * t_host / args.steps
    lane1 = bench.learner.lane_stats()
    n_lane = lane1["steps"] - lane0["steps"]
    # DRA_VAR_FLAG_SYNC: how many of the timed steps ran in the event-free lane, and the C call's parts there (host microseconds)
    host["lane"] = {"steps": n_lane, "hazard_bumps": lane1["hazard_bumps"] - lane0["hazard_bumps"],
                    "host_waits": lane1["host_waits"] - lane0["host_waits"],
                    "c_call_parts_us": {k: (lane1["host_us_per_step"][k] * lane1["steps"] - lane0["host_us_per_step"][k] * lane0["steps"])
                                        / max(1, n_lane) for k in lane1["host_us_per_step"]}}
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    long_run = None
    if args.steps < 500 and not args.no_long_run:
        # the contract's K steps are a few milliseconds; the same measurement over 2000 steps next to it (every rank runs
        # it so that the replicas keep loading the node the same way; not part of `value`)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(2000):
            bench.step()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        dt_long = time.perf_counter() - t1
        if distributed:
            t = torch.tensor([dt_long], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_long = float(t.item())
        long_run = {"steps": 2000, "value": world * 2000 / dt_long, "ms_per_step": 1e3 * dt_long / 2000}
    if rank == 0:
        parity = None
        if not args.no_parity_check and not args.no_actor:   # first: the learner is still in exactly the timed state
            try:
                parity = parity_check(bench)
            except Exception as e:      # the checker must never take the measurement down with it
                parity = {"ok": False, "error": repr(e)}
        roof = bench.roofline(200 if args.steps < 500 else 500)
        roof["source"] = "hip_events (a pair around the kernel on its launch stream, live: includes the launch boundary)"
        # NOT measured in this process (a profiler cannot run inside it): read from the newest committed summaries of the same
        # command on a builder box, and labelled so
        roof["rocprofv3"] = rocprof_kernel(roof["kernel"], roof.get("algorithmic_flops"))
        if roof["rocprofv3"] is not None:
            roof["rocprofv3"]["source"] = "committed file %s (rocprofv3 --kernel-trace --stats of this command on a builder box)" % roof["rocprofv3"].get("file")
        roof["traffic"] = pmc_traffic(roof["kernel"])
        roof["traffic_source"] = "newest committed profiles/r*_pmc_traffic*.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on a builder box)"
        # the same live measurement with the empty event pair's cost taken off the (kernel + boundary + record) reading:
        # a LOWER bound of the kernel's duration, hence an upper bound of the fraction -- the truth lies between the two
        if roof.get("algorithmic_flops") and roof.get("avg_ms") and roof.get("event_pair_empty_ms"):
            t_low = max(roof["avg_ms"] - roof["event_pair_empty_ms"], 1e-6)
            roof["frac_event_pair_corrected"] = roof["algorithmic_flops"] / (t_low * 1e-3) / 1e12 / 157.3
        mf = getattr(bench, "roofline_mfma", None)
        if mf is not None:
            # The longest launch of the update is the optimizer's (fold + norm + RMSprop), but its 40 MB working set lives in
            # the 256 MiB Infinity Cache: its GB/s are a fabric / MALL number, not an HBM one.  The path's
            # limiter is the MFMA-bound backward: `roofline` IS that kernel (conv2's dgrad + wgrad launch), the optimizer is
            # kept as `roofline.longest_kernel`.
            mf["rocprofv3"] = rocprof_kernel(mf["kernel"], mf.get("algorithmic_flops"))
            if mf["rocprofv3"] is not None:
                mf["rocprofv3"]["source"] = roof["rocprofv3"]["source"] if roof.get("rocprofv3") else "committed file"
            mf["traffic"] = pmc_traffic(mf["kernel"])
            mf["traffic_source"] = roof["traffic_source"]
            mf["source"] = roof["source"]
            if mf.get("avg_ms") and mf.get("event_pair_empty_ms"):
                t_low = max(mf["avg_ms"] - mf["event_pair_empty_ms"], 1e-6)
                mf["frac_event_pair_corrected"] = mf["algorithmic_flops"] / (t_low * 1e-3) / 1e12 / 157.3
            roof["bound"] = "fabric/MALL"
            roof["peak_note"] = "8000 GB/s is the HBM spec, quoted as a yardstick only: FETCH_SIZE counts Infinity-Cache hits"
            mf["longest_kernel"] = roof
            roof = mf
        # DRA_VAR_FWD_CHAIN / DRA_VAR_BWD_CHAIN (round 6): the timed pipeline no longer launches the per-layer convolution kernels
        # -- conv1 + conv2 + conv3 forward of both nets are ONE launch (conv_fwd_chain_kernel, the longest kernel of the update
        # stream), the three conv backward layers another (bwd_chain_kernel).  The headline kernel is the chained forward launch,
        # replayed alone in this run (dra_dqn_learner_chain_replay); the chained backward and the per-layer reading (what the
        # eager profile launches; the headline of rounds 2-5) stay beside it under their own keys.
        chains = getattr(bench, "roofline_chains", None) or {}
        fc = chains.get("conv_fwd_chain")
        chained = bool(fc and fc.get("frac"))
        if chained:
            for name, ch in chains.items():
                if not ch.get("frac"):
                    continue
                ch["rocprofv3"] = rocprof_kernel(name, ch["algorithmic_flops"])
                if ch["rocprofv3"] is not None:
                    ch["rocprofv3"]["source"] = "committed file %s (rocprofv3 --kernel-trace --stats of this command on a builder box)" % ch["rocprofv3"].get("file")
                ch["traffic"] = pmc_traffic(name)
                alg = ch.get("algorithmic_bytes_hbm")
                if name == "conv_fwd_chain" and (bench.learner.variant & 8388608):
                    # DRA_VAR_DEFER_FC4: in the pipeline (where the counters were collected) the launch also steps fc4's weights of
                    # the previous update -- parameter, gradient and two state buffers read; parameter, two state buffers and the
                    # actor's copy written: 8 x 4 B per weight
                    ch["algorithmic_bytes_riders"] = 8 * 4 * 512 * 3136
                    alg = (alg or 0) + ch["algorithmic_bytes_riders"]
                if ch.get("traffic") and alg:
                    ch["traffic_ratio"] = ch["traffic"] / alg
            fc["traffic_source"] = roof.get("traffic_source")
            fc["source"] = "graph replay of the chained launch alone (this run)"
            fc["in_pipeline_note"] = ("in the timed pipeline the same launch also carries the deferred fc4 optimizer segment "
                                      "(DRA_VAR_DEFER_FC4: ~45 MB of parameter / state traffic as trailing workgroups, +3-4 us); a "
                                      "replay never steps parameters, so rocprofv3's in-pipeline duration of this kernel is that "
                                      "much longer than the live reading")
            fc["backward_chain"] = chains.get("conv_bwd_chain")
            fc["per_layer_launch"] = roof      # conv2's backward launch alone (+ the optimizer launch as its longest_kernel)
            roof = fc
        # `frac` / `achieved` of the headline kernel are THIS RUN's (ADVICE r5 / VERDICT r5 item 3): the kernel replayed alone, 64
        # dependent launches in one captured graph between two events -- microseconds per launch INCLUDING one in-graph launch
        # boundary.  That is the conservative live reading and it lands within a few per cent of the rocprofv3 duration of the same
        # kernel inside the running pipeline (r06: 9.8-9.9 us per launch against 10.3-10.5 us): alone and replayed, the kernel's
        # operands are L2-warm, which about cancels the boundary.  Beside it: the same replay minus an EMPTY kernel's per-launch
        # period (`frac_kernel_alone_warm`: the kernel's own duration with warm caches, an upper bound of the in-pipeline
        # fraction), the event pair around one eager launch (`frac_hip_events`, kernel + boundary + record), and the committed
        # builder-box rocprofv3 figure under its own key only.
        roof["frac_hip_events"], roof["achieved_hip_events"] = (None, None) if chained else (roof.get("frac"), roof.get("achieved"))
        rp = roof.get("rocprofv3")
        if rp and rp.get("frac"):
            roof["frac_rocprofv3_committed"] = rp["frac"]
        gr = roof.get("graph_replay")
        if gr and gr.get("frac_with_boundary"):
            roof["frac"] = gr["frac_with_boundary"]
            roof["achieved"] = gr["frac_with_boundary"] * 157.3
            roof["avg_ms"] = gr["us_per_launch"] * 1e-3
            roof["frac_kernel_alone_warm"] = gr["frac"]
            roof["frac_source"] = ("this run: graph replay of the kernel alone, 64 dependent launches between two events, per-launch "
                                   "period incl. one in-graph boundary (conservative); frac_kernel_alone_warm = the same minus an "
                                   "empty kernel's period; frac_hip_events = event pair around one eager launch; "
                                   "frac_rocprofv3_committed = builder-box profile")
        else:
            roof["frac_source"] = "hip_events of this run (event pair around one eager launch, launch boundary included)"
        if roof.get("traffic") and roof.get("algorithmic_bytes_hbm") and "traffic_ratio" not in roof:
            roof["traffic_ratio"] = roof["traffic"] / roof["algorithmic_bytes_hbm"]
        # the update chain owns a CU partition while the device actor runs beside it (lever 5):
        # the kernel is timed on that stream, i.e. on this many of the device's CUs
        roof["stream_cus"] = bench.learner.update_cus or torch.cuda.get_device_properties(0).multi_processor_count
        extra = bench.report()
        try:
            extra["replay_gather"] = replay_gather_line(bench)
        except Exception as e:      # an extra line must never take the headline down
            extra["replay_gather"] = {"error": repr(e)}
        if not args.no_actor:
            extra["host_us_per_step"] = bench.host_profile(100)
        ups = world * args.steps / dt
        out = {
            "metric": "gradient-updates/sec", "value": ups, "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "setup_steps": setup_steps, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DQN Breakout 84x84x4 uint8, NatureConvBody, batch 32, %d-frame HBM replay ring, "
                                   "centered RMSprop, clip 5 (BASELINE configs[1])" % args.ring,
                       "global_batch": B * world, "parallelism": "replicas x%d" % world,
                       "actor_in_step": not args.no_actor, "async_actor": not args.sync_actor,
                       "kernel_variant": bench.learner.variant},
            "env_steps_per_sec": (4 * ups) if not args.no_actor else None,
            "roofline": roof,
        }
        out.update(extra)
        out["host"] = host      # per step: time in the enqueueing C call, of which blocked on the GPU; whole python loop
        out["per_rank_cpu_affinity"] = affinity
        if long_run is not None:
            out["short_run"] = True
            out["long_run"] = long_run
        if parity is not None:
            out["parity_check"] = parity
        if world == 1 and not args.no_cpu_baseline:
            out["agent_api"] = agent_api()
            try:
                out["agent_api"]["other_configs"] = other_config_lines()
            except Exception as e:      # extra lines must never take the headline down
                out["agent_api"]["other_configs"] = {"error": repr(e)}
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()          # rank 0 is still measuring the roofline / baselines: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
