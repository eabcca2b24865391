#!/usr/bin/env python
"""Where a minibatch of the persistent PPO-MLP update kernel goes: shader-clock cycles per phase (dra_ppo_mlp_update_profile,
thread 0 of the actor's / critic's workgroup) at BASELINE configs[2] shapes (17 -> 64 -> 64 -> {6, 1}, minibatch 64), next to the
launch's HIP-event time.  Measurement aid, not product code.

    python tools/prof_ppo_mlp.py [n_rows] [epochs]
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import deeprl_amd as d  # noqa: E402
from deeprl_amd import ppo_mlp  # noqa: E402
from deeprl_amd._lib import lib, ptr, stream_ptr  # noqa: E402

PHASES = ["prefetch issue", "F1 (x W1, tanh)", "barrier", "F2 (h1 W2, tanh)", "barrier", "F3 + loss", "barrier + gate",
          "B3 (dz2, dW3)", "barrier", "B2 (dW2, dz1)", "barrier", "B1 (dW1)", "Adam", "barrier", "publish + commit", "barrier"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    d.select_device(0)
    dev = d.Config.DEVICE
    import test_gpu_ppo_mlp as T
    from oracle import ppo_mlp_oracle as O      # (inputs only: random parameters / rollout rows; nothing is compared here)
    rs = np.random.RandomState(0)
    s_dim, a_dim, hidden, mb = 17, 6, 64, 64
    actor, critic = O.init_params(s_dim, a_dim, hidden, seed=1)
    entries = T._entries(rs, n, s_dim, a_dim, actor, critic)
    fa, pa = T._flat_pair(d, actor, 3e-4)
    fc, pc = T._flat_pair(d, critic, 1e-3)
    steps = torch.zeros(2, dtype=torch.int64, device=dev)
    cfg = ppo_mlp.Cfg()
    cfg.state_dim, cfg.action_dim, cfg.hidden, cfg.mini_batch = s_dim, a_dim, hidden, mb
    cfg.ratio_clip, cfg.entropy_weight, cfg.kl_limit = 0.2, 0.0, 1e9
    na, nc = T._net_struct(fa, pa, steps[0:1], True), T._net_struct(fc, pc, steps[1:2], False)
    e = [x.to(dev).contiguous() for x in entries]
    perm = torch.from_numpy(np.concatenate([rs.permutation(n) for _ in range(epochs)]).astype(np.int64)).to(dev)
    floats = ctypes.c_int64()
    lib.dra_ppo_mlp_packed_floats(n, epochs, mb, s_dim, ctypes.byref(floats))
    packed = torch.empty(floats.value, dtype=torch.float32, device=dev)
    lib.dra_ppo_mlp_pack(ptr(e[0]), ptr(e[1]), ptr(e[2]), ptr(e[4]), ptr(e[3]), ptr(perm), n, epochs, mb, s_dim, a_dim, ptr(packed),
                         stream_ptr())
    out3 = torch.zeros(3, dtype=torch.float32, device=dev)
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    cycles = torch.zeros(32, dtype=torch.int64, device=dev)
    total = epochs * ((n + mb - 1) // mb)
    res = {}
    for name, call in (("product", lambda: lib.dra_ppo_mlp_update(ctypes.byref(cfg), ctypes.byref(na), ctypes.byref(nc), ptr(packed), n,
                                                                  epochs, ptr(out3), ptr(counts), None, stream_ptr())),
                       ("profile", lambda: lib.dra_ppo_mlp_update_profile(ctypes.byref(cfg), ctypes.byref(na), ctypes.byref(nc),
                                                                          ptr(packed), n, epochs, ptr(out3), ptr(counts), ptr(cycles),
                                                                          stream_ptr()))):
        call()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        call()
        ev1.record()
        torch.cuda.synchronize()
        res[name + "_ms"] = ev0.elapsed_time(ev1)
        res[name + "_us_per_minibatch"] = res[name + "_ms"] * 1e3 / total
    c = cycles.cpu().numpy().reshape(2, 16).astype(np.float64) / total
    res["minibatches"] = total
    for role, row in zip(("actor", "critic"), c):
        res[role + "_cycles_per_minibatch"] = {"%02d %s" % (i, p): round(float(v), 1) for i, (p, v) in enumerate(zip(PHASES, row))}
        res[role + "_cycles_total"] = round(float(row.sum()), 1)
    res["clock_MHz_implied"] = round(c[0].sum() / res["profile_us_per_minibatch"], 1)
    res["rollout"] = rollout_profile(d, T, O, fa, pa, fc, pc, cfg, steps)
    print(json.dumps(res, indent=1))


ROLLOUT_PHASES = ["F1 (+ step-top loads / stores)", "barrier + F2", "barrier + heads", "barrier + environment",
                  "barrier + statistics", "barrier + normalisation", "barrier"]


def rollout_profile(d, T, O, fa, pa, fc, pc, cfg, steps, n=16, t_len=2048):
    """cycles per phase of the rollout kernel's step loop at BASELINE configs[2] shapes (16 environments, 2048 steps)"""
    dev = d.Config.DEVICE
    s_dim, a_dim = cfg.state_dim, cfg.action_dim
    na, nc = T._net_struct(fa, pa, steps[0:1], True), T._net_struct(fc, pc, steps[1:2], False)
    f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    o = dict(state=f(t_len, n, s_dim), action=f(t_len, n, a_dim), log_pi_a=f(t_len, n), v=f(t_len + 1, n), reward=f(t_len, n),
             mask=f(t_len, n))
    env_state = torch.zeros(n, s_dim, dtype=torch.float64, device=dev)
    env_counter = torch.zeros(n, dtype=torch.int64, device=dev)
    env_seed = torch.arange(n, dtype=torch.int64, device=dev)
    rms = torch.cat([torch.zeros(s_dim), torch.ones(s_dim), torch.tensor([1e-4])]).double().to(dev)
    cur_state = torch.zeros(n, s_dim, dtype=torch.float32, device=dev)
    sampler = torch.zeros(1, dtype=torch.int64, device=dev)
    cycles = torch.zeros(8, dtype=torch.int64, device=dev)
    io = ppo_mlp.RolloutIO()
    io.env_state, io.env_counter, io.env_seed, io.rms = env_state.data_ptr(), env_counter.data_ptr(), env_seed.data_ptr(), rms.data_ptr()
    io.cur_state, io.sampler_step = cur_state.data_ptr(), sampler.data_ptr()
    io.out_state, io.out_action, io.out_log_pi_a = o['state'].data_ptr(), o['action'].data_ptr(), o['log_pi_a'].data_ptr()
    io.out_v, io.out_reward, io.out_mask = o['v'].data_ptr(), o['reward'].data_ptr(), o['mask'].data_ptr()
    io.env0, io.n_global, io.noise_seed, io.horizon = 0, n, 1, 1000
    io.reward_coef, io.rms_epsilon, io.rms_clip, io.rms_update, io.t_len, io.n_env = 1.0, 1e-8, 10.0, 1, t_len, n
    out = {}
    for name, call in (("product", lambda: lib.dra_ppo_mlp_rollout(ctypes.byref(cfg), ctypes.byref(na), ctypes.byref(nc),
                                                                   ctypes.byref(io), stream_ptr())),
                       ("profile", lambda: lib.dra_ppo_mlp_rollout_profile(ctypes.byref(cfg), ctypes.byref(na), ctypes.byref(nc),
                                                                           ctypes.byref(io), ptr(cycles), stream_ptr()))):
        call()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        call()
        ev1.record()
        torch.cuda.synchronize()
        out[name + "_ms"] = ev0.elapsed_time(ev1)
        out[name + "_us_per_step"] = out[name + "_ms"] * 1e3 / t_len
    c = cycles.cpu().numpy().astype(np.float64) / t_len
    out["cycles_per_step"] = {"%d %s" % (i, p): round(float(v), 1) for i, (p, v) in enumerate(zip(ROLLOUT_PHASES, c))}
    out["cycles_per_step_total"] = round(float(c[:7].sum()), 1)
    return out


if __name__ == "__main__":
    main()
