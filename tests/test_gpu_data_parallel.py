"""Data-parallel A2C / PPO (BASELINE configs[4]) as the AGENTS run it: 2 ranks x 2 environments against
1 process x 4 environments, three updates each.  The ranks partition the environments, draw the same permutations,
normalise advantages over the global rollout, exchange ONE flat gradient per optimizer step, and must land on the
single-process parameters (fp32: the batch reduction is split differently, nothing else differs).

The GPU box has one GPU: both ranks share it and torch.distributed's gloo backend carries the gradients; on a node
where every rank owns a GPU the same agents use csrc/comm.hip (dra_allreduce_grads over RCCL), exercised here with a
communicator of size 1."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def _run(kind, out, world, port, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if world == 1:
        cmd = [sys.executable, WORKER, kind, out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port), WORKER, kind, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


@pytest.mark.parametrize("kind,port", [("a2c", 29641), ("ppo", 29642)])
def test_two_ranks_equal_one_process(tmp_path, kind, port):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    one = _run(kind, str(tmp_path / "one.npz"), 1, port)
    two = _run(kind, str(tmp_path / "two.npz"), 2, port)
    assert int(one["total_steps"]) == int(two["total_steps"]) > 0      # both count GLOBAL environment steps
    moved = 0.0
    from parity_log import record_parity
    steps = int(one["__adam_steps"]) if "__adam_steps" in one else 0
    for k in one:
        if k == "total_steps" or k.startswith("__adam"):
            continue
        # rtol 1e-5 / atol 2e-6 absolute on O(0.05) weights: the two runs differ only in how the batch sum is split.
        err = np.abs(two[k] - one[k])
        allowed = 2e-6 + 1e-5 * np.abs(one[k])
        if kind == "ppo":
            # Adam (eps 1e-8, examples.py:534) normalises every gradient element by its own RMS: u = lr * m / (sqrt(v) + eps).
            # A gradient element that differs by the contraction noise e_g between the two runs moves the update by
            # lr * e_g / sqrt(v_i) to first order -- negligible where |g_i| >> e_g, a whole step of lr where the element IS
            # noise.  With e_g = 1e-5 of the tensor's largest gradient RMS (the contraction bar of test_gpu_kernels.py),
            # S Adam steps and a factor 4 for the second-order terms (v and the later gradients see the perturbation too):
            #     |dp_i| <= 2e-6 + 1e-5 |p_i| + 4 S lr min(1, e_g / sqrt(v_i))
            # v_i = the bias-corrected second-moment estimate the single-process run ends with (written by dp_worker.py).
            v = one["__adam_v." + k] / (1.0 - 0.999 ** steps)
            rms = np.sqrt(np.maximum(v, 0.0))
            e_g = 1e-5 * float(rms.max())
            allowed = allowed + 4.0 * steps * 2.5e-4 * np.minimum(1.0, e_g / np.maximum(rms, 1e-30))
        ratio = float((err / allowed).max())
        record_parity("data_parallel %s %s" % (kind, k), err_max=float(err.max()), err_over_allowed=ratio,
                      frac_over_plain=float((err > 2e-6 + 1e-5 * np.abs(one[k])).mean()))
        assert ratio <= 1.0, "%s: max error %.3g = %.2f of the per-element bound" % (k, err.max(), ratio)
        moved = max(moved, float(np.abs(one[k]).max()))
    assert moved > 0


def test_ranks_seeded_differently_start_from_rank0_parameters(tmp_path):
    """ADVICE r2 (launch.py seeds 1 + RANK under torchrun): ranks that build their networks from DIFFERENT torch seeds must
    still train ONE model -- DataParallel broadcasts rank 0's flat parameters / optimizer state at agent construction.
    Two ranks with seeds (0, 1) == one process with seed 0, and both ranks end on identical parameters."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    one = _run("a2c", str(tmp_path / "one.npz"), 1, 29643)
    out2 = str(tmp_path / "two.npz")
    two = _run("a2c", out2, 2, 29643, {"DP_WORKER_RANK_SEEDS": "1"})
    r0, r1 = dict(np.load(out2 + ".rank0.npz")), dict(np.load(out2 + ".rank1.npz"))
    for k in one:
        if k == "total_steps":
            continue
        assert np.array_equal(r0[k], r1[k]), k                     # same bits on both ranks
        err = np.abs(two[k] - one[k])
        assert (err <= 2e-6 + 1e-5 * np.abs(one[k])).all(), (k, err.max())


@pytest.mark.parametrize("kind,port", [("a2c", 29644), ("ppo", 29645)])
def test_rccl_two_ranks_equal_gloo_two_ranks(tmp_path, kind, port):
    """On a node where two ranks can each own a GPU: the same agents with RCCL carrying the gradient (comm.hip's
    ncclAllReduce, n_ranks = 2: the branch a one-GPU box cannot execute) end on the same parameters, bit for bit, as with gloo
    (a two-rank sum is one addition per element whichever library performs it)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on one node (the round's test box has one)")
    g = _run(kind, str(tmp_path / "gloo.npz"), 2, port)
    n = _run(kind, str(tmp_path / "rccl.npz"), 2, port + 10, {"DP_WORKER_BACKEND": "nccl"})
    for k in g:
        assert np.array_equal(g[k], n[k]), k


def test_rccl_comm_of_size_one():
    """csrc/comm.hip through its C ABI: unique id, init_rank, dra_allreduce_grads (this rank's scale, then the sum over 1 rank),
    dra_allreduce_f64, destroy."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import deeprl_amd as d
    from deeprl_amd.dist import RcclComm
    d.select_device(0)
    comm = RcclComm()
    assert comm.world == 1 and comm.rank == 0
    g = torch.arange(1, 1 + 4099, dtype=torch.float32, device=d.Config.DEVICE)
    want = (g * 0.25).cpu().numpy()
    comm.allreduce_grads(g, 0.25)
    torch.cuda.synchronize()
    assert np.array_equal(g.cpu().numpy(), want)
    comm.close()


def test_comm_check_on_one_rank():
    """`bench.py --comm-check`: on the one-GPU box the self-diagnosis runs over a communicator of size 1
    and reports what a SCALE run would read first -- devices, RCCL's own rank count, the 6.75 MB exchange timed alone."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--comm-check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    # (RCCL's own NCCL_DEBUG=INFO banner shares the stream and is flushed at exit: take the JSON line, not the last line)
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"comm_check"')][-1])
    assert rec["comm_check"] and rec["enough_devices"] and rec["world_size"] == 1
    one = rec["per_rank"][0]
    assert one["rccl_ranks"] == 1 and one["rccl_rank"] == 0 and 0.0 < one["allreduce_us"] < 1e4
