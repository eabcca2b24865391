"""Host-side logic of deeprl_amd/nets.py that needs no GPU: the ReLU-mask hand-over between layer Functions and the in-place
parameter-gradient switch (round 4)."""
import torch


def test_relu_mask_mark_is_consumed_once_and_checks_address_and_size():
    from deeprl_amd import nets
    dx = torch.zeros(6, 4)
    nets._mark_masked(dx)
    assert nets._already_masked(dx.view(6, 4)) is True          # the layer below receives that very tensor (a view of it)
    assert nets._already_masked(dx) is False                    # ... once
    nets._mark_masked(dx)
    other = torch.zeros(6, 4)
    assert nets._already_masked(other) is False                 # another tensor: masked as before
    assert nets._already_masked(dx) is False                    # and the stale mark is gone
    nets._mark_masked(dx)
    assert nets._already_masked(dx[:3]) is False                # same address, different extent: not the marked gradient


def test_direct_param_grads_switch_nests_and_restores():
    from deeprl_amd import nets
    assert nets._DIRECT[0] is False
    with nets.direct_param_grads():
        assert nets._DIRECT[0] is True
        with nets.direct_param_grads(False):
            assert nets._DIRECT[0] is False
        assert nets._DIRECT[0] is True
    assert nets._DIRECT[0] is False
    try:
        with nets.direct_param_grads():
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert nets._DIRECT[0] is False


def test_grad_slot_needs_a_matching_dense_device_gradient():
    from deeprl_amd import nets
    p = torch.nn.Parameter(torch.zeros(3, 5))
    assert nets._grad_slot(p) is None and nets._grad_slot(None) is None       # no .grad yet
    p.grad = torch.zeros(3, 5)
    assert nets._grad_slot(p) is None                                         # a CPU gradient is never written in place


def test_nature_conv_body_marks_the_layers_that_read_a_fused_relu():
    from deeprl_amd import nets
    body = nets.NatureConvBody()
    assert not getattr(body.conv1, "input_is_relu", False)
    assert body.conv2.input_is_relu and body.conv3.input_is_relu and body.fc4.input_is_relu
    noisy = nets.NatureConvBody(noisy_linear=True)
    assert not getattr(noisy.fc4, "input_is_relu", False)


def test_direct_param_grads_claims_each_parameter_once():
    """nets.direct_param_grads: the first backward use of a parameter may overwrite its .grad, a second use inside the same
    backward pass (weight sharing) must go through autograd's accumulation; the record is cleared per pass."""
    import torch
    from deeprl_amd import nets
    p, q = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))
    assert not nets._claim_direct((p, None))            # outside the context: never direct
    with nets.direct_param_grads():
        assert nets._claim_direct((p, None))
        assert not nets._claim_direct((p, q))           # p already written: the pair falls back, q is not claimed by it
        assert nets._claim_direct((q,))
        with nets.direct_param_grads(False):
            assert not nets._claim_direct((torch.nn.Parameter(torch.zeros(1)),))
    with nets.direct_param_grads():
        assert nets._claim_direct((p, q))               # a new pass starts clean
    assert not nets._WRITTEN


def test_masked_gradient_mark_needs_a_live_tensor():
    """The 'already ReLU-masked' hand-off matches a gradient by address AND by the marked tensor still being alive: a stale
    mark (its tensor freed, the address recycled) must not make an unrelated gradient skip its mask."""
    import torch
    from deeprl_amd import nets
    a = torch.zeros(8)
    nets._mark_masked(a)
    assert nets._already_masked(a)
    assert not nets._already_masked(a)                  # consumed
    b = torch.zeros(8)
    nets._mark_masked(b)
    addr, numel = b.data_ptr(), b.numel()
    del b
    class Fake:                                         # an unrelated gradient that happens to get the same address / size
        def data_ptr(self): return addr
        def numel(self): return numel
    assert not nets._already_masked(Fake())


def test_shared_body_with_two_optimisers_cannot_run_in_the_reference_either():
    """PPOAgent refuses a parameterised phi_body shared by separate actor / critic optimisers (agents.py).  So does the
    reference, implicitly: PPO_agent.py:89-96 backpropagates policy_loss and then value_loss through ONE forward; with a shared
    parameterised body the first backward frees the shared graph (and actor_opt.step() rewrites its weights in place), so the
    second raises inside autograd.  Reproduced here with plain torch modules of the same structure."""
    import pytest
    import torch
    import torch.nn as nn
    torch.manual_seed(0)
    phi = nn.Sequential(nn.Linear(4, 8), nn.Tanh(), nn.Linear(8, 8), nn.Tanh())
    fa, fc = nn.Linear(8, 2), nn.Linear(8, 1)
    actor_opt = torch.optim.Adam(list(fa.parameters()) + list(phi.parameters()), 3e-4)
    critic_opt = torch.optim.Adam(list(fc.parameters()) + list(phi.parameters()), 1e-3)
    h = phi(torch.randn(16, 4))
    policy_loss, value_loss = fa(h).pow(2).mean(), fc(h).pow(2).mean()
    actor_opt.zero_grad()
    policy_loss.backward()
    actor_opt.step()
    critic_opt.zero_grad()
    with pytest.raises(RuntimeError):
        value_loss.backward()


def test_rollout_slots_hand_out_the_announced_rows_then_fall_back():
    """nets.RolloutSlots: begin(rows, n) draws ONE [rows, n] block from torch's generator; forward i of the rollout gets row i
    and the rows to write into; any other batch size, a forward beyond the announced ones, or one after end() gets (None, None)
    -- the caller draws torch.rand(B) itself; pin(u) overrides with a static row (a forward captured in its own graph)."""
    import torch
    from deeprl_amd.nets import RolloutSlots
    s = RolloutSlots()
    assert s.take(4) == (None, None) and s.next_uniform() is None
    torch.manual_seed(3)
    s.begin(3, 4)
    torch.manual_seed(3)
    ref = torch.empty(3, 4).uniform_()
    assert torch.equal(s.uniform, ref)
    u0, rows0 = s.take(4)
    assert torch.equal(u0, ref[0]) and rows0[0].data_ptr() == s.action[0].data_ptr() and rows0[3].data_ptr() == s.v[0].data_ptr()
    assert s.take(5) == (None, None)                      # another batch size does not consume a row
    assert torch.equal(s.next_uniform(), ref[1])
    u2, rows2 = s.take(4)
    assert torch.equal(u2, ref[2]) and rows2[1].data_ptr() == s.log_pi_a[2].data_ptr()
    assert s.take(4) == (None, None) and s.next_uniform() is None
    buf = s.uniform.data_ptr()
    s.begin(3, 4)                                          # same shape: same buffers (a captured graph keeps reading them)
    assert s.uniform.data_ptr() == buf and s.i == 0
    pinned = torch.zeros(4)
    s.pin(pinned)
    u, rows = s.take(4)
    assert u is pinned and rows is None and s.i == 0
    s.pin(None)
    s.end()
    assert s.take(4) == (None, None)


def test_deferred_folds_are_grouped_into_one_norm_launch_when_they_tile_the_buffer_from_zero(monkeypatch):
    """optim.FusedOptimizer.defer_fold / step (host logic, kernels mocked): conv layers whose gradient segments tile the flat
    buffer from offset 0 are folded by ONE dra_grad_sqnorm_segs call whose partial sums the step then reads; segments that do
    not (a gap, or not starting at 0) are folded one by one and the plain norm pass runs; a view that is not this optimizer's
    buffer / not 16-byte aligned is refused (the layer folds itself); zero_grad() forgets registered slabs."""
    import torch
    from deeprl_amd import ops, optim
    calls = []
    monkeypatch.setattr(ops, "grad_sqnorm_segs", lambda grad, segs, partials: calls.append(("segs", grad.numel(), [s[:2] + s[3:] for s in segs])) or 7)
    monkeypatch.setattr(ops, "grad_sqnorm", lambda grad, partials, **kw: calls.append(("norm", grad.numel())))
    monkeypatch.setattr(ops, "rmsprop_step", lambda *a, **k: calls.append(("step", a[5])))
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (32, 64, 16, 40)]
    opt = optim.FusedOptimizer.adopt(torch.optim.RMSprop(params, lr=1e-3))
    g = opt.flat.grad
    slabs = torch.zeros(4 * 64)
    assert opt.defer_fold(g[0:32], 32, slabs, 4) and opt.defer_fold(g[96:112], 16, slabs, 2) and opt.defer_fold(g[32:96], 64, slabs, 3)
    assert not opt.defer_fold(torch.zeros(32), 32, slabs, 4)            # another buffer
    assert not opt.defer_fold(g[2:34], 32, slabs, 4)                    # not 16-byte aligned
    opt.step(5.0)
    assert calls == [("segs", 152, [(0, 32, 32, 4), (32, 64, 64, 3), (96, 16, 16, 2)]), ("step", 7)] and not opt._pending_folds
    calls.clear()
    assert opt.defer_fold(g[32:96], 64, slabs, 3) and opt.defer_fold(g[96:112], 16, slabs, 2)      # does not start at 0
    opt.step(5.0)
    assert [c[0] for c in calls] == ["segs", "segs", "norm", "step"] and calls[0][1] == 64 and calls[-1][1] == opt.n_partials
    calls.clear()
    assert opt.defer_fold(g[0:32], 32, slabs, 4)
    opt.zero_grad()
    opt.step(None)
    assert calls == [("step", opt.n_partials)]


def test_zero_grad_is_skipped_only_after_a_backward_that_overwrote_every_gradient():
    """nets.direct_param_grads(covers=[opt]) + FusedOptimizer.zero_grad(direct=True): the fill is dead work when the previous
    direct backward claimed every parameter exactly once; a parameter reached twice (accumulation) or never brings it back."""
    import torch
    from deeprl_amd import nets, optim
    params = [torch.nn.Parameter(torch.zeros(8)) for _ in range(3)]
    opt = optim.FusedOptimizer.adopt(torch.optim.RMSprop(params, lr=1e-3))
    fills = []
    opt.flat.zero_grad = lambda: fills.append(1)
    opt.zero_grad(direct=True)
    assert fills == [1]                                   # nothing known yet
    with nets.direct_param_grads(True, covers=[opt]):
        assert nets._claim_direct(params[:2]) and nets._claim_direct(params[2:])
    opt.zero_grad(direct=True)
    assert fills == [1] and opt.all_direct                # every gradient will be overwritten again
    opt.zero_grad()                                       # a caller that does not promise a direct backward still gets its fill
    assert fills == [1, 1]
    with nets.direct_param_grads(True, covers=[opt]):
        assert nets._claim_direct(params[:2]) and not nets._claim_direct(params[1:])      # params[1] reached twice
    opt.zero_grad(direct=True)
    assert fills == [1, 1, 1] and not opt.all_direct
    with nets.direct_param_grads(True, covers=[opt]):
        assert nets._claim_direct(params[:2])                                                # params[2] never written
    assert not opt.all_direct
