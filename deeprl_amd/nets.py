"""Network bodies and heads with the reference's module / parameter names (so state_dicts
interchange with deep_rl/network/*), whose contractions run on the hand-written fp32-MFMA HIP
kernels through autograd Functions.

Reference: deep_rl/network/network_utils.py:23-83 (layer_init, NoisyLinear),
network_bodies.py:10-82 (NatureConvBody, DDPGConvBody, FCBody, DummyBody), network_heads.py:11-293
(the `forward -> dict` heads).  Softmax / distribution glue in the heads stays PyTorch
(element-wise, bandwidth-trivial); every conv / linear forward, input-gradient and
weight-gradient is a HIP kernel, and the fused learner (deeprl_amd/learner.py) bypasses autograd
altogether for the DQN family.
"""
import math

import contextlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .support import Config, tensor


class BaseNet:
    def __init__(self):
        pass

    def reset_noise(self):
        pass


def layer_init(layer, w_scale=1.0):
    """network_utils.py:23-27: orthogonal weights (scaled), zero bias -- drawn on the CPU generator
    before the module moves to the device, so seeds reproduce the reference's initial weights."""
    nn.init.orthogonal_(layer.weight.data)
    layer.weight.data.mul_(w_scale)
    nn.init.constant_(layer.bias.data, 0)
    return layer


# ----------------------------------------------------------------------------------------------------
# Parameter gradients written IN PLACE.  autograd hands a parameter's gradient to AccumulateGrad, which ADDS it to the existing
# .grad -- one elementwise launch per parameter (12 per update of the actor-critic pixel nets: 9 % of an A2C agent step's kernel
# time, profiles/r04ab_kernel_stats_a2c_pixel_16.txt), and for fc4 a 6.4 MB read-modify-write of a buffer that was just zeroed.
# Inside `with direct_param_grads():` the layer Functions below write their weight / bias gradients straight into the
# parameters' .grad (the views of the optimizer's flat gradient buffer) and return None for them.  That is an OVERWRITE: only
# for a backward pass in which every such parameter is used ONCE (A2C's batched update, a PPO minibatch) -- 0 + g == g, so the
# result is what the accumulation gives.
_DIRECT = [False]
_WRITTEN = set()        # ids of the parameters whose .grad the current backward pass has already overwritten


_SHARED = [False]       # the current backward pass reached some parameter a second time
_DEFER = [None]         # the optimizer (optim.FusedOptimizer) that will fold the conv layers' slabs in its own norm launch


@contextlib.contextmanager
def direct_param_grads(enable=True, defer_folds_to=None, covers=()):
    """Inside the context the layer Functions write each parameter's gradient straight into its .grad the FIRST time the
    backward pass reaches it; a parameter reached again (a module applied twice in one forward: weight sharing, siamese or
    recurrent bodies) gets its further contributions through autograd's own accumulation, on top of the direct write -- the sum
    is what plain accumulation gives (a second overwrite would silently drop the first contribution).
    defer_folds_to (a FusedOptimizer whose step() follows this backward with nothing reading the gradient in between): a conv
    layer whose gradient segment lives in that optimizer's flat buffer leaves its per-(sample, chunk) slabs unfolded and
    registers them (FusedOptimizer.defer_fold); the optimizer folds every registered layer inside the launch that forms the
    gradient norm -- one launch instead of one per layer plus the norm's (profiles/r05r_kernel_stats_*: fold_norm_kernel x 3 +
    grad_sqnorm_kernel per update)."""
    prev, prev_defer = _DIRECT[0], _DEFER[0]
    _DIRECT[0] = bool(enable)
    _DEFER[0] = defer_folds_to if enable else None
    if enable:
        _WRITTEN.clear()
        _SHARED[0] = False
    try:
        yield
    finally:
        # `covers`: did this backward overwrite EVERY parameter of these optimizers, each exactly once?  Then the zero fill in
        # front of the next such backward is dead work (FusedOptimizer.zero_grad(direct=True) skips it)
        for opt in (covers if enable else ()):
            opt.all_direct = (not _SHARED[0]) and all(id(p) in _WRITTEN for p in opt.flat.params)
        _DIRECT[0] = prev
        _DEFER[0] = prev_defer
        _WRITTEN.clear()


def _claim_direct(params):
    """True when direct writes are on and NONE of `params` has been written in this backward pass; claims them."""
    if not _DIRECT[0]:
        return False
    ids = [id(p) for p in params if p is not None]
    if any(i in _WRITTEN for i in ids):
        _SHARED[0] = True       # a parameter reached twice: its further contributions ACCUMULATE (the buffer must start at zero)
        return False
    _WRITTEN.update(ids)
    return True


# ReLU masks applied by the PRODUCER of a gradient.  A fused-ReLU layer's backward starts with dpre = dy * (y > 0) -- a launch of
# its own in front of every conv layer's backward.  The layer ABOVE knows the same mask (its input x IS that y) and its
# input-gradient kernel has an epilogue for it (xact): it hands down dx * (x > 0) and leaves the tensor's address here; the layer
# below skips its own mask when the gradient it receives is that very tensor (anything autograd copied or accumulated in between
# has another address and is masked as before).  (y > 0) in {0, 1}: the values are the same either way.
_MASKED = [0, 0, None]       # (address, element count, weak reference) of the gradient the layer above has already masked


def _mark_masked(dx):
    import weakref
    _MASKED[0], _MASKED[1], _MASKED[2] = dx.data_ptr(), dx.numel(), weakref.ref(dx)


def _already_masked(dy):
    """The mark names a tensor OBJECT that is still alive (a freed tensor's address may be handed to an unrelated gradient by
    the caching allocator: a stale mark must not match it) with the address / size this gradient has (autograd may hand the
    consumer a view or the same storage through another Python object)."""
    ref = _MASKED[2]
    alive = ref is not None and ref() is not None
    hit = alive and _MASKED[0] != 0 and dy.data_ptr() == _MASKED[0] and dy.numel() == _MASKED[1]
    _MASKED[0] = _MASKED[1] = 0          # consumed (or not ours): a mark never outlives the next layer's backward
    _MASKED[2] = None
    return hit


def _grad_slot(p):
    g = None if p is None else p.grad
    return g if (g is not None and g.dtype == torch.float32 and g.is_cuda and g.shape == p.shape) else None


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act, x_relu=False):
        y = ops.linear_fwd([x], [w], [b], act=act)[0]
        ctx.save_for_backward(x, w, y)
        ctx.act = act
        ctx.x_relu = bool(x_relu)       # x is the output of a fused-ReLU layer (NatureConvBody: conv3 -> fc4)
        ctx.has_bias = b is not None
        ctx.params = (w, b)             # (the Parameter objects themselves: their .grad is where a direct write goes)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        # (a consumer that knows this layer ends in a fused ReLU hands the gradient of the pre-activation over: _PolicyHeadFn)
        dpre = dy if (ctx.act == "relu" and _already_masked(dy)) else (ops.act_bwd(dy, y, ctx.act) if ctx.act else dy)
        gw, gb = (_grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])) if _claim_direct(ctx.params) else (None, None)
        direct = gw is not None and gw.is_contiguous() and (not ctx.has_bias or (gb is not None and gb.is_contiguous()))
        if (direct and ctx.needs_input_grad[0] and w.shape[0] == 512 and w.shape[1] >= 1024 and w.is_contiguous()
                and x.is_contiguous()):
            # fc4-shaped: both gradients in one launch (the two problems share dy and nothing else)
            dx, _, _ = ops.linear_bwd_xw_512(dpre, x, w, ctx.x_relu, dw=gw, db=gb if ctx.has_bias else None, want_bias=ctx.has_bias)
            if ctx.x_relu:
                _mark_masked(dx)
            return dx, None, None, None, None
        if direct:
            ops.linear_bwd_w(dpre, x, dw=gw, db=gb if ctx.has_bias else None, want_bias=ctx.has_bias)
            dw = db = None
        else:
            _SHARED[0] = True       # (claimed or not, these gradients reach .grad through autograd's accumulation)
            dw, db = ops.linear_bwd_w(dpre, x, want_bias=ctx.has_bias)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.x_relu:
                dx = ops.linear_bwd_x(dpre, w, xact=x, act="relu")
                _mark_masked(dx)
            else:
                dx = ops.linear_bwd_x(dpre, w)
        return dx, dw, db, None, None


class _LinearPairFn(torch.autograd.Function):
    """Two heads on the same features (fc_action / fc_critic of the actor-critic nets on phi): forward as ONE launch
    (ops.linear_fwd_pair: one wave per input row, bit-identical with two _LinearFn), backward = the two layers' own kernels,
    d phi = d phi_a + d phi_c as autograd's accumulation forms it."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1):
        y0, y1 = ops.linear_fwd_pair(x, w0, b0, w1, b1)
        ctx.save_for_backward(x, w0, w1)
        ctx.params = (w0, b0, w1, b1)
        return y0, y1

    @staticmethod
    def backward(ctx, g0, g1):
        x, w0, w1 = ctx.saved_tensors
        if g0 is not None and g1 is not None:
            # both gradients present (the usual case): ONE launch for d phi and both layers' weight / bias gradients
            direct = _claim_direct(ctx.params)
            slots = [_grad_slot(p) if direct else None for p in ctx.params]
            direct = all(g is not None and g.is_contiguous() for g in slots)
            _SHARED[0] = _SHARED[0] or not direct
            dx, dw0, db0, dw1, db1 = ops.linear_bwd_pair(g0.contiguous(), g1.contiguous(), x, w0, w1,
                                                         *((slots[0], slots[1], slots[2], slots[3]) if direct else ()),
                                                         want_dx=bool(ctx.needs_input_grad[0]))
            return (dx, None, None, None, None) if direct else (dx, dw0, db0, dw1, db1)
        outs, dx = [], None
        _SHARED[0] = True
        for g, w, pw, pb in ((g0, w0, ctx.params[0], ctx.params[1]), (g1, w1, ctx.params[2], ctx.params[3])):
            if g is None:
                outs += [None, None]
                continue
            g = g.contiguous()
            outs += list(ops.linear_bwd_w(g, x, want_bias=True))
            if ctx.needs_input_grad[0]:
                d = ops.linear_bwd_x(g, w)
                dx = d if dx is None else dx + d
        return (dx,) + tuple(outs)


class _PolicyHeadFn(torch.autograd.Function):
    """The whole policy head of the update's forward -- fc_action / fc_critic on the shared features, Categorical(logits) of the
    stored actions -> (log_pi_a, entropy, v) [B] -- as ONE launch each way (ops.policy_heads_given / policy_heads_bwd; before: heads,
    categorical forward | categorical backward, paired-heads backward, the ReLU mask of the layer below).  x_relu: the features
    are a fused-ReLU output, the input gradient comes back as the gradient of its pre-activation (marked, _already_masked)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, action, x_relu):
        lp, ent, v, logits = ops.policy_heads_given(x, w0, b0, w1, b1, action)
        ctx.save_for_backward(x, w0, w1, logits, action)
        ctx.params = (w0, b0, w1, b1)
        ctx.x_relu = bool(x_relu)
        return lp, ent, v

    @staticmethod
    def backward(ctx, g_lp, g_ent, g_v):
        x, w0, w1, logits, action = ctx.saved_tensors
        direct = _claim_direct(ctx.params)
        slots = [_grad_slot(p) if direct else None for p in ctx.params]
        direct = all(g is not None and g.is_contiguous() for g in slots)
        _SHARED[0] = _SHARED[0] or not direct
        want_dx = bool(ctx.needs_input_grad[0])
        dx, dw0, db0, dw1, db1 = ops.policy_heads_bwd(logits, action, g_lp, g_ent, g_v, x, w0, w1,
                                                      *((slots[0], slots[1], slots[2], slots[3]) if direct else ()),
                                                      want_dx=want_dx, relu_mask=ctx.x_relu and want_dx)
        if ctx.x_relu and dx is not None:
            _mark_masked(dx)
        return (dx, None, None, None, None, None, None) if direct else (dx, dw0, db0, dw1, db1, None, None)


class _Fc4PolicyHeadFn(torch.autograd.Function):
    """NatureConvBody's fc4 (+ ReLU) and the policy head of the update's forward as one autograd node: forward = fc4's one-pass
    K-slice launch + the head launch that folds the slices (ops.fc4_policy_heads_given: no finish launch), backward = the head's
    one launch (ReLU mask included) + fc4's one launch (input and weight gradients as two roles).  Same arithmetic as
    _LinearFn followed by _PolicyHeadFn."""

    @staticmethod
    def forward(ctx, y3, w4, b4, w0, b0, w1, b1, action, phi_pre=None):
        if phi_pre is not None:     # fc4's output for exactly these inputs and parameters, computed by the rollout (see _ConvKocFn.forward)
            phi = phi_pre.view_as(phi_pre)
            lp, ent, v, logits = ops.policy_heads_given(phi, w0, b0, w1, b1, action)
        else:
            lp, ent, v, logits, phi = ops.fc4_policy_heads_given(y3, w4, b4, w0, b0, w1, b1, action)
        ctx.save_for_backward(y3, w4, phi, w0, w1, logits, action)
        ctx.params = (w4, b4, w0, b0, w1, b1)
        return lp, ent, v

    @staticmethod
    def backward(ctx, g_lp, g_ent, g_v):
        y3, w4, phi, w0, w1, logits, action = ctx.saved_tensors
        p4, pb4, p0, pb0, p1, pb1 = ctx.params
        direct_h = _claim_direct((p0, pb0, p1, pb1))
        hs = [_grad_slot(p) if direct_h else None for p in (p0, pb0, p1, pb1)]
        direct_h = all(g is not None and g.is_contiguous() for g in hs)
        dphi, dw0, db0, dw1, db1 = ops.policy_heads_bwd(logits, action, g_lp, g_ent, g_v, phi, w0, w1,
                                                        *((hs[0], hs[1], hs[2], hs[3]) if direct_h else ()), relu_mask=True)
        direct_4 = _claim_direct((p4, pb4))
        g4 = [_grad_slot(p) if direct_4 else None for p in (p4, pb4)]
        direct_4 = all(g is not None and g.is_contiguous() for g in g4)
        _SHARED[0] = _SHARED[0] or not (direct_h and direct_4)
        dy3, dw4, db4 = ops.linear_bwd_xw_512(dphi, y3, w4, True, *((g4[0], g4[1]) if direct_4 else ()))
        _mark_masked(dy3)
        if not ctx.needs_input_grad[0]:
            dy3 = None
        return (dy3,) + ((None, None) if direct_4 else (dw4, db4)) + ((None,) * 4 if direct_h else (dw0, db0, dw1, db1)) + (None, None)


class _CategoricalFn(torch.autograd.Function):
    """Categorical(logits) -> (log_pi_a, entropy) for given actions as one kernel each way (losses.hip K12)."""

    @staticmethod
    def forward(ctx, logits, action):
        _, lp, ent = ops.categorical_fwd(logits, action=action)
        ctx.save_for_backward(logits, action)
        return lp, ent

    @staticmethod
    def backward(ctx, g_lp, g_ent):
        logits, action = ctx.saved_tensors
        return ops.categorical_bwd(logits, action, g_lp, g_ent), None


class RolloutSlots:
    """Where a categorical actor-critic's no-grad forwards of ONE rollout get their uniforms and put their results.  The
    reference draws inside every forward (network_heads.py:251 dist.sample()); here the agent announces a rollout of `rows`
    forwards over `n` environments (begin: ONE uniform_() on torch's global device generator for all of them) and forward number i
    reads row i and writes action / log_pi_a / entropy / v into row i of persistent [rows, n] buffers -- the agent reads the
    rollout back as views, without the per-key torch.stack / torch.cat launches, and the sampling launch per step disappears
    into the head kernel (ops.policy_heads_sample).  Forwards outside an announced rollout (evaluation episodes, another batch
    size, more forwards than announced) draw torch.rand(B) as before.  `pin(u)`: every forward reads the given static [n] buffer
    and allocates its outputs (a forward captured in a hipGraph of its own: PPOAgent._act copies next_uniform() into it)."""

    def __init__(self):
        self.rows, self.n, self.i = 0, 0, 0
        self.pinned = None
        self.uniform = self.action = self.log_pi_a = self.entropy = self.v = None

    def begin(self, rows, n):
        if (rows, n) != (self.rows, self.n):
            dev = Config.DEVICE
            self.uniform = torch.empty((rows, n), dtype=torch.float32, device=dev)
            self.action = torch.zeros((rows, n), dtype=torch.int64, device=dev)
            self.log_pi_a, self.entropy, self.v = (torch.zeros((rows, n), dtype=torch.float32, device=dev) for _ in range(3))
            self.rows, self.n = rows, n
        self.uniform.uniform_()
        self.i = 0

    def end(self):
        self.i = self.rows          # later forwards draw for themselves

    def next_uniform(self):
        """Row i of the announced draw (None outside a rollout); advances."""
        if self.i >= self.rows:
            return None
        self.i += 1
        return self.uniform[self.i - 1]

    def take(self, b):
        """-> (uniform [b] or None, (action, log_pi_a, entropy, v) rows to write into or None) for a forward over b samples."""
        if self.pinned is not None and self.pinned.numel() == b:
            return self.pinned, None
        if self.i >= self.rows or b != self.n:
            return None, None
        i = self.i
        self.i += 1
        return self.uniform[i], (self.action[i], self.log_pi_a[i], self.entropy[i], self.v[i])

    def pin(self, u):
        self.pinned = u


def categorical_policy(logits, action=None, sampler=None, uniform=None):
    """network_heads.py:249-254: (action, log_pi_a [B,1], entropy [B,1]) of Categorical(logits=logits).  action None: drawn
    by `sampler(logits)` when given, else by inverse CDF from `uniform` [B] (a row of the rollout's one draw: RolloutSlots) or
    one torch.rand(B) on torch's global device generator."""
    logits = logits.float().contiguous()
    if action is None:
        if sampler is not None:
            action = sampler(logits)
        else:
            with torch.no_grad():
                u = uniform if uniform is not None else torch.rand(logits.shape[0], dtype=torch.float32, device=logits.device)
                action, lp, ent = ops.categorical_fwd(logits.detach(), uniform=u)
            if not (torch.is_grad_enabled() and logits.requires_grad):
                # a rollout step (no_grad): the sampling launch already produced log_pi_a / entropy of its own draw
                return action.long().reshape(-1).contiguous(), lp.unsqueeze(-1), ent.unsqueeze(-1)
    action = action.long().reshape(-1).contiguous()
    lp, ent = _CategoricalFn.apply(logits, action)
    return action, lp.unsqueeze(-1), ent.unsqueeze(-1)


def linear(x, w, b, act=None, x_relu=False):
    x = x.float() if x.dtype != torch.float32 else x
    lead = x.shape[:-1]
    y = _LinearFn.apply(x.reshape(-1, x.shape[-1]).contiguous(), w, b, act, x_relu)
    y = y.reshape(lead + (w.shape[0],))
    if act == "relu":
        y.dra_fused_relu = True     # (a consumer may hand the gradient of the pre-activation back: _PolicyHeadFn, _already_masked)
    return y


class _ConvFn(torch.autograd.Function):
    KSPLIT = 16

    @staticmethod
    def forward(ctx, x, w, b, layer, u8_coef):
        y = ops.conv_fwd(layer, [x], [w], [b], act="relu", u8_coef=u8_coef)[0]
        ctx.save_for_backward(x, w, y)
        ctx.layer, ctx.u8_coef = layer, u8_coef
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        dpre = dy if _already_masked(dy) else ops.act_bwd(dy, y, "relu")
        n_w, n_b = w.numel(), w.shape[0]
        dw_s, db_s = ops.conv_bwd_w(ctx.layer, dpre, x, ksplit=_ConvFn.KSPLIT, u8_coef=ctx.u8_coef)
        stride = dw_s.stride(0)
        flat = torch.empty(stride, dtype=torch.float32, device=w.device)
        partials = torch.empty(ops.norm_partials(), dtype=torch.float64, device=w.device)
        ops.grad_sqnorm(flat, partials, slabs=dw_s, n_slabs=_ConvFn.KSPLIT, slab_stride=stride)  # fold split-K slabs
        dw = flat[:n_w].view_as(w)
        db = flat[n_w:n_w + n_b]
        dx = None
        if ctx.needs_input_grad[0] and ctx.layer > 1:
            dx = ops.conv_bwd_x(ctx.layer, dpre, w)
        return dx, dw, db, None, None


import os as _os
# one slab per (sample, row chunk) + the segmented fold at the on-policy minibatch sizes: a2c_pixel (batch 80) 106.6k -> 114.9k,
# ppo_pixel (batch 256) 56.6k -> 65.2k env-steps/s against the fixed split-K weight gradient (profiles/r02zw_onpolicy_*.jsonl).
# Round 4 measured the backward launches at batch 512 / 1024 both ways (profiles/r04f_conv_big.jsonl vs r04g_conv_big_oneshot.jsonl):
# conv1 286 / 568 us -> 57 / 109 us, conv2 175 / 362 -> 88 / 159 us, conv3 124 / 240 -> 68 / 128 us (35-44 % of the fp32-MFMA peak
# instead of 7-20 %); the slabs are 168 MB at batch 1024 (conv1), which is where the limit stays.
_ONESHOT_WGRAD_MAX_BATCH = int(_os.environ.get("DRA_ONESHOT_WGRAD_MAX_BATCH", "1024"))


class _ConvKocFn(torch.autograd.Function):
    """Same layer on the one-round-trip kernels (conv_v2.hip forward, oneshot.h backward: weight gradient and
    input gradient in ONE launch).  Used when the weight Parameter is a [OC,C,KH,KW] view of [(c,kh,kw)][oc]
    ("KOC") storage, which is how FlatParams stores the three NatureConvBody weights."""
    KSPLIT = 16

    @staticmethod
    def forward(ctx, x, w, b, layer, u8_coef, x_relu=False, y_pre=None):
        wt = w.permute(1, 2, 3, 0)                      # the contiguous KOC storage
        # y_pre: this layer's output for exactly this input and these parameters, computed earlier under no_grad (the A2C rollout's
        # forwards: the reference keeps the rollout's own forward graph, A2C_agent.py:29-64) -- no launch, the node only records
        # what its backward needs
        y = y_pre.view_as(y_pre) if y_pre is not None else ops.conv_fwd_koc(layer, [x], [wt], [b], act="relu", u8_coef=u8_coef)[0]
        ctx.save_for_backward(x, w, y)
        ctx.layer, ctx.u8_coef = layer, u8_coef
        ctx.params = (w, b)
        ctx.x_relu = bool(x_relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        layer = ctx.layer
        dy = dy.contiguous()
        dpre = dy if _already_masked(dy) else ops.act_bwd(dy, y, "relu")
        oc, c, kh, kw = w.shape
        n_w = w.numel()
        # per-(sample, row chunk) slabs pay off at DQN batch sizes; large batches use the fixed split-K weight gradient
        # (VAR_DGRAD_SCATTER acts from 256 samples on: conv2 / conv3 input gradient contracted over the output positions --
        # 46.8 -> 42.8 / 38.7 -> 30.5 us per layer at PPO's minibatch of 256, profiles/r06zzk_ab_dgrad_scatter.jsonl)
        variant = (ops.VAR_FUSED_BWD | ops.VAR_ONESHOT_DGRAD | ops.VAR_ONESHOT_WGRAD | ops.VAR_DGRAD_SCATTER) if x.shape[0] <= _ONESHOT_WGRAD_MAX_BATCH else \
            (ops.VAR_FUSED_BWD | ops.VAR_ONESHOT_DGRAD)
        # layers 2 / 3: the input is the layer below's fused-ReLU output (NatureConvBody) -- the input-gradient epilogue masks
        mask_below = layer > 1 and ctx.needs_input_grad[0] and getattr(ctx, "x_relu", False)
        dx, slabs, n_slabs, stride = ops.conv_bwd_fused_koc(layer, dpre, x, w.permute(1, 2, 3, 0), ksplit=_ConvKocFn.KSPLIT,
                                                            u8_coef=ctx.u8_coef, variant=variant, xact=x if mask_below else None)
        if mask_below:
            _mark_masked(dx)
        # direct_param_grads(): FlatParams lays [weight (KOC) | bias] out back to back, which is a slab's own layout -- the fold
        # writes the layer's gradient segment of the optimizer's flat buffer itself
        gw, gb = (_grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])) if _claim_direct(ctx.params) else (None, None)
        direct = (gw is not None and gb is not None and stride == n_w + oc and gw.permute(1, 2, 3, 0).is_contiguous()
                  and gb.is_contiguous() and gb.data_ptr() == gw.data_ptr() + 4 * n_w and gw.data_ptr() % 16 == 0)
        _SHARED[0] = _SHARED[0] or not direct
        if direct and _DEFER[0] is not None and _DEFER[0].defer_fold(gw, stride, slabs, n_slabs):
            return (dx if ctx.needs_input_grad[0] and layer > 1 else None), None, None, None, None, None, None
        flat = torch.as_strided(gw, (stride,), (1,)) if direct else torch.empty(stride, dtype=torch.float32, device=w.device)
        if n_slabs > 32 and stride % 4 == 0:
            # one slab per (sample, row chunk): hundreds of slabs -- the segmented fold keeps 160 of them in flight per element
            # (dra_grad_sqnorm's per-element serial walk is for the <= 64 slabs of the fixed split-K kernels)
            partials = torch.empty(ops.norm_partials_max(), dtype=torch.float64, device=w.device)
            ops.grad_sqnorm_segs(flat, [(0, stride, slabs, stride, n_slabs)], partials)
        else:
            partials = torch.empty(ops.norm_partials(), dtype=torch.float64, device=w.device)
            ops.grad_sqnorm(flat, partials, slabs=slabs, n_slabs=n_slabs, slab_stride=stride)  # fixed-order slab fold
        if direct:
            return (dx if ctx.needs_input_grad[0] and layer > 1 else None), None, None, None, None, None, None
        dw = flat[:n_w].view(c, kh, kw, oc).permute(3, 0, 1, 2)
        db = flat[n_w:n_w + oc]
        return (dx if ctx.needs_input_grad[0] and layer > 1 else None), dw, db, None, None, None, None


class Linear(nn.Linear):
    """nn.Linear whose forward / backward are the HIP contractions; `fused_act` folds the body's
    gate into the GEMM epilogue."""
    fused_act = None
    input_is_relu = False       # set by a body whose previous layer ends in a fused ReLU (see _MASKED)

    def forward(self, x):
        return linear(x, self.weight, self.bias, self.fused_act, self.input_is_relu)


class Conv2d(nn.Conv2d):
    """One of the three NatureConvBody convolutions (+ fused ReLU).  Other geometries have no HIP
    kernel in this build and raise rather than fall back."""

    def forward(self, x):
        layer = ops.conv_layer_for(self.weight.shape, self.stride[0], x.shape[-1])
        if layer is None or self.padding != (0, 0) or self.stride[0] != self.stride[1]:
            raise NotImplementedError("deeprl_amd has HIP kernels for the NatureConvBody convolutions only; got "
                                      "weight %s stride %s input %s" % (tuple(self.weight.shape), self.stride,
                                                                        tuple(x.shape)))
        # uint8 frames: the normaliser's coefficient travels on the tensor (RescaleNormalizer) or on the layer (device-resident
        # rollouts hand conv1 the raw frames: agents._device_state_fn)
        u8_coef = (getattr(x, "dra_u8_coef", None) or getattr(self, "u8_coef", None)) if x.dtype == torch.uint8 else None
        if x.dtype == torch.uint8 and u8_coef is None:
            raise TypeError("uint8 input to a convolution needs a normaliser (RescaleNormalizer marks it)")
        if self.weight.permute(1, 2, 3, 0).is_contiguous():     # KOC storage (FlatParams): one-round-trip kernels
            # (_y_pre: set by A2CAgent for ONE call -- the rollout's stored output of this layer for this batch, see _ConvKocFn.forward)
            y_pre = self.__dict__.pop("_y_pre", None)
            if y_pre is not None and (not torch.is_grad_enabled() or y_pre.shape[0] != x.shape[0]):
                y_pre = None
            return _ConvKocFn.apply(x.contiguous(), self.weight, self.bias, layer, u8_coef, getattr(self, "input_is_relu", False), y_pre)
        return _ConvFn.apply(x.contiguous(), self.weight, self.bias, layer, u8_coef)


# ---------------------------------------------------------------------------------------------------- NoisyLinear
class NoisyLinear(nn.Module):
    """Linear layer with factorised Gaussian parameter noise (Fortunato et al.; parameter / buffer names and initial
    values of network_utils.py:31-83 so that Rainbow checkpoints interchange):  y = x (W_mu + W_sigma * eps_W)^T +
    (b_mu + b_sigma * eps_b),  eps_W = f(e_out) f(e_in)^T,  eps_b = f(e_b),  f(e) = sign(e) sqrt|e|,  e ~ N(0, NOISY_LAYER_STD^2)
    redrawn by reset_noise().  Evaluation mode uses the means only.  The mixed weight feeds the HIP GEMM."""

    def __init__(self, in_features, out_features, std_init=0.4):
        super(NoisyLinear, self).__init__()
        self.in_features, self.out_features, self.std_init = in_features, out_features, std_init
        shape = (out_features, in_features)
        for name, s in (('weight', shape), ('bias', (out_features,))):
            setattr(self, name + '_mu', nn.Parameter(torch.zeros(s), requires_grad=True))
            setattr(self, name + '_sigma', nn.Parameter(torch.zeros(s), requires_grad=True))
            self.register_buffer(name + '_epsilon', torch.zeros(s))
        for name, n in (('noise_in', in_features), ('noise_out_weight', out_features), ('noise_out_bias', out_features)):
            self.register_buffer(name, torch.zeros(n))
        self.fused_act = None
        self.reset_parameters()
        self.reset_noise()

    def reset_parameters(self):
        bound = 1 / math.sqrt(self.in_features)
        for mu, sigma, fan in ((self.weight_mu, self.weight_sigma, self.in_features),
                               (self.bias_mu, self.bias_sigma, self.out_features)):
            mu.data.uniform_(-bound, bound)
            sigma.data.fill_(self.std_init / math.sqrt(fan))

    @staticmethod
    def transform_noise(x):
        return x.sign().mul(x.abs().sqrt())

    def reset_noise(self):
        """network_utils.py:73-80.  The three normal vectors are drawn from torch's CPU generator, in the reference's
        order and with its tensor sizes (the reference's buffers live on Config.DEVICE = CPU there), and uploaded: like
        every other random stream of the hot path the noise is host-drawn, so a seeded run consumes the generator exactly
        as the reference does.  The factorised products are formed on the device."""
        for e in (self.noise_in, self.noise_out_weight, self.noise_out_bias):       # this draw order
            e.copy_(torch.empty(e.shape, dtype=e.dtype).normal_(std=Config.NOISY_LAYER_STD))
        self.refresh_epsilon()

    def refresh_epsilon(self):
        """weight_epsilon = f(noise_out_weight) f(noise_in)^T, bias_epsilon = f(noise_out_bias) from the current noise vectors."""
        f = self.transform_noise
        self.weight_epsilon.copy_(torch.outer(f(self.noise_out_weight), f(self.noise_in)))
        self.bias_epsilon.copy_(f(self.noise_out_bias))

    def forward(self, x):
        if not self.training:
            return linear(x, self.weight_mu, self.bias_mu, self.fused_act)
        return linear(x, self.weight_mu + self.weight_sigma * self.weight_epsilon,
                      self.bias_mu + self.bias_sigma * self.bias_epsilon, self.fused_act)


# ---------------------------------------------------------------------------------------------------- bodies
class NatureConvBody(nn.Module):
    """network_bodies.py:10-33."""

    def __init__(self, in_channels=4, noisy_linear=False):
        super(NatureConvBody, self).__init__()
        self.feature_dim = 512
        self.conv1 = layer_init(Conv2d(in_channels, 32, kernel_size=8, stride=4))
        self.conv2 = layer_init(Conv2d(32, 64, kernel_size=4, stride=2))
        self.conv3 = layer_init(Conv2d(64, 64, kernel_size=3, stride=1))
        if noisy_linear:
            self.fc4 = NoisyLinear(7 * 7 * 64, self.feature_dim)
        else:
            self.fc4 = layer_init(Linear(7 * 7 * 64, self.feature_dim))
        self.fc4.fused_act = "relu"
        # conv2 / conv3 / fc4 read the fused-ReLU output of the layer below: their input-gradient kernels apply its mask (see _MASKED)
        self.conv2.input_is_relu = self.conv3.input_is_relu = True
        if not noisy_linear:
            self.fc4.input_is_relu = True
        self.noisy_linear = noisy_linear

    def reset_noise(self):
        if self.noisy_linear:
            self.fc4.reset_noise()

    def forward(self, x):
        y = self.conv1(x)  # ReLU fused into every layer's epilogue
        y = self.conv2(y)
        y = self.conv3(y)
        y = y.view(y.size(0), -1)
        return self.fc4(y)


class DDPGConvBody(nn.Module):
    """network_bodies.py:36-47 -- unused by every example; kept for name compatibility only."""

    def __init__(self, in_channels=4):
        super(DDPGConvBody, self).__init__()
        raise NotImplementedError("DDPGConvBody is outside the hot path (SURVEY.md section 2 #7): no HIP kernel")


_GATES = {F.relu: "relu", torch.relu: "relu", torch.tanh: "tanh", F.tanh: "tanh"}


class FCBody(nn.Module):
    """network_bodies.py:50-73: Linear + gate stack; relu / tanh gates fold into the GEMM epilogue."""

    def __init__(self, state_dim, hidden_units=(64, 64), gate=F.relu, noisy_linear=False):
        super(FCBody, self).__init__()
        dims = (state_dim,) + tuple(hidden_units)
        if noisy_linear:
            self.layers = nn.ModuleList([NoisyLinear(i, o) for i, o in zip(dims[:-1], dims[1:])])
        else:
            self.layers = nn.ModuleList([layer_init(Linear(i, o)) for i, o in zip(dims[:-1], dims[1:])])
        self.gate = gate
        self._fused = _GATES.get(gate)
        for layer in self.layers:
            layer.fused_act = self._fused
        self.feature_dim = dims[-1]
        self.noisy_linear = noisy_linear

    def reset_noise(self):
        if self.noisy_linear:
            for layer in self.layers:
                layer.reset_noise()

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
            if self._fused is None:
                x = self.gate(x)
        return x


class DummyBody(nn.Module):
    def __init__(self, state_dim):
        super(DummyBody, self).__init__()
        self.feature_dim = state_dim

    def forward(self, x):
        return x


# ---------------------------------------------------------------------------------------------------- heads
class VanillaNet(nn.Module, BaseNet):
    """network_heads.py:11-21."""

    def __init__(self, output_dim, body):
        super(VanillaNet, self).__init__()
        self.fc_head = layer_init(Linear(body.feature_dim, output_dim))
        self.body = body
        self.to(Config.DEVICE)

    def forward(self, x):
        phi = self.body(tensor(x))
        return dict(q=self.fc_head(phi))


def _head(features, outputs, w_scale=1.0, noisy=False):
    """A linear head on the HIP GEMM: orthogonal / zero-bias initialised (network_utils.py:23-27), or a NoisyLinear."""
    return NoisyLinear(features, outputs) if noisy else layer_init(Linear(features, outputs), w_scale)


class DuelingNet(nn.Module, BaseNet):
    """q = v + (adv - mean_a adv) over a shared body (module names of network_heads.py:24-37)."""

    def __init__(self, action_dim, body):
        super(DuelingNet, self).__init__()
        self.fc_value = _head(body.feature_dim, 1)
        self.fc_advantage = _head(body.feature_dim, action_dim)
        self.body = body
        self.to(Config.DEVICE)

    def forward(self, x, to_numpy=False):
        phi = self.body(tensor(x))
        adv = self.fc_advantage(phi)
        return dict(q=self.fc_value(phi) + adv - adv.mean(1, keepdim=True))


class CategoricalNet(nn.Module, BaseNet):
    """network_heads.py:40-54; also returns the pre-softmax `logits` the fused C51 loss consumes."""

    def __init__(self, action_dim, num_atoms, body):
        super(CategoricalNet, self).__init__()
        self.fc_categorical = layer_init(Linear(body.feature_dim, action_dim * num_atoms))
        self.action_dim = action_dim
        self.num_atoms = num_atoms
        self.body = body
        self.to(Config.DEVICE)

    def forward(self, x):
        phi = self.body(tensor(x))
        pre_prob = self.fc_categorical(phi).view((-1, self.action_dim, self.num_atoms))
        return dict(prob=F.softmax(pre_prob, dim=-1), log_prob=F.log_softmax(pre_prob, dim=-1), logits=pre_prob)


class RainbowNet(nn.Module, BaseNet):
    """Dueling heads over atoms: logits[a] = value + (advantage[a] - mean_a advantage), softmax over the atoms; the
    heads (and the body's fc4) are NoisyLinear when noisy_linear (module names of network_heads.py:57-86)."""

    def __init__(self, action_dim, num_atoms, body, noisy_linear):
        super(RainbowNet, self).__init__()
        self.fc_value = _head(body.feature_dim, num_atoms, noisy=noisy_linear)
        self.fc_advantage = _head(body.feature_dim, action_dim * num_atoms, noisy=noisy_linear)
        self.action_dim, self.num_atoms, self.body, self.noisy_linear = action_dim, num_atoms, body, noisy_linear
        self.to(Config.DEVICE)

    def reset_noise(self):
        if self.noisy_linear:
            for m in (self.fc_value, self.fc_advantage, self.body):
                m.reset_noise()

    def forward(self, x):
        phi = self.body(tensor(x))
        adv = self.fc_advantage(phi).view(-1, self.action_dim, self.num_atoms)
        logits = self.fc_value(phi).view(-1, 1, self.num_atoms) + adv - adv.mean(1, keepdim=True)
        return dict(prob=F.softmax(logits, dim=-1), log_prob=F.log_softmax(logits, dim=-1), logits=logits)


class QuantileNet(nn.Module, BaseNet):
    """network_heads.py:89-102."""

    def __init__(self, action_dim, num_quantiles, body):
        super(QuantileNet, self).__init__()
        self.fc_quantiles = layer_init(Linear(body.feature_dim, action_dim * num_quantiles))
        self.action_dim = action_dim
        self.num_quantiles = num_quantiles
        self.body = body
        self.to(Config.DEVICE)

    def forward(self, x):
        phi = self.body(tensor(x))
        quantiles = self.fc_quantiles(phi).view((-1, self.action_dim, self.num_quantiles))
        return dict(quantile=quantiles)


class OptionCriticNet(nn.Module, BaseNet):
    """Option-critic heads on one body: option values q [N, O], termination probabilities beta = sigmoid(.) [N, O] and
    one softmax policy per option pi [N, O, A] (+ its log) -- module names of network_heads.py:105-127."""

    def __init__(self, body, action_dim, num_options):
        super(OptionCriticNet, self).__init__()
        self.fc_q = _head(body.feature_dim, num_options)
        self.fc_pi = _head(body.feature_dim, num_options * action_dim)
        self.fc_beta = _head(body.feature_dim, num_options)
        self.num_options, self.action_dim, self.body = num_options, action_dim, body
        self.to(Config.DEVICE)

    def forward(self, x):
        phi = self.body(tensor(x))
        pi_logits = self.fc_pi(phi).view(-1, self.num_options, self.action_dim)
        return dict(q=self.fc_q(phi), beta=torch.sigmoid(self.fc_beta(phi)), log_pi=F.log_softmax(pi_logits, dim=-1),
                    pi=F.softmax(pi_logits, dim=-1))


def _ac_bodies(state_dim, phi_body, actor_body, critic_body):
    phi_body = phi_body if phi_body is not None else DummyBody(state_dim)
    actor_body = actor_body if actor_body is not None else DummyBody(phi_body.feature_dim)
    critic_body = critic_body if critic_body is not None else DummyBody(phi_body.feature_dim)
    return phi_body, actor_body, critic_body


class DeterministicActorCriticNet(nn.Module, BaseNet):
    """DDPG's pair on an optional shared feature body (module names of network_heads.py:130-170): actor(phi) =
    tanh(fc_action(actor_body(phi))), critic(phi, a) = fc_critic(critic_body([phi, a])); the network owns its two
    optimisers (actor_opt over actor + phi parameters, critic_opt over critic + phi parameters)."""

    def __init__(self, state_dim, action_dim, actor_opt_fn, critic_opt_fn, phi_body=None, actor_body=None,
                 critic_body=None):
        super(DeterministicActorCriticNet, self).__init__()
        self.phi_body, self.actor_body, self.critic_body = _ac_bodies(state_dim, phi_body, actor_body, critic_body)
        self.fc_action = _head(self.actor_body.feature_dim, action_dim, 1e-3)
        self.fc_critic = _head(self.critic_body.feature_dim, 1, 1e-3)
        own = lambda *mods: [p for m in mods for p in m.parameters()]
        self.actor_params, self.critic_params = own(self.actor_body, self.fc_action), own(self.critic_body, self.fc_critic)
        self.phi_params = own(self.phi_body)
        self.actor_opt = actor_opt_fn(self.actor_params + self.phi_params)
        self.critic_opt = critic_opt_fn(self.critic_params + self.phi_params)
        self.to(Config.DEVICE)

    def feature(self, obs):
        return self.phi_body(tensor(obs))

    def actor(self, phi):
        return torch.tanh(self.fc_action(self.actor_body(phi)))

    def critic(self, phi, a):
        return self.fc_critic(self.critic_body(torch.cat([phi, a], dim=1)))

    def forward(self, obs):
        return self.actor(self.feature(obs))


class GaussianActorCriticNet(nn.Module, BaseNet):
    """network_heads.py:173-214 (PPO / A2C continuous)."""

    def __init__(self, state_dim, action_dim, phi_body=None, actor_body=None, critic_body=None):
        super(GaussianActorCriticNet, self).__init__()
        self.phi_body, self.actor_body, self.critic_body = _ac_bodies(state_dim, phi_body, actor_body, critic_body)
        self.fc_action = layer_init(Linear(self.actor_body.feature_dim, action_dim), 1e-3)
        self.fc_critic = layer_init(Linear(self.critic_body.feature_dim, 1), 1e-3)
        self.std = nn.Parameter(torch.zeros(action_dim))
        self.phi_params = list(self.phi_body.parameters())
        self.actor_params = list(self.actor_body.parameters()) + list(self.fc_action.parameters()) + self.phi_params
        self.actor_params.append(self.std)
        self.critic_params = list(self.critic_body.parameters()) + list(self.fc_critic.parameters()) + self.phi_params
        self.to(Config.DEVICE)

    def forward(self, obs, action=None):
        obs = tensor(obs)
        phi = self.phi_body(obs)
        phi_a = self.actor_body(phi)
        phi_v = self.critic_body(phi)
        mean = torch.tanh(self.fc_action(phi_a))
        v = self.fc_critic(phi_v)
        scale = F.softplus(self.std)
        dist = torch.distributions.Normal(mean, scale)
        if action is None:
            # dist.sample() is torch.normal(mean, scale): standard normals, times scale, plus mean.  Written out
            # because torch.normal checks `scale >= 0` on the HOST, which a captured rollout graph cannot do.
            with torch.no_grad():
                sampler = getattr(self, "sampler", None)
                if sampler is not None:     # counter-hash normals (ppo_mlp.gauss_sample): the stream the device rollout draws
                    action = sampler(mean.detach(), scale.detach())
                else:
                    action = torch.randn_like(mean).mul_(scale).add_(mean)
        log_prob = dist.log_prob(action).sum(-1).unsqueeze(-1)
        entropy = dist.entropy().sum(-1).unsqueeze(-1)
        return {'action': action, 'log_pi_a': log_prob, 'entropy': entropy, 'mean': mean, 'v': v}


class CategoricalActorCriticNet(nn.Module, BaseNet):
    """network_heads.py:217-255 (A2C / PPO Atari)."""

    def __init__(self, state_dim, action_dim, phi_body=None, actor_body=None, critic_body=None):
        super(CategoricalActorCriticNet, self).__init__()
        self.phi_body, self.actor_body, self.critic_body = _ac_bodies(state_dim, phi_body, actor_body, critic_body)
        self.fc_action = layer_init(Linear(self.actor_body.feature_dim, action_dim), 1e-3)
        self.fc_critic = layer_init(Linear(self.critic_body.feature_dim, 1), 1e-3)
        self.actor_params = list(self.actor_body.parameters()) + list(self.fc_action.parameters())
        self.critic_params = list(self.critic_body.parameters()) + list(self.fc_critic.parameters())
        self.phi_params = list(self.phi_body.parameters())
        self.rollout_slots = RolloutSlots()
        self.to(Config.DEVICE)

    def _fc4_head_fused(self, obs, action, phi_pre=None):
        """The update's forward over NatureConvBody with fc4 and the policy head as one autograd node (_Fc4PolicyHeadFn), or None
        when the network / batch is not that case."""
        body = self.phi_body
        if not (isinstance(action, torch.Tensor) and action.is_cuda and action.dim() == 1 and obs.is_cuda and obs.dim() == 4
                and 32 < obs.shape[0] <= 4096 and action.shape[0] == obs.shape[0] and getattr(self, 'fuse_fc4_head', True)
                and self._nature_heads_ok()):
            return None
        y = body.conv3(body.conv2(body.conv1(obs)))
        if not (y.dtype == torch.float32 and y.is_contiguous()):
            return None
        if phi_pre is not None and (not torch.is_grad_enabled() or phi_pre.shape[0] != y.shape[0]):
            phi_pre = None
        lp, ent, v = _Fc4PolicyHeadFn.apply(y.view(y.size(0), -1), body.fc4.weight, body.fc4.bias, self.fc_action.weight,
                                            self.fc_action.bias, self.fc_critic.weight, self.fc_critic.bias, action.long().contiguous(),
                                            phi_pre)
        return {'action': action, 'log_pi_a': lp.unsqueeze(-1), 'entropy': ent.unsqueeze(-1), 'v': v.unsqueeze(-1)}

    def _nature_heads_ok(self):
        """phi_body is NatureConvBody (fc4 3136 -> 512 with the fused ReLU), the actor / critic bodies are identities and the two
        heads are plain Linear layers (<= 64 actions, one value): the shapes the fused head launches are written for."""
        body = self.phi_body
        return (type(body) is NatureConvBody and type(self.actor_body) is DummyBody and type(self.critic_body) is DummyBody
                and type(body.fc4) is Linear and body.fc4.fused_act == "relu" and tuple(body.fc4.weight.shape) == (512, 3136)
                and body.fc4.bias is not None and body.fc4.weight.is_contiguous()
                and type(self.fc_action) is Linear and type(self.fc_critic) is Linear and self.fc_action.bias is not None
                and self.fc_critic.bias is not None and self.fc_action.fused_act is None and self.fc_critic.fused_act is None
                and self.fc_action.weight.shape[0] <= 64 and self.fc_critic.weight.shape[0] == 1)

    def _rollout_fc4_head(self, obs):
        """A rollout step's forward (no gradient, action sampled) over NatureConvBody at <= 32 rows with fc4 through the one-pass
        K-slice kernel and its finish inside the head launch (ops.fc4_policy_heads_sample), or None when this is not that case."""
        if not (obs.is_cuda and obs.dim() == 4 and obs.shape[0] <= 32 and getattr(self, "sampler", None) is None
                and not (torch.is_grad_enabled() and self.fc_action.weight.requires_grad) and getattr(self, 'rollout_fc4_slices', True)
                and self._nature_heads_ok()):
            return None
        body = self.phi_body
        y = body.conv3(body.conv2(body.conv1(obs)))
        if not (y.dtype == torch.float32 and y.is_contiguous()):
            return None
        uniform, rows = self.rollout_slots.take(y.shape[0])
        if uniform is None:
            uniform = torch.rand(y.shape[0], dtype=torch.float32, device=y.device)
        a, lp, ent, v = ops.fc4_policy_heads_sample(y.view(y.size(0), -1).detach(), body.fc4.weight.detach(), body.fc4.bias.detach(),
                                                    self.fc_action.weight.detach(), self.fc_action.bias.detach(),
                                                    self.fc_critic.weight.detach(), self.fc_critic.bias.detach(), uniform, rows)
        return {'action': a, 'log_pi_a': lp.unsqueeze(-1), 'entropy': ent.unsqueeze(-1), 'v': v.unsqueeze(-1)}

    def forward(self, obs, action=None):
        obs = tensor(obs)
        # (_phi_pre: set by A2CAgent for ONE call -- the rollout's fc4 output for this batch; only the fused fc4 + head node takes it)
        phi_pre = self.__dict__.pop("_phi_pre", None)
        if action is not None:
            out = self._fc4_head_fused(obs, action, phi_pre)
            if out is not None:
                return out
        else:
            out = self._rollout_fc4_head(obs)
            if out is not None:
                return out
        phi = self.phi_body(obs)
        phi_a = self.actor_body(phi)
        phi_v = self.critic_body(phi)
        sampler = getattr(self, "sampler", None)
        # `sampler` (set by a data-parallel agent, dist.py): draws that do not depend on how the environments are spread
        # over ranks; default = one uniform per sample from torch's global generator (one draw per announced rollout,
        # RolloutSlots, else one per forward), inverse CDF inside the fused kernel (the reference's dist.sample() is
        # torch.multinomial on ITS generator: the action stream of a GPU run is not the reference's CPU stream either way)
        uniform = rows = None
        if action is None and sampler is None and phi_a.is_cuda and phi_a.dim() == 2:
            uniform, rows = self.rollout_slots.take(phi_a.shape[0])
        pair = (phi_a is phi_v and phi_a.is_cuda and phi_a.dim() == 2 and phi_a.shape[1] <= 512 and phi_a.shape[0] <= 65536
                and phi_a.dtype == torch.float32
                and type(self.fc_action) is Linear and type(self.fc_critic) is Linear and self.fc_action.bias is not None
                and self.fc_critic.bias is not None and self.fc_action.fused_act is None and self.fc_critic.fused_act is None)
        with_grad = torch.is_grad_enabled() and (phi_a.requires_grad or self.fc_action.weight.requires_grad)
        if (pair and action is None and sampler is None and not with_grad and self.fc_action.weight.shape[0] <= 64
                and self.fc_critic.weight.shape[0] == 1):
            # a rollout step: both heads, the sample, its log-probability and the entropy in ONE launch
            if uniform is None:
                uniform = torch.rand(phi_a.shape[0], dtype=torch.float32, device=phi_a.device)
            a, lp, ent, v = ops.policy_heads_sample(phi_a.detach(), self.fc_action.weight.detach(), self.fc_action.bias.detach(),
                                                    self.fc_critic.weight.detach(), self.fc_critic.bias.detach(), uniform, rows)
            return {'action': a, 'log_pi_a': lp.unsqueeze(-1), 'entropy': ent.unsqueeze(-1), 'v': v.unsqueeze(-1)}
        if (pair and isinstance(action, torch.Tensor) and action.is_cuda and self.fc_action.weight.shape[0] <= 64
                and self.fc_critic.weight.shape[0] == 1 and phi_a.shape[0] <= ops.HEADS_BWD_MAX_BATCH and action.dim() == 1
                and action.shape[0] == phi_a.shape[0]):
            # the update's forward: both heads + log-probability / entropy of the stored actions in one launch each way
            x_relu = bool(getattr(phi_a, 'dra_fused_relu', False))      # (the body ends in a Linear with a fused ReLU)
            lp, ent, v = _PolicyHeadFn.apply(phi_a.contiguous(), self.fc_action.weight, self.fc_action.bias, self.fc_critic.weight,
                                             self.fc_critic.bias, action.long().contiguous(), x_relu)
            return {'action': action, 'log_pi_a': lp.unsqueeze(-1), 'entropy': ent.unsqueeze(-1), 'v': v.unsqueeze(-1)}
        if pair:
            # both heads read the same features: one launch, the same per-output arithmetic (rollout steps and updates alike)
            if with_grad:
                logits, v = _LinearPairFn.apply(phi_a.contiguous(), self.fc_action.weight, self.fc_action.bias,
                                                self.fc_critic.weight, self.fc_critic.bias)
            else:
                logits, v = ops.linear_fwd_pair(phi_a, self.fc_action.weight, self.fc_action.bias, self.fc_critic.weight,
                                                self.fc_critic.bias)
        else:
            logits = self.fc_action(phi_a)
            v = self.fc_critic(phi_v)
        if logits.is_cuda and logits.shape[-1] <= 64:
            action, log_prob, entropy = categorical_policy(logits, action, sampler, uniform)
            return {'action': action, 'log_pi_a': log_prob, 'entropy': entropy, 'v': v}
        dist = torch.distributions.Categorical(logits=logits)
        if action is None:
            action = sampler(logits) if sampler is not None else dist.sample()
        log_prob = dist.log_prob(action).unsqueeze(-1)
        entropy = dist.entropy().unsqueeze(-1)
        return {'action': action, 'log_pi_a': log_prob, 'entropy': entropy, 'v': v}


class TD3Net(nn.Module, BaseNet):
    """TD3's actor and twin critics, each on its own body (module names of network_heads.py:258-293):
    forward(obs) = tanh(fc_action(actor_body(obs))), q(obs, a) = (Q1, Q2) on [obs, a]."""

    def __init__(self, action_dim, actor_body_fn, critic_body_fn, actor_opt_fn, critic_opt_fn):
        super(TD3Net, self).__init__()
        self.actor_body, self.critic_body_1, self.critic_body_2 = actor_body_fn(), critic_body_fn(), critic_body_fn()
        self.fc_action = _head(self.actor_body.feature_dim, action_dim, 1e-3)
        self.fc_critic_1 = _head(self.critic_body_1.feature_dim, 1, 1e-3)
        self.fc_critic_2 = _head(self.critic_body_2.feature_dim, 1, 1e-3)
        own = lambda *mods: [p for m in mods for p in m.parameters()]
        self.actor_params = own(self.actor_body, self.fc_action)
        self.critic_params = own(self.critic_body_1, self.fc_critic_1) + own(self.critic_body_2, self.fc_critic_2)
        self.actor_opt, self.critic_opt = actor_opt_fn(self.actor_params), critic_opt_fn(self.critic_params)
        self.to(Config.DEVICE)

    def forward(self, obs):
        return torch.tanh(self.fc_action(self.actor_body(tensor(obs))))

    def q(self, obs, a):
        x = torch.cat([tensor(obs), tensor(a)], dim=1)
        return self.fc_critic_1(self.critic_body_1(x)), self.fc_critic_2(self.critic_body_2(x))
