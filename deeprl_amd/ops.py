"""Thin torch-tensor wrappers over the C-ABI (one function per exported kernel group).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every bit of
arithmetic on the hot path is a hand-written HIP kernel in deeprl_amd/csrc reached through
deeprl_amd._lib.  All tensors must be CUDA(ROCm) tensors; there is no CPU path.
"""
import ctypes

import numpy as np
import torch

from ._lib import DraError, lib, ptr, ptr_array, stream_ptr

ACT = {None: 0, "none": 0, "relu": 1, "tanh": 2}
_f32 = torch.float32


def _dev(t):
    if not t.is_cuda:
        raise DraError("deeprl_amd ops need device tensors (got %s); there is no CPU fallback" % t.device)
    return t


def _c(t, dtype=None):
    _dev(t)
    if dtype is not None and t.dtype != dtype:
        raise DraError("expected dtype %s, got %s" % (dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------ ring
class Ring:
    """HBM ring handle (dra_ring_*).  Host bookkeeping (pos/size) lives in component/replay.py."""

    def __init__(self, capacity, frame_bytes, action_bytes, history, n_step, discount):
        h = ctypes.c_void_p()
        lib.dra_ring_create(ctypes.byref(h), int(capacity), int(frame_bytes), int(action_bytes), int(history),
                            int(n_step), float(discount))
        self.h = h
        self.capacity, self.frame_bytes, self.action_bytes = int(capacity), int(frame_bytes), int(action_bytes)
        self.history, self.n_step = int(history), int(n_step)

    def arrays(self):
        """(frames u8[capacity * frame_bytes], actions u8[capacity * action_bytes], rewards f64[capacity], masks i32[capacity])
        as tensors over the ring's own device memory (true resume dumps / restores them)."""
        f, a, r, m = self.pointers()
        return (_wrap_device_pointer(f, self.capacity * self.frame_bytes, torch.uint8),
                _wrap_device_pointer(a, self.capacity * self.action_bytes, torch.uint8),
                _wrap_device_pointer(r, self.capacity, torch.float64), _wrap_device_pointer(m, self.capacity, torch.int32))

    def close(self):
        if self.h:
            lib.dra_ring_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def put_host(self, slot, frame, action, reward, mask):
        frame = np.ascontiguousarray(frame)
        action = np.ascontiguousarray(action)
        if frame.nbytes != self.frame_bytes or action.nbytes != self.action_bytes:
            raise DraError("put_host: frame/action byte size mismatch (%d/%d vs %d/%d)" %
                           (frame.nbytes, action.nbytes, self.frame_bytes, self.action_bytes))
        lib.dra_ring_put_host(self.h, int(slot), frame.ctypes.data_as(ctypes.c_void_p),
                              action.ctypes.data_as(ctypes.c_void_p), float(reward), int(mask), stream_ptr())

    def put_device(self, slot0, frames, actions=None, rewards=None, masks=None, action_val=0, reward_val=0.0,
                   mask_val=1, count=1):
        lib.dra_ring_put(self.h, int(slot0), int(count), ptr(_c(frames)), ptr(actions), int(action_val), ptr(rewards),
                         float(reward_val), ptr(masks), int(mask_val), stream_ptr())

    def fill_synthetic(self, slot0, count, counter0, seed, n_actions=4, done_period=800):
        lib.dra_ring_fill_synthetic(self.h, int(slot0), int(count), int(counter0), int(seed), int(n_actions),
                                    int(done_period), stream_ptr())

    def gather(self, idx, state_shape, state_dtype, action_dtype=torch.int64, want_f32=False, out=None, block=True):
        """idx: int64 device tensor [B].  Returns dict of device tensors shaped like the reference's sample().  block (and no
        caller-owned `out`): state / next_state are two VIEWS of one [B, history + n_step, ...] block in which every frame of a
        sample's run is written once (dra_ring_gather_block) -- same values, non-contiguous along the batch axis."""
        idx = _c(idx, torch.int64)
        b = idx.numel()
        dev = idx.device
        if out is not None and out.get("block") is not None:        # a dict this method returned in block form: refill in place
            lib.dra_ring_gather_block(self.h, ptr(idx), b, ptr(out["block"]), ptr(out["action"]), ptr(out["reward"]), ptr(out["mask"]),
                                      ptr(out.get("reward_f32")), ptr(out.get("mask_f32")), stream_ptr())
            return out
        if out is None and block:
            h, n = self.history, self.n_step
            blk = torch.empty((b, h + n) + tuple(state_shape), dtype=state_dtype, device=dev)
            esize = torch.empty(0, dtype=action_dtype).element_size()
            ashape = (b,) if self.action_bytes == esize else (b, self.action_bytes // esize)
            out = dict(block=blk, state=blk[:, :h] if h > 1 else blk[:, 0], next_state=blk[:, n:] if h > 1 else blk[:, n],
                       action=torch.empty(ashape, dtype=action_dtype, device=dev),
                       reward=torch.empty(b, dtype=torch.float64, device=dev), mask=torch.empty(b, dtype=torch.int32, device=dev))
            if want_f32:
                out["reward_f32"] = torch.empty(b, dtype=_f32, device=dev)
                out["mask_f32"] = torch.empty(b, dtype=_f32, device=dev)
            lib.dra_ring_gather_block(self.h, ptr(idx), b, ptr(blk), ptr(out["action"]), ptr(out["reward"]), ptr(out["mask"]),
                                      ptr(out.get("reward_f32")), ptr(out.get("mask_f32")), stream_ptr())
            return out
        if out is None:
            stack = (b, self.history) + tuple(state_shape) if self.history > 1 else (b,) + tuple(state_shape)
            esize = torch.empty(0, dtype=action_dtype).element_size()
            ashape = (b,) if self.action_bytes == esize else (b, self.action_bytes // esize)
            out = dict(state=torch.empty(stack, dtype=state_dtype, device=dev),
                       next_state=torch.empty(stack, dtype=state_dtype, device=dev),
                       action=torch.empty(ashape, dtype=action_dtype, device=dev),
                       reward=torch.empty(b, dtype=torch.float64, device=dev),
                       mask=torch.empty(b, dtype=torch.int32, device=dev))
            if want_f32:
                out["reward_f32"] = torch.empty(b, dtype=_f32, device=dev)
                out["mask_f32"] = torch.empty(b, dtype=_f32, device=dev)
        lib.dra_ring_gather(self.h, ptr(idx), b, ptr(out["state"]), ptr(out["next_state"]), ptr(out["action"]),
                            ptr(out["reward"]), ptr(out["mask"]), ptr(out.get("reward_f32")), ptr(out.get("mask_f32")),
                            stream_ptr())
        return out

    def pointers(self):
        ps = [ctypes.c_void_p() for _ in range(4)]
        lib.dra_ring_pointers(self.h, *[ctypes.byref(p) for p in ps])
        return [p.value for p in ps]


def u8_to_f32(x_u8, lut):
    _dev(x_u8)
    if x_u8.dtype != torch.uint8:
        raise DraError("expected dtype %s, got %s" % (torch.uint8, x_u8.dtype))
    out = torch.empty(x_u8.shape, dtype=_f32, device=x_u8.device)
    if (not x_u8.is_contiguous() and x_u8.dim() >= 2 and x_u8.shape[0] > 0 and x_u8[0].is_contiguous()
            and x_u8[0].numel() % 16 == 0 and x_u8.stride(0) % 16 == 0 and x_u8.stride(0) >= x_u8[0].numel()
            and x_u8.data_ptr() % 16 == 0):
        # rows that are contiguous inside and a fixed distance apart (the state / next_state views of Ring.gather): read in place
        lib.dra_u8_to_f32_lut_rows(ptr(x_u8), ptr(out), int(x_u8.shape[0]), int(x_u8[0].numel()), int(x_u8.stride(0)),
                                   ptr(_c(lut, _f32)), stream_ptr())
        return out
    x = x_u8 if x_u8.is_contiguous() else x_u8.contiguous()
    lib.dra_u8_to_f32_lut(ptr(x), ptr(out), x.numel(), ptr(_c(lut, _f32)), stream_ptr())
    return out


# ------------------------------------------------------------------------------------------ sum tree
class SumTree:
    def __init__(self, capacity):
        h = ctypes.c_void_p()
        lib.dra_sumtree_create(ctypes.byref(h), int(capacity))
        self.h = h
        self.capacity = int(capacity)
        self.n_nodes = 2 * self.capacity - 1

    def close(self):
        if self.h:
            lib.dra_sumtree_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, leaf_idx, prio, ordered=False):
        leaf_idx, prio = _c(leaf_idx, torch.int64), _c(prio, torch.float64)
        lib.dra_sumtree_update(self.h, ptr(leaf_idx), ptr(prio), leaf_idx.numel(), int(bool(ordered)), stream_ptr())

    def set(self, leaf_idx, prio):
        lib.dra_sumtree_set(self.h, int(leaf_idx), float(prio), stream_ptr())

    def set_from(self, leaf_idx, prio_dev):
        """set() with the priority read from device memory (f64 tensor, first element)."""
        lib.dra_sumtree_set_from(self.h, int(leaf_idx), ptr(prio_dev), stream_ptr())

    def set_many_from(self, write0, n, prio_dev, stream=None):
        """n consecutive adds at write cursor write0.. (mod capacity), all at the device-resident priority prio_dev[0]."""
        lib.dra_sumtree_set_many_from(self.h, int(write0), int(n), ptr(prio_dev), stream_ptr(stream))

    def sample_into(self, u, out_idx, out_p, out_total, stream=None):
        """sample() into caller-provided buffers (e.g. pinned host memory the kernel writes directly; `u` may be pinned too)."""
        lib.dra_sumtree_sample(self.h, ptr(u), u.numel(), ptr(out_idx), ptr(out_p), ptr(out_total), stream_ptr(stream))

    def commit_f32(self, leaf_idx, pos, prio_f32, stat, force_ordered=False, stream=None):
        """Priority write-back with the values still on the device: leaf_idx[i] <- f64(prio_f32[pos[i]]); stat (f64[2] device
        tensor) = {running max, running min} over ALL of prio_f32 (dra_sumtree_commit_f32)."""
        n = 0 if leaf_idx is None else leaf_idx.numel()
        lib.dra_sumtree_commit_f32(self.h, ptr(leaf_idx) if n else None, ptr(pos) if n else None, n, ptr(prio_f32),
                                   prio_f32.numel(), ptr(stat), int(bool(force_ordered)), stream_ptr(stream))

    def sample(self, u):
        u = _c(u, torch.float64)
        b = u.numel()
        idx = torch.empty(b, dtype=torch.int64, device=u.device)
        p = torch.empty(b, dtype=torch.float64, device=u.device)
        total = torch.empty(1, dtype=torch.float64, device=u.device)
        lib.dra_sumtree_sample(self.h, ptr(u), b, ptr(idx), ptr(p), ptr(total), stream_ptr())
        return idx, p, total

    def rebuild(self):
        lib.dra_sumtree_rebuild(self.h, stream_ptr())

    def as_tensor(self):
        """Zero-copy f64 view of the heap array (for tests / checkpoints)."""
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        lib.dra_sumtree_pointer(self.h, ctypes.byref(p), ctypes.byref(n))
        return _wrap_device_pointer(p.value, n.value, torch.float64)


def _wrap_device_pointer(addr, numel, dtype):
    """torch view over library-owned HBM via __cuda_array_interface__ (no copy)."""
    typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.uint8: "|u1", torch.int64: "<i8",
               torch.int32: "<i4"}[dtype]

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = dict(shape=(int(numel),), typestr=typestr, data=(int(addr), False), version=2)
    return torch.as_tensor(h, device="cuda")


# ------------------------------------------------------------------------------------------ losses
def _action_args(action):
    if action.dtype == torch.int64:
        return ptr(_c(action)), 1
    if action.dtype == _f32:
        return ptr(_c(action)), 0
    raise DraError("action must be int64 or float32, got %s" % action.dtype)


def td_loss(q, q_next_target, action, reward, mask, gamma_n, q_next_online=None, sampling_prob=None, beta=0.0,
            replay_eps=0.01, replay_alpha=0.5):
    q, qt = _c(q, _f32), _c(q_next_target, _f32)
    b, a = q.shape
    dev = q.device
    out = dict(loss=torch.empty((), dtype=_f32, device=dev), dq=torch.empty_like(q),
               delta=torch.empty(b, dtype=_f32, device=dev))
    if sampling_prob is not None:
        out["prio"] = torch.empty(b, dtype=_f32, device=dev)
        out["weights"] = torch.empty(b, dtype=_f32, device=dev)
    ap, ai = _action_args(action)
    lib.dra_td_loss(ptr(q), ptr(qt), ptr(None if q_next_online is None else _c(q_next_online, _f32)), ap, ai,
                    ptr(_c(reward, _f32)), ptr(_c(mask, _f32)), b, a, float(gamma_n),
                    ptr(None if sampling_prob is None else _c(sampling_prob, _f32)), float(beta), float(replay_eps),
                    float(replay_alpha), ptr(out["loss"]), ptr(out["dq"]), ptr(out["delta"]), ptr(out.get("prio")),
                    ptr(out.get("weights")), stream_ptr())
    return out


def c51_loss(logits, logits_next_target, action, reward, mask, gamma_n, atoms, v_min, v_max,
             logits_next_online=None, weights=None):
    lg, lt = _c(logits, _f32), _c(logits_next_target, _f32)
    b, a, n = lg.shape
    out = dict(kl=torch.empty(b, dtype=_f32, device=lg.device), dlogits=torch.empty_like(lg))
    ap, ai = _action_args(action)
    lib.dra_c51_loss(ptr(lg), ptr(lt), ptr(None if logits_next_online is None else _c(logits_next_online, _f32)), ap,
                     ai, ptr(_c(reward, _f32)), ptr(_c(mask, _f32)), b, a, n, float(gamma_n), float(v_min),
                     float(v_max), ptr(_c(atoms, _f32)), ptr(out["kl"]), ptr(out["dlogits"]),
                     ptr(None if weights is None else _c(weights, _f32)), stream_ptr())
    out["loss"] = weighted_mean(out["kl"], weights)
    return out


def qr_loss(theta, theta_next_target, action, reward, mask, gamma_n):
    th, tt = _c(theta, _f32), _c(theta_next_target, _f32)
    b, a, n = th.shape
    dev = th.device
    ws = torch.empty(b * n, dtype=_f32, device=dev)
    out = dict(loss_vec=torch.empty(n, dtype=_f32, device=dev), loss=torch.empty((), dtype=_f32, device=dev),
               dtheta=torch.empty_like(th))
    ap, ai = _action_args(action)
    lib.dra_qr_loss(ptr(th), ptr(tt), ap, ai, ptr(_c(reward, _f32)), ptr(_c(mask, _f32)), b, a, n, float(gamma_n),
                    ptr(ws), ptr(out["loss_vec"]), ptr(out["loss"]), ptr(out["dtheta"]), stream_ptr())
    return out


def per_weights(loss_vec, sampling_prob, beta, replay_eps, replay_alpha):
    sp = _c(sampling_prob, _f32)
    b = sp.numel()
    prio = torch.empty(b, dtype=_f32, device=sp.device) if loss_vec is not None else None
    w = torch.empty(b, dtype=_f32, device=sp.device)
    lib.dra_per_weights(ptr(None if loss_vec is None else _c(loss_vec, _f32)), ptr(sp), b, float(beta),
                        float(replay_eps), float(replay_alpha), ptr(prio), ptr(w), stream_ptr())
    return prio, w


def weighted_mean(x, w=None):
    x = _c(x, _f32)
    out = torch.empty((), dtype=_f32, device=x.device)
    lib.dra_weighted_mean(ptr(x), ptr(None if w is None else _c(w, _f32)), x.numel(), ptr(out), stream_ptr())
    return out


def ppo_loss(log_pi_a, entropy, v, old_log_pi_a, adv, ret, ratio_clip, entropy_weight):
    lp = _c(log_pi_a, _f32)
    m = lp.numel()
    out3 = torch.empty(3, dtype=_f32, device=lp.device)
    g = [torch.empty(lp.shape, dtype=_f32, device=lp.device) for _ in range(3)]
    lib.dra_ppo_loss(ptr(lp), ptr(_c(entropy, _f32)), ptr(_c(v, _f32)), ptr(_c(old_log_pi_a, _f32)), ptr(_c(adv, _f32)),
                     ptr(_c(ret, _f32)), m, float(ratio_clip), float(entropy_weight), ptr(out3), ptr(g[0]), ptr(g[1]),
                     ptr(g[2]), stream_ptr())
    return out3, g


def a2c_loss(log_pi_a, entropy, v, adv, ret, entropy_weight, value_loss_weight):
    lp = _c(log_pi_a, _f32)
    m = lp.numel()
    out4 = torch.empty(4, dtype=_f32, device=lp.device)
    g = [torch.empty(lp.shape, dtype=_f32, device=lp.device) for _ in range(3)]
    lib.dra_a2c_loss(ptr(lp), ptr(_c(entropy, _f32)), ptr(_c(v, _f32)), ptr(_c(adv, _f32)), ptr(_c(ret, _f32)), m,
                     float(entropy_weight), float(value_loss_weight), ptr(out4), ptr(g[0]), ptr(g[1]), ptr(g[2]),
                     stream_ptr())
    return out4, g


# ------------------------------------------------------------------------------------------ scan
def gae(reward, mask, value, gamma, tau, use_gae):
    """reward, mask: [T,N(,1)] f32; value: [T+1,N(,1)] f32 -> (adv, ret) shaped like reward."""
    r, m, v = _c(reward, _f32), _c(mask, _f32), _c(value, _f32)
    t_len = r.shape[0]
    n_env = r.numel() // t_len
    if v.numel() != (t_len + 1) * n_env or m.numel() != r.numel():
        raise DraError("gae: shape mismatch")
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    lib.dra_gae(ptr(r), ptr(m), ptr(v), t_len, n_env, float(gamma), float(tau), int(bool(use_gae)), ptr(adv), ptr(ret),
                stream_ptr())
    return adv, ret


def adv_normalize_(adv):
    a = _dev(adv)
    if not a.is_contiguous() or a.dtype != _f32:
        raise DraError("adv_normalize_ needs a contiguous f32 tensor")
    lib.dra_adv_normalize(ptr(a), a.numel(), stream_ptr())
    return adv


# ------------------------------------------------------------------------------------------ contractions
_CONV_GEOM = {1: (4, 84, 32, 8, 4), 2: (32, 20, 64, 4, 2), 3: (64, 9, 64, 3, 1)}  # C, H, OC, K, S


def conv_layer_for(weight_shape, stride, in_hw):
    """Which NatureConvBody layer (1..3) a Conv2d is, or None."""
    for layer, (c, h, oc, k, s) in _CONV_GEOM.items():
        if tuple(weight_shape) == (oc, c, k, k) and stride == s and in_hw == h:
            return layer
    return None


def conv_out_shape(layer, batch):
    c, h, oc, k, s = _CONV_GEOM[layer]
    o = (h - k) // s + 1
    return (batch, oc, o, o)


def conv_fwd(layer, xs, ws, bs, act="relu", u8_coef=None):
    """Batched forward: lists of nz inputs / weights / biases -> list of nz outputs (one launch)."""
    nz = len(xs)
    c, h, oc, k, s = _CONV_GEOM[layer]
    batch = xs[0].shape[0]
    is_u8 = xs[0].dtype == torch.uint8
    if is_u8 and u8_coef is None:
        raise DraError("uint8 input needs u8_coef")
    xs = [_c(x, torch.uint8 if is_u8 else _f32) for x in xs]
    for x in xs:
        if tuple(x.shape) != (batch, c, h, h):
            raise DraError("conv_fwd layer %d expects [B,%d,%d,%d], got %s" % (layer, c, h, h, tuple(x.shape)))
    ws = [_c(w, _f32) for w in ws]
    bs = [_c(b, _f32) for b in bs]
    ys = [torch.empty(conv_out_shape(layer, batch), dtype=_f32, device=xs[0].device) for _ in range(nz)]
    lib.dra_conv_fwd(layer, nz, ptr_array(xs), ptr_array(ws), ptr_array(bs), ptr_array(ys), batch, int(is_u8),
                     float(u8_coef if is_u8 else 1.0), ACT[act], stream_ptr())
    return ys


def conv_bwd_w(layer, dy, x, ksplit=16, u8_coef=None, slabs=None):
    """Returns (dw_slabs [ksplit, OC*K], db_slabs [ksplit, OC]) views of one slab buffer."""
    c, h, oc, k, s = _CONV_GEOM[layer]
    kk = c * k * k
    batch = x.shape[0]
    is_u8 = x.dtype == torch.uint8
    stride = oc * kk + oc
    stride = (stride + 3) // 4 * 4
    if slabs is None:
        slabs = torch.empty(ksplit * stride, dtype=_f32, device=x.device)
    dw0 = slabs
    lib.dra_conv_bwd_w(layer, ptr(_c(dy, _f32)), ptr(_c(x)), ptr(slabs), ctypes.c_void_p(slabs.data_ptr() + 4 * oc * kk),
                       stride, ksplit, batch, int(is_u8), float(u8_coef if is_u8 else 1.0), stream_ptr())
    v = dw0.view(ksplit, stride)
    return v[:, :oc * kk], v[:, oc * kk:oc * kk + oc]


def to_koc(w):
    """[OC,C,KH,KW] -> the [K=(c,kh,kw)][OC] layout of conv_v2.hip (a torch copy; test / glue helper)."""
    return w.permute(1, 2, 3, 0).contiguous().view(-1, w.shape[0])


def from_koc(wt, shape):
    oc, c, kh, kw = shape
    return wt.view(c, kh, kw, oc).permute(3, 0, 1, 2).contiguous()


def conv_fwd_koc(layer, xs, wts, bs, act="relu", u8_coef=None):
    nz = len(xs)
    c, h, oc, k, s = _CONV_GEOM[layer]
    batch = xs[0].shape[0]
    is_u8 = xs[0].dtype == torch.uint8
    xs = [_c(x, torch.uint8 if is_u8 else _f32) for x in xs]
    wts = [_c(w, _f32) for w in wts]
    bs = [_c(b, _f32) for b in bs]
    ys = [torch.empty(conv_out_shape(layer, batch), dtype=_f32, device=xs[0].device) for _ in range(nz)]
    lib.dra_conv_fwd_koc(layer, nz, ptr_array(xs), ptr_array(wts), ptr_array(bs), ptr_array(ys), batch, int(is_u8),
                         float(u8_coef if is_u8 else 1.0), ACT[act], stream_ptr())
    return ys


def conv_bwd_w_koc(layer, dy, x, ksplit=16, u8_coef=None):
    """Returns (dwt_slabs [ksplit, K*OC] in KOC layout, db_slabs [ksplit, OC])."""
    c, h, oc, k, s = _CONV_GEOM[layer]
    kk = c * k * k
    is_u8 = x.dtype == torch.uint8
    stride = (oc * kk + oc + 3) // 4 * 4
    slabs = torch.empty(ksplit * stride, dtype=_f32, device=x.device)
    lib.dra_conv_bwd_w_koc(layer, ptr(_c(dy, _f32)), ptr(_c(x)), ptr(slabs), ctypes.c_void_p(slabs.data_ptr() + 4 * oc * kk),
                           stride, ksplit, x.shape[0], int(is_u8), float(u8_coef if is_u8 else 1.0), stream_ptr())
    v = slabs.view(ksplit, stride)
    return v[:, :oc * kk], v[:, oc * kk:oc * kk + oc]


def conv_bwd_x_koc(layer, dy, wt, xact=None, act="relu"):
    c, h, oc, k, s = _CONV_GEOM[layer]
    batch = dy.shape[0]
    dx = torch.empty((batch, c, h, h), dtype=_f32, device=dy.device)
    lib.dra_conv_bwd_x_koc(layer, ptr(_c(dy, _f32)), ptr(_c(wt, _f32)), ptr(None if xact is None else _c(xact, _f32)),
                           ptr(dx), batch, ACT[act], stream_ptr())
    return dx


# variant bits of the fused / one-pass kernels (include/deeprl_amd.h DRA_VAR_*)
VAR_FUSED_BWD, VAR_ONESHOT_DGRAD, VAR_ONESHOT_FWD, VAR_ONESHOT_WGRAD = 1, 2, 4, 8
VAR_PINNED_IDX, VAR_ACTOR_V2, VAR_ACTOR_PARAMS, VAR_PIPE_GATHER, VAR_CU_PARTITION, VAR_ACTOR_V3 = 16, 32, 64, 128, 256, 512
VAR_ACTOR_FUSED_HEAD = 1024
VAR_GATHER_IN_GRAPH = 2048
VAR_ACTOR_RING = 4096
VAR_ACTOR_FUSED_CONV1 = 8192
VAR_GATHER_ON_UPDATE = 16384
VAR_RING_DIRECT = 32768
VAR_IDX_PREFETCH = 131072
VAR_DGRAD_SCATTER = 262144    # conv2 / conv3 input gradient at batch >= 256 in scatter (col2im) form
VAR_LATE_FOLD = 524288
VAR_ACTOR_MEGA = 1048576
VAR_DEFER_FC4 = 8388608       # fc4's segment of the optimizer step rides in the next update's forward launches
VAR_ACTOR_PERSIST = 16777216  # the whole agent step of the device actor as one launch ({value, tag} hand-overs)
VAR_FWD_CHAIN = 33554432      # conv1 + conv2 + conv3 of the update's forward as one chained launch
VAR_BWD_CHAIN = 67108864      # conv3 + conv2 + conv1 backward of the update as one chained launch
VAR_FLAG_SYNC = 134217728     # the steady-state pipelined step without events (device counts + pinned words)
VAR_LANE_EAGER = 268435456    # ... and the update as plain launches instead of a graph replay
VAR_TARGET_AHEAD = 536870912  # target(next_states) of update t + 1 issued one call early, under update t (needs the next indices)
VAR_BWD_CHAIN_FC = 1073741824 # fc4's + the head's backward as leading roles of the chained backward launch
VAR_HEAD_CHAIN = 65536        # head launch + fc4's / the head's backward launch as one launch in dependency order
VAR_ALL = 2097151 | 8388608 | 16777216 | 33554432 | 67108864 | 134217728 | 268435456 | 536870912 | 1073741824 | 65536


def set_tuning(mask):
    """Process-wide default variant mask for learners created afterwards (dra_set_tuning)."""
    lib.dra_set_tuning(int(mask))


def get_tuning():
    m = ctypes.c_int(0)
    lib.dra_get_tuning(ctypes.byref(m))
    return m.value


def conv_wgrad_slabs(layer, batch, ksplit, variant):
    n = ctypes.c_int(0)
    lib.dra_conv_wgrad_slabs(int(layer), int(batch), int(ksplit), int(variant), ctypes.byref(n))
    return n.value


def conv_bwd_fused(layer, dy, x, wt=None, xact=None, ksplit=16, u8_coef=None, act="relu", variant=0):
    """One launch: KOC weight / bias gradient slabs (+ the input gradient for layers 2, 3).
    Returns (dwt_slabs [n_slabs, K*OC], db_slabs [n_slabs, OC], dx or None, the flat slab buffer)."""
    c, h, oc, k, s = _CONV_GEOM[layer]
    kk = c * k * k
    batch = x.shape[0]
    is_u8 = x.dtype == torch.uint8
    stride = (oc * kk + oc + 3) // 4 * 4
    n_slabs = conv_wgrad_slabs(layer, batch, ksplit, variant)
    slabs = torch.full((n_slabs * stride,), float("nan"), dtype=_f32, device=x.device)
    dx = torch.full((batch, c, h, h), float("nan"), dtype=_f32, device=x.device) if layer > 1 else None
    lib.dra_conv_bwd_fused(layer, ptr(_c(dy, _f32)), ptr(_c(x)), ptr(None if wt is None else _c(wt, _f32)),
                           ptr(None if xact is None else _c(xact, _f32)), ptr(slabs),
                           ctypes.c_void_p(slabs.data_ptr() + 4 * oc * kk), stride, ksplit, ptr(dx), batch, int(is_u8),
                           float(u8_coef if is_u8 else 1.0), ACT[act], int(variant), stream_ptr())
    v = slabs.view(n_slabs, stride)
    return v[:, :oc * kk], v[:, oc * kk:oc * kk + oc], dx, slabs


def conv_bwd_fused_koc(layer, dy, x, wt, ksplit=16, u8_coef=None, variant=0, xact=None):
    """Production form of conv_bwd_fused (generic autograd path): dy = gradient w.r.t. this layer's pre-activation,
    wt = KOC weights.  One launch -> (dx or None, slab buffer [n_slabs * stride], n_slabs, stride); slab s holds
    dWt [K*OC] then db [OC]; the caller folds them (grad_sqnorm)."""
    c, h, oc, k, s = _CONV_GEOM[layer]
    kk = c * k * k
    batch = x.shape[0]
    is_u8 = x.dtype == torch.uint8
    stride = (oc * kk + oc + 3) // 4 * 4
    n_slabs = conv_wgrad_slabs(layer, batch, ksplit, variant)
    slabs = torch.empty(n_slabs * stride, dtype=_f32, device=x.device)
    if stride != oc * kk + oc:
        slabs.view(n_slabs, stride)[:, oc * kk + oc:].zero_()      # alignment gap is folded too: keep it finite
    dx = torch.empty((batch, c, h, h), dtype=_f32, device=x.device) if layer > 1 else None
    lib.dra_conv_bwd_fused(layer, ptr(_c(dy, _f32)), ptr(_c(x)), ptr(_c(wt, _f32)), ptr(None if xact is None else _c(xact, _f32)),
                           ptr(slabs), ctypes.c_void_p(slabs.data_ptr() + 4 * oc * kk), stride, ksplit, ptr(dx), batch, int(is_u8),
                           float(u8_coef if is_u8 else 1.0), ACT["relu"], int(variant), stream_ptr())
    return dx, slabs, n_slabs, stride


def fc_bwd_fused(dq, h4, dh4, x3, w4, act="relu", variant=0):
    """One launch: head weight gradient, fc4 weight gradient, fc4 input gradient (hidden = 512)."""
    dq, h4, dh4, x3, w4 = [_c(t, _f32) for t in (dq, h4, dh4, x3, w4)]
    batch, a = dq.shape
    fin = x3.shape[1]
    dev = dq.device
    nan = float("nan")
    dwh = torch.full((a, 512), nan, dtype=_f32, device=dev)
    dbh = torch.full((a,), nan, dtype=_f32, device=dev)
    dw4 = torch.full((512, fin), nan, dtype=_f32, device=dev)
    db4 = torch.full((512,), nan, dtype=_f32, device=dev)
    dx3 = torch.full((batch, fin), nan, dtype=_f32, device=dev)
    lib.dra_fc_bwd_fused(ptr(dq), ptr(h4), ptr(dh4), ptr(x3), ptr(w4), ptr(dwh), ptr(dbh), ptr(dw4), ptr(db4), ptr(dx3),
                         batch, a, fin, ACT[act], int(variant), stream_ptr())
    return dwh, dbh, dw4, db4, dx3


def linear_fwd_slabs(xs, ws, ksplit=8, one_pass=False):
    """Raw split-K partial sums [nz][ksplit][B][O] of the fc4-shaped forward."""
    nz = len(xs)
    xs = [_c(x, _f32) for x in xs]
    ws = [_c(w, _f32) for w in ws]
    batch, fin = xs[0].shape
    fout = ws[0].shape[0]
    slabs = torch.full((nz, ksplit, batch, fout), float("nan"), dtype=_f32, device=xs[0].device)
    fn = lib.dra_linear_fwd_slabs_one if one_pass else lib.dra_linear_fwd_slabs
    fn(nz, ptr_array(xs), ptr_array(ws), batch, fin, fout, ksplit, ptr(slabs), stream_ptr())
    return slabs


class FoldSeg(ctypes.Structure):
    """Mirror of dra_fold_seg."""
    _fields_ = [("begin", ctypes.c_int64), ("count", ctypes.c_int64), ("slabs", ctypes.c_void_p),
                ("slab_stride", ctypes.c_int64), ("n_slabs", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def grad_sqnorm_segs(grad, segs, partials):
    """segs: list of (begin, count, slabs_tensor, slab_stride, n_slabs).  Returns the number of partials written."""
    arr = _fold_seg_array(segs)
    n = ctypes.c_int(0)
    lib.dra_grad_sqnorm_segs(ptr(grad), grad.numel(), arr, len(segs), ptr(partials), ctypes.byref(n), stream_ptr())
    return n.value


MAX_FOLD_SEGS = 4      # DRA_MAX_FOLD_SEGS (include/deeprl_amd.h)


def norm_partials_max():
    return lib.dra_norm_partials_max.raw()


def _fold_seg_array(segs):
    arr = (FoldSeg * max(1, len(segs)))()
    for i, (b, c, t, st, ns) in enumerate(segs):
        arr[i].begin, arr[i].count, arr[i].slabs, arr[i].slab_stride, arr[i].n_slabs = b, c, t.data_ptr(), st, ns
    return arr


def conv_bwd_x(layer, dy, w, xact=None, act="relu"):
    """Gradient w.r.t. the layer's input; with `xact` (the layer-below's activated output) the
    activation derivative of that layer is folded in (gradient w.r.t. its PRE-activation)."""
    c, h, oc, k, s = _CONV_GEOM[layer]
    batch = dy.shape[0]
    dx = torch.empty((batch, c, h, h), dtype=_f32, device=dy.device)
    lib.dra_conv_bwd_x(layer, ptr(_c(dy, _f32)), ptr(_c(w, _f32)), ptr(None if xact is None else _c(xact, _f32)), ptr(dx),
                       batch, ACT[act], stream_ptr())
    return dx


def act_bwd(dy, y, act):
    """dpre = dy * act'(y) (activation derivative through its output)."""
    dy, y = _c(dy, _f32), _c(y, _f32)
    out = torch.empty_like(dy)
    lib.dra_act_bwd(ptr(dy), ptr(y), ptr(out), dy.numel(), ACT[act], stream_ptr())
    return out


_ws_cache = {}


def _workspace(device, floats):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() < floats:
        t = torch.empty(max(floats, 1 << 20), dtype=_f32, device=device)
        _ws_cache[key] = t
    return t


def linear_fwd_pair(x, w0, b0, w1, b1, act=None):
    """(x W0^T + b0, x W1^T + b1) in one launch (two heads on the same features; in_features <= 512)."""
    x, w0, w1 = _c(x, _f32), _c(w0, _f32), _c(w1, _f32)
    b0 = None if b0 is None else _c(b0, _f32)
    b1 = None if b1 is None else _c(b1, _f32)
    batch, fin = x.shape
    y0 = torch.empty((batch, w0.shape[0]), dtype=_f32, device=x.device)
    y1 = torch.empty((batch, w1.shape[0]), dtype=_f32, device=x.device)
    lib.dra_linear_fwd_pair(ptr(x), ptr(w0), ptr(b0), ptr(y0), int(w0.shape[0]), ptr(w1), ptr(b1), ptr(y1), int(w1.shape[0]),
                            batch, fin, ACT[act], stream_ptr())
    return y0, y1


def gather_rows(tensors, idx):
    """[t[idx] for t in tensors] (row gathers along dim 0 of contiguous device tensors that share their row count) in ONE launch."""
    idx = _c(idx, torch.int64)
    n = int(idx.numel())
    k = len(tensors)
    if not 1 <= k <= 8:
        raise ValueError("gather_rows: 1..8 tensors")
    rows = int(tensors[0].shape[0])
    srcs, outs = [], []
    for t in tensors:
        t = _c(t)
        if t.dim() < 1 or int(t.shape[0]) != rows:
            raise ValueError("gather_rows: tensors must share dim 0")
        srcs.append(t)
        outs.append(torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device))
    src_arr = (ctypes.c_void_p * k)(*[t.data_ptr() for t in srcs])
    dst_arr = (ctypes.c_void_p * k)(*[t.data_ptr() for t in outs])
    rb = (ctypes.c_int64 * k)(*[(t.numel() // rows) * t.element_size() for t in srcs])
    lib.dra_gather_rows(k, src_arr, dst_arr, rb, ptr(idx), n, rows, stream_ptr())
    return outs


def policy_heads_sample(x, w0, b0, w1, b1, uniform, out=None):
    """A rollout step's policy head in one launch: (action i64 [B], log_pi_a [B], entropy [B], v [B]) of
    Categorical(logits = x W0^T + b0) sampled by inverse CDF from uniform [B], v = x w1^T + b1 -- linear_fwd_pair +
    categorical_fwd, bit for bit.  out: four preallocated tensors of those shapes (rows of a rollout's buffers)."""
    x, w0, w1, uniform = _c(x, _f32), _c(w0, _f32), _c(w1, _f32), _c(uniform, _f32)
    b0 = None if b0 is None else _c(b0, _f32)
    b1 = None if b1 is None else _c(b1, _f32)
    batch, fin = x.shape
    if w1.shape[0] != 1 or uniform.numel() != batch:
        raise ValueError("policy_heads_sample: one value output and one uniform per row")
    if out is None:
        out = (torch.empty(batch, dtype=torch.int64, device=x.device), torch.empty(batch, dtype=_f32, device=x.device),
               torch.empty(batch, dtype=_f32, device=x.device), torch.empty(batch, dtype=_f32, device=x.device))
    a, lp, ent, v = out
    for t_, dt in ((a, torch.int64), (lp, _f32), (ent, _f32), (v, _f32)):
        if t_.dtype != dt or t_.numel() != batch or not t_.is_contiguous():
            raise ValueError("policy_heads_sample: output rows must be contiguous [B] tensors (i64, f32, f32, f32)")
    lib.dra_policy_heads_sample(ptr(x), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(uniform), batch, fin, int(w0.shape[0]), ptr(a),
                                ptr(lp), ptr(ent), ptr(v), None, stream_ptr())
    return a, lp, ent, v


def policy_heads_given(x, w0, b0, w1, b1, action):
    """The update's forward through the policy head in one launch -> (log_pi_a [B], entropy [B], v [B], logits [B, A]) for the
    given actions (linear_fwd_pair + categorical_fwd(action=...), bit for bit)."""
    x, w0, w1, action = _c(x, _f32), _c(w0, _f32), _c(w1, _f32), _c(action, torch.int64)
    b0 = None if b0 is None else _c(b0, _f32)
    b1 = None if b1 is None else _c(b1, _f32)
    batch, fin = x.shape
    a = int(w0.shape[0])
    if w1.shape[0] != 1 or action.numel() != batch:
        raise ValueError("policy_heads_given: one value output and one action per row")
    lp, ent, v = (torch.empty(batch, dtype=_f32, device=x.device) for _ in range(3))
    logits = torch.empty((batch, a), dtype=_f32, device=x.device)
    lib.dra_policy_heads_given(ptr(x), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(action), batch, fin, a, ptr(lp), ptr(ent), ptr(v),
                               ptr(logits), stream_ptr())
    return lp, ent, v, logits


def fc4_policy_heads_given(y3, w4, b4, w0, b0, w1, b1, action):
    """fc4 (3136 -> 512, + ReLU) and the policy head for given actions as TWO launches: the one-pass K-slice forward (8 slabs) and
    the head launch that folds them -> (log_pi_a [B], entropy [B], v [B], logits [B, A], phi [B, 512]); linear_fwd + its finish +
    policy_heads_given were three, same arithmetic."""
    y3, w4, b4, w0, w1, action = _c(y3, _f32), _c(w4, _f32), _c(b4, _f32), _c(w0, _f32), _c(w1, _f32), _c(action, torch.int64)
    b0 = None if b0 is None else _c(b0, _f32)
    b1 = None if b1 is None else _c(b1, _f32)
    batch = int(y3.shape[0])
    a = int(w0.shape[0])
    dev = y3.device
    slabs = torch.empty((8, batch, 512), dtype=_f32, device=dev)      # (8 slices: the fastest at 80 / 256 rows, tools/fc4_small_probe.py)
    lib.dra_linear_fwd_slabs_one(1, ptr_array([y3]), ptr_array([w4]), batch, 3136, 512, 8, ptr(slabs), stream_ptr())
    lp, ent, v = (torch.empty(batch, dtype=_f32, device=dev) for _ in range(3))
    logits = torch.empty((batch, a), dtype=_f32, device=dev)
    phi = torch.empty((batch, 512), dtype=_f32, device=dev)
    lib.dra_policy_heads_given_fold(ptr(slabs), 8, ptr(b4), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(action), batch, a, ptr(lp),
                                      ptr(ent), ptr(v), ptr(logits), ptr(phi), stream_ptr())
    return lp, ent, v, logits, phi


def fc4_policy_heads_sample(y3, w4, b4, w0, b0, w1, b1, uniform, out=None):
    """fc4 (3136 -> 512, + ReLU) and the sampling policy head of a rollout step (<= 32 rows) as TWO launches: the one-pass K-slice
    forward with 28 slices and the head launch that folds them -> (action i64 [B], log_pi_a, entropy, v [B]).  (The eight-wave GEMV
    + policy_heads_sample are two as well, 5.6 / 8.1 / 13.2 + 6 us at 8 / 16 / 32 rows against 4.3 + 5: tools/fc4_small_probe.py.)"""
    y3, w4, b4, w0, w1, uniform = _c(y3, _f32), _c(w4, _f32), _c(b4, _f32), _c(w0, _f32), _c(w1, _f32), _c(uniform, _f32)
    b0 = None if b0 is None else _c(b0, _f32)
    b1 = None if b1 is None else _c(b1, _f32)
    batch = int(y3.shape[0])
    dev = y3.device
    slabs = torch.empty((28, batch, 512), dtype=_f32, device=dev)
    lib.dra_linear_fwd_slabs_one(1, ptr_array([y3]), ptr_array([w4]), batch, 3136, 512, 28, ptr(slabs), stream_ptr())
    if out is None:
        out = (torch.empty(batch, dtype=torch.int64, device=dev), torch.empty(batch, dtype=_f32, device=dev),
               torch.empty(batch, dtype=_f32, device=dev), torch.empty(batch, dtype=_f32, device=dev))
    a, lp, ent, v = out
    lib.dra_policy_heads_sample_fold28(ptr(slabs), ptr(b4), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(uniform), batch, int(w0.shape[0]),
                                       ptr(a), ptr(lp), ptr(ent), ptr(v), stream_ptr())
    return a, lp, ent, v


HEADS_BWD_MAX_BATCH = 8192


def policy_heads_bwd(logits, action, g_lp, g_ent, g_v, x, w0, w1, dw0=None, db0=None, dw1=None, db1=None, want_dx=True,
                     relu_mask=False):
    """Backward of policy_heads_given in one launch -> (dx or None, dw0, db0, dw1, db1); g_* [B] or None (= zero); dw / db may
    be given (written in place); relu_mask: x is a fused-ReLU output, dx is the gradient of its PRE-activation."""
    logits, action, x, w0, w1 = _c(logits, _f32), _c(action, torch.int64), _c(x, _f32), _c(w0, _f32), _c(w1, _f32)
    g_lp, g_ent, g_v = [None if g is None else _c(g, _f32) for g in (g_lp, g_ent, g_v)]
    batch, fin = x.shape
    a = int(w0.shape[0])
    dev = x.device
    dw0 = torch.empty((a, fin), dtype=_f32, device=dev) if dw0 is None else dw0
    db0 = torch.empty(a, dtype=_f32, device=dev) if db0 is None else db0
    dw1 = torch.empty((1, fin), dtype=_f32, device=dev) if dw1 is None else dw1
    db1 = torch.empty(1, dtype=_f32, device=dev) if db1 is None else db1
    dx = torch.empty((batch, fin), dtype=_f32, device=dev) if want_dx else None
    lib.dra_policy_heads_bwd(ptr(logits), ptr(action), ptr(g_lp), ptr(g_ent), ptr(g_v), ptr(x), ptr(w0), ptr(w1), ptr(dx), ptr(dw0),
                             ptr(db0), ptr(dw1), ptr(db1), batch, fin, a, 1 if relu_mask else 0, stream_ptr())
    return dx, dw0, db0, dw1, db1


def linear_bwd_pair(g0, g1, x, w0, w1, dw0=None, db0=None, dw1=None, db1=None, want_dx=True):
    """Backward of linear_fwd_pair in one launch -> (dx or None, dw0, db0, dw1, db1); dw / db may be given (written in place)."""
    g0, g1, x, w0, w1 = [_c(t, _f32) for t in (g0, g1, x, w0, w1)]
    batch, fin = x.shape
    o0, o1 = w0.shape[0], w1.shape[0]
    dev = x.device
    dw0 = torch.empty((o0, fin), dtype=_f32, device=dev) if dw0 is None else dw0
    db0 = torch.empty(o0, dtype=_f32, device=dev) if db0 is None else db0
    dw1 = torch.empty((o1, fin), dtype=_f32, device=dev) if dw1 is None else dw1
    db1 = torch.empty(o1, dtype=_f32, device=dev) if db1 is None else db1
    dx = torch.empty((batch, fin), dtype=_f32, device=dev) if want_dx else None
    lib.dra_linear_bwd_pair(ptr(g0), ptr(g1), ptr(x), ptr(w0), ptr(w1), ptr(dx), ptr(dw0), ptr(db0), ptr(dw1), ptr(db1), batch, fin,
                            o0, o1, stream_ptr())
    return dx, dw0, db0, dw1, db1


def linear_fwd(xs, ws, bs, act=None):
    nz = len(xs)
    xs = [_c(x, _f32) for x in xs]
    batch, fin = xs[0].shape
    fout = ws[0].shape[0]
    ws = [_c(w, _f32) for w in ws]
    bs = [None if b is None else _c(b, _f32) for b in bs]
    ys = [torch.empty((batch, fout), dtype=_f32, device=xs[0].device) for _ in range(nz)]
    need = nz * 32 * batch * fout
    wsb = _workspace(xs[0].device, need)
    lib.dra_linear_fwd(nz, ptr_array(xs), ptr_array(ws), ptr_array(bs), ptr_array(ys), batch, fin, fout, ACT[act],
                       ptr(wsb), wsb.numel(), stream_ptr())
    return ys


def linear_bwd_w(dy, x, dw=None, db=None, want_bias=True):
    dy, x = _c(dy, _f32), _c(x, _f32)
    batch, fout = dy.shape
    fin = x.shape[1]
    if dw is None:
        dw = torch.empty((fout, fin), dtype=_f32, device=x.device)
    if db is None and want_bias:
        db = torch.empty(fout, dtype=_f32, device=x.device)
    lib.dra_linear_bwd_w(ptr(dy), ptr(x), ptr(dw), ptr(db), batch, fin, fout, stream_ptr())
    return dw, db


def linear_bwd_xw_512(dy, x, w, x_relu, dw=None, db=None, want_bias=True):
    """Input gradient and weight / bias gradient of a 512-output layer with in_features >= 1024 in ONE launch -> (dx, dw, db):
    linear_bwd_x(dy, w, xact=x if x_relu, act='relu') and linear_bwd_w(dy, x), same arithmetic."""
    dy, x, w = _c(dy, _f32), _c(x, _f32), _c(w, _f32)
    batch, fin = x.shape
    if dw is None:
        dw = torch.empty((512, fin), dtype=_f32, device=x.device)
    if db is None and want_bias:
        db = torch.empty(512, dtype=_f32, device=x.device)
    dx = torch.empty((batch, fin), dtype=_f32, device=x.device)
    lib.dra_linear_bwd_xw_one512(ptr(dy), ptr(w), ptr(x), 1 if x_relu else 0, ptr(dx), ptr(dw), ptr(db), batch, fin, stream_ptr())
    return dx, dw, db


def linear_bwd_x(dy, w, xact=None, act=None, dx=None):
    dy, w = _c(dy, _f32), _c(w, _f32)
    batch, fout = dy.shape
    fin = w.shape[1]
    if dx is None:
        dx = torch.empty((batch, fin), dtype=_f32, device=dy.device)
    lib.dra_linear_bwd_x(ptr(dy), ptr(w), ptr(None if xact is None else _c(xact, _f32)), ptr(dx), batch, fin, fout,
                         ACT[act], stream_ptr())
    return dx


# ------------------------------------------------------------------------------------------ optimiser
def norm_partials():
    return lib.dra_norm_partials.raw()


def grad_sqnorm(grad, partials, slabs=None, n_slabs=0, slab_stride=0):
    lib.dra_grad_sqnorm(ptr(grad), grad.numel(), ptr(slabs), int(n_slabs), int(slab_stride), ptr(partials), stream_ptr())


def rmsprop_step(param, grad, square_avg, grad_avg, partials, n_partials, max_norm, lr, alpha, eps, centered,
                 out_norm=None):
    lib.dra_rmsprop_step(ptr(param), ptr(grad), ptr(square_avg), ptr(grad_avg), param.numel(), ptr(partials),
                         int(n_partials), float(max_norm if max_norm else 0.0), float(lr), float(alpha), float(eps),
                         int(bool(centered)), ptr(out_norm), stream_ptr())


def adam_step(param, grad, exp_avg, exp_avg_sq, partials, n_partials, max_norm, lr, beta1, beta2, eps, step,
              out_norm=None):
    lib.dra_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), ptr(partials),
                      int(n_partials), float(max_norm if max_norm else 0.0), float(lr), float(beta1), float(beta2),
                      float(eps), int(step), ptr(out_norm), stream_ptr())


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, partials, n_partials, max_norm, beta1, beta2, eps, hyper_dev,
                  out_norm=None):
    """Adam with {lr/(1-b1^t), 1/sqrt(1-b2^t)} read from the device tensor `hyper_dev` (graph-replayable)."""
    lib.dra_adam_step_dev(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), ptr(partials),
                          int(n_partials), float(max_norm if max_norm else 0.0), float(beta1), float(beta2), float(eps),
                          ptr(hyper_dev), ptr(out_norm), stream_ptr())


def categorical_fwd(logits, action=None, uniform=None):
    """Categorical(logits): (action, log_pi_a [B], entropy [B]); action None -> sampled by inverse CDF from `uniform` [B]."""
    logits = _c(logits, _f32)
    b, a = logits.shape
    lp = torch.empty(b, dtype=_f32, device=logits.device)
    ent = torch.empty(b, dtype=_f32, device=logits.device)
    if action is None:
        out_a = torch.empty(b, dtype=torch.int64, device=logits.device)
        lib.dra_categorical_fwd(ptr(logits), b, a, None, ptr(_c(uniform, _f32)), ptr(out_a), ptr(lp), ptr(ent), stream_ptr())
        return out_a, lp, ent
    action = _c(action, torch.int64)
    lib.dra_categorical_fwd(ptr(logits), b, a, ptr(action), None, None, ptr(lp), ptr(ent), stream_ptr())
    return action, lp, ent


class PerChain2IO(ctypes.Structure):
    """include/deeprl_amd.h dra_per_chain2_io (pinned host block of one dra_sumtree_per_chain2 launch)."""
    _fields_ = [("add_n", ctypes.c_int32), ("batch", ctypes.c_int32), ("next_batch", ctypes.c_int32), ("force_ordered", ctypes.c_int32),
                ("history", ctypes.c_int32), ("n_step", ctypes.c_int32), ("add_write0", ctypes.c_int64),
                ("memory_size", ctypes.c_int64), ("pos_after", ctypes.c_int64), ("size_after", ctypes.c_int64),
                ("rng_produced", ctypes.c_uint64), ("beta_next", ctypes.c_float), ("reserved", ctypes.c_int32),
                ("out_raw_idx", ctypes.c_int64 * 1024), ("out_idx", ctypes.c_int64 * 1024), ("out_p", ctypes.c_double * 1024),
                ("out_total", ctypes.c_double), ("out_n_valid", ctypes.c_int32), ("out_flags", ctypes.c_int32),
                ("out_rng_cursor", ctypes.c_uint64), ("out_seq", ctypes.c_uint64)]


PER_RNG_WORDS = 65536     # include/deeprl_amd.h DRA_PER_RNG_WORDS


class AtariPreprocess:
    """envs.py:39-47 (baselines' MaxAndSkipEnv max + WarpFrame: RGB2GRAY, INTER_AREA resize to 84x84) as one kernel
    (csrc/preproc.hip): raw [n_env][2][H][W][3] uint8 (the last two frames of each environment's frame skip; host array or
    device tensor) -> device uint8 [n_env][84][84], what FrameStack / the replay ring take.  The two axis tables are built
    once on the host by dra_resize_area_tab (OpenCV's computeResizeAreaTab)."""

    def __init__(self, height=210, width=160, out_h=84, out_w=84, device=None):
        from .support import Config
        self.h, self.w, self.oh, self.ow = int(height), int(width), int(out_h), int(out_w)
        self.device = device or Config.DEVICE
        self.tabs = []
        for ssize, dsize in ((self.w, self.ow), (self.h, self.oh)):
            cap = 2 * ssize + dsize + 8
            si, al, off = (ctypes.c_int * cap)(), (ctypes.c_float * cap)(), (ctypes.c_int * (dsize + 1))()
            n = lib.dra_resize_area_tab.raw(ssize, dsize, si, al, off, cap)
            if n <= 0:
                raise DraError("dra_resize_area_tab(%d, %d) failed: %d" % (ssize, dsize, n))
            self.tabs += [torch.tensor(list(si[:n]), dtype=torch.int32, device=self.device),
                          torch.tensor(list(al[:n]), dtype=torch.float32, device=self.device),
                          torch.tensor(list(off), dtype=torch.int32, device=self.device)]

    def __call__(self, raw2, out=None):
        if not isinstance(raw2, torch.Tensor):
            raw2 = torch.from_numpy(np.ascontiguousarray(raw2, dtype=np.uint8)).to(self.device, non_blocking=True)
        raw2 = _c(raw2, torch.uint8)
        n = raw2.shape[0]
        if tuple(raw2.shape[1:]) != (2, self.h, self.w, 3):
            raise DraError("AtariPreprocess: raw frames must be [n_env, 2, %d, %d, 3] uint8" % (self.h, self.w))
        if out is None:
            out = torch.empty((n, self.oh, self.ow), dtype=torch.uint8, device=raw2.device)
        t = self.tabs
        lib.dra_atari_preprocess(ptr(raw2), n, self.h, self.w, self.oh, self.ow, ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(t[3]), ptr(t[4]),
                                 ptr(t[5]), ptr(out), stream_ptr())
        return out


def gumbel_sample(logits, seed, step_dev, lo):
    """Rank-invariant categorical sample (dra_gumbel_sample): noise hashed from (seed, *step_dev, global row lo + i, action);
    advances the device step counter.  Graph-capturable (no host-side generator)."""
    logits = _c(logits, _f32)
    b, a = logits.shape
    out = torch.empty(b, dtype=torch.int64, device=logits.device)
    lib.dra_gumbel_sample(ptr(logits), b, a, int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(step_dev), int(lo), ptr(out), stream_ptr())
    return out


def categorical_bwd(logits, action, g_lp, g_ent):
    logits = _c(logits, _f32)
    b, a = logits.shape
    out = torch.empty_like(logits)
    lib.dra_categorical_bwd(ptr(logits), b, a, ptr(_c(action, torch.int64)), ptr(None if g_lp is None else _c(g_lp, _f32)),
                            ptr(None if g_ent is None else _c(g_ent, _f32)), ptr(out), stream_ptr())
    return out


def adam_step_counter(param, grad, exp_avg, exp_avg_sq, partials, n_partials, max_norm, lr, beta1, beta2, eps, step_dev,
                      out_norm=None, param_copy=None):
    """Adam whose 1-based step count is the int64 device tensor `step_dev` (bumped by the caller's graph); optional
    mirror of the updated parameters."""
    lib.dra_adam_step_counter(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), ptr(partials),
                              int(n_partials), float(max_norm if max_norm else 0.0), float(lr), float(beta1), float(beta2),
                              float(eps), ptr(step_dev), ptr(out_norm), ptr(param_copy), stream_ptr())


def copy_f32(dst, src):
    lib.dra_copy_f32(ptr(dst), ptr(src), src.numel(), stream_ptr())


def soft_update(target_flat, src_flat, mix):
    """target <- target * (1 - mix) + src * mix over two flat fp32 buffers of equal length (DDPG_agent.py:26-30): one
    launch of `dra_soft_update`; both products are rounded to fp32 before the add, as in the reference's expression."""
    _dev(target_flat), _dev(src_flat)
    if target_flat.numel() != src_flat.numel() or target_flat.dtype != torch.float32 or src_flat.dtype != torch.float32:
        raise DraError("soft_update: two float32 buffers of equal length")
    keep = float(np.float32(1.0 - mix))
    lib.dra_soft_update(ptr(target_flat), ptr(src_flat), target_flat.numel(), keep, float(mix), stream_ptr())
