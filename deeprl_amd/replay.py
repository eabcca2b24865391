"""Host mirror of the reference replay interface over the HBM ring + HIP kernels.

Same names, constructor signatures, method names and error behaviour as
deep_rl/component/replay.py (Storage :20-54, UniformReplay :57-149, PrioritizedReplay
:152-196, ReplayWrapper :199-278); data lives in HBM and `sample()` returns DEVICE tensors
(the reference's `tensor()` / normalisers pass torch tensors through unchanged, torch_utils.py:20-22,
normalizer.py:58-61, so agents need no change).

What stays on the host, on purpose: `pos` / `size` bookkeeping, `valid_index`, and every RNG
draw (numpy legacy global RandomState for uniform sampling, python `random` for prioritized
sampling) -- the index stream is therefore the reference's own, draw for draw.
"""
import contextlib
import random
from collections import namedtuple

import numpy as np
import torch

from . import ops
from ._lib import DraError
from .support import Config

def draw_uniform_indices(size, pos, batch, history, n_step):
    """UniformReplay.sample's rejection loop (replay.py:92-110), vectorised without changing the
    np.random stream: randint(0, size, size=k) yields the same values as k scalar draws, and each
    block asks for exactly the number still missing, which the scalar loop would also draw."""
    out = np.empty(batch, dtype=np.int64)
    have = 0
    while have < batch:
        cand = np.random.randint(0, size, size=batch - have)
        lo = cand - history + 1
        hi = cand + n_step
        ok = ((lo >= 0) & (hi < pos)) | ((lo >= pos) & (hi < size))
        good = cand[ok]
        out[have:have + len(good)] = good
        have += len(good)
    return out


Transition = namedtuple('Transition', ['state', 'action', 'reward', 'next_state', 'mask'])
PrioritizedTransition = namedtuple('Transition',
                                   ['state', 'action', 'reward', 'next_state', 'mask', 'sampling_prob', 'idx'])

_DEFAULT_KEYS = ['state', 'action', 'reward', 'mask', 'v', 'q', 'pi', 'log_pi', 'entropy', 'advantage', 'ret', 'q_a',
                 'log_pi_a', 'mean', 'next_state']

_NP2TORCH = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
             np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32}


class Storage:
    """On-policy rollout buffer: per-key python lists of [N, ...] device tensors, time-major
    `extract` (replay.py:20-54).  Used by the A2C / PPO / n-step agents."""

    def __init__(self, memory_size, keys=None):
        self.keys = list(keys or []) + _DEFAULT_KEYS
        self.memory_size = memory_size
        self.reset()

    def feed(self, data):
        for k, v in data.items():
            if k not in self.keys:
                raise RuntimeError('Undefined key')
            getattr(self, k).append(v)

    def placeholder(self):
        for k in self.keys:
            if len(getattr(self, k)) == 0:
                setattr(self, k, [None] * self.memory_size)

    def reset(self):
        for k in self.keys:
            setattr(self, k, [])
        self.pos = 0
        self._size = 0

    def extract(self, keys):
        entry = namedtuple('Entry', keys)
        return entry(*[torch.cat(getattr(self, k)[:self.memory_size], dim=0) for k in keys])


class _PinnedUploader:
    """Rotating pinned staging buffers for the few hundred bytes of indices / uniforms that cross
    the host->device boundary per sample; an event per slot guards reuse.

    device_copy=False hands the KERNEL the pinned buffer itself (pinned host memory is device-addressable: a
    single-workgroup tree kernel reading 32 words over the host link pays ~1.5 us inside the kernel instead of a copy command
    in front of it and ~15 us of host time for the tensor copy -- cProfile of the prioritized agent step,
    profiles/r02zz1_host_profile_per.txt: the PER pipeline was bound by exactly this host path).  The kernel that reads the
    buffer is enqueued AFTER upload() returns, so the slot's reuse guard is recorded by consumed(stream), which the caller
    invokes right behind that launch; a slot that comes round again without it waits for the whole stream."""

    def __init__(self, dtype, numel, device, slots=8):
        self.bufs = [torch.empty(numel, dtype=dtype).pin_memory() for _ in range(slots)]
        self.views = [b.numpy() for b in self.bufs]
        self.events = [None] * slots
        self.used = [False] * slots
        self.device = device
        self.k = 0
        self.unguarded = []          # slots handed to a kernel that has not been enqueued yet

    def consumed(self, stream=None):
        """The kernel(s) reading the buffers handed out with device_copy=False have been enqueued on `stream`."""
        for k in self.unguarded:
            self.events[k].record(stream)
        self.unguarded = []

    def upload_into(self, dst, array):
        """dst[:len(array)] <- array through the next staging slot: ONE host-to-device copy on the current stream (upload()
        followed by dst.copy_() is two: the staging tensor's own device copy, then a device-to-device one)."""
        k = self.k
        self.k = (k + 1) % len(self.bufs)
        if k in self.unguarded:
            torch.cuda.current_stream().synchronize()
            self.unguarded.remove(k)
        elif self.used[k]:
            self.events[k].synchronize()
        n = len(array)
        self.views[k][:n] = array
        dst[:n].copy_(self.bufs[k][:n], non_blocking=True)
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        self.events[k].record()
        self.used[k] = True

    def upload(self, array, device_copy=True, stream=None):
        k = self.k
        self.k = (k + 1) % len(self.bufs)
        if k in self.unguarded:      # its reader was never reported: nothing finer than the stream to wait for
            (stream or torch.cuda.current_stream()).synchronize()
            self.unguarded.remove(k)
        elif self.used[k]:
            self.events[k].synchronize()
        n = len(array)
        self.views[k][:n] = array                      # (numpy casts to the buffer's dtype)
        out = self.bufs[k][:n]
        if device_copy:
            out = out.to(self.device, non_blocking=True)
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        if device_copy:
            self.events[k].record(stream)
        else:
            self.unguarded.append(k)
        self.used[k] = True
        return out


class UniformReplay(Storage):
    TransitionCLS = Transition

    def __init__(self, memory_size, batch_size, n_step=1, discount=1, history_length=1, keys=None):
        super(UniformReplay, self).__init__(memory_size, keys)
        self.batch_size = batch_size
        self.n_step = n_step
        self.discount = discount
        self.history_length = history_length
        self.pos = 0
        self._size = 0
        self._ring = None
        self._state_shape = self._state_dtype = self._action_dtype = None
        self._idx_up = None

    # -- host bookkeeping (replay.py:69-73, 105-110, 142-146) ---------------------------------
    def size(self):
        return self._size

    def full(self):
        return self._size == self.memory_size

    def valid_index(self, index):
        if index - self.history_length + 1 >= 0 and index + self.n_step < self.pos:
            return True
        if index - self.history_length + 1 >= self.pos and index + self.n_step < self.size():
            return True
        return False

    def compute_valid_indices(self):
        lo = list(range(self.history_length - 1, self.pos - self.n_step))
        hi = list(range(self.pos + self.history_length - 1, self.size() - self.n_step))
        return np.asarray(lo + hi)

    # -- feed (replay.py:75-90) ----------------------------------------------------------------
    def _device(self):
        dev = Config.DEVICE
        if dev.type != 'cuda':
            raise DraError("deeprl_amd replay lives in HBM: call select_device(gpu_id >= 0) first (no CPU path)")
        return dev

    def _on_device(self):
        """torch.cuda.device(ring's device) only when it is not already current (the context manager costs ~5 us, a host-environment
        agent step feeds four times, a prioritized one enters it four more times)."""
        dev = self._device()
        if dev.index is None or torch.cuda.current_device() == dev.index:
            return contextlib.nullcontext()
        return torch.cuda.device(dev)

    def _lazy_ring(self, state, action):
        if self._ring is not None:
            return
        dev = self._device()
        if isinstance(state, torch.Tensor):
            self._state_shape, self._state_dtype = tuple(state.shape), state.dtype
            fbytes = state.numel() * state.element_size()
        else:
            self._state_shape, self._state_dtype = tuple(state.shape), _NP2TORCH[state.dtype]
            fbytes = state.nbytes
        if isinstance(action, torch.Tensor):
            self._action_dtype, abytes = action.dtype, action.numel() * action.element_size()
        else:
            self._action_dtype, abytes = _NP2TORCH[action.dtype], action.nbytes
        with torch.cuda.device(dev):
            self._ring = ops.Ring(self.memory_size, fbytes, abytes, self.history_length, self.n_step, self.discount)
        self._idx_up = _PinnedUploader(torch.int64, 4096, dev)

    def feed(self, data):
        for k in data.keys():
            if k not in self.keys:
                raise RuntimeError('Undefined key')
        states, actions = data['state'], data['action']
        rewards, masks = data['reward'], data['mask']
        n_env = len(states)
        if n_env != 1:
            # replay.py:87 writes every env of a multi-env feed to storage[self.pos]: only 1-env
            # feeds are well defined in the reference; refuse the rest loudly.
            raise DraError("UniformReplay.feed: one environment per feed (got %d)" % n_env)
        state, action = states[0], actions[0]
        if not isinstance(state, torch.Tensor):
            state = np.asarray(state)
        if not isinstance(action, torch.Tensor):
            action = np.asarray(action)
        self._lazy_ring(state, action)
        reward = rewards[0]
        mask = masks[0]
        slot = self.pos
        with self._on_device():
            if isinstance(state, torch.Tensor):
                a_t = action if isinstance(action, torch.Tensor) else None
                self._ring.put_device(slot, state, actions=a_t, action_val=0 if a_t is not None else int(action),
                                      reward_val=float(reward), mask_val=int(mask))
            else:
                self._ring.put_host(slot, state, action, float(reward), int(mask))
        if slot >= self._size:
            self._size += 1
        self.pos = (slot + 1) % self.memory_size

    def advance(self, n=1):
        """`n` transitions were written into ring slots pos, pos+1, ... by a DEVICE producer (the device-resident
        actor / environment of csrc/learner.hip): the host part of feed() -- replay.py:84-90's pos / size bookkeeping."""
        if self._ring is None:
            raise DraError("advance(): no ring yet")
        for _ in range(int(n)):
            if self.pos >= self._size:
                self._size += 1
            self.pos = (self.pos + 1) % self.memory_size

    def device_ring(self, state_shape=(1, 84, 84), state_dtype=np.uint8, action_dtype=np.int64):
        """The HBM ring, created now (feed() creates it lazily from the first transition's shapes)."""
        self._lazy_ring(np.zeros(state_shape, dtype=state_dtype), np.zeros((), dtype=action_dtype))
        return self._ring

    # -- sample (replay.py:92-103, 112-140) ----------------------------------------------------
    def draw_indices(self, batch_size=None):
        """The reference's rejection loop (one np.random.randint(0, size) per attempt, valid_index on each), drawn in blocks of
        exactly the number still missing: same indices, same np.random state afterwards (randint(0, size, size=k) yields the
        values of k scalar draws: tests/test_cabi_symbols.py::test_vectorised_index_draw_consumes_the_reference_stream), at
        6 us instead of 40 us per minibatch of 32 -- the scalar loop was the largest host item of an async agent step."""
        if batch_size is None:
            batch_size = self.batch_size
        return draw_uniform_indices(self.size(), self.pos, batch_size, self.history_length, self.n_step)

    def gather(self, idx, want_f32=False, out=None):
        """Device gather of validated indices (numpy int64 or device tensor) -> dict of device tensors
        (`out`: a dict returned by an earlier call, refilled in place -- static buffers of a captured update)."""
        with self._on_device():
            if not isinstance(idx, torch.Tensor):
                idx = self._idx_up.upload(idx)
            return self._ring.gather(idx, self._state_shape, self._state_dtype, self._action_dtype, want_f32=want_f32,
                                     out=out)

    def sample(self, batch_size=None):
        g = self.gather(self.draw_indices(batch_size))
        return Transition(state=g['state'], action=g['action'], reward=g['reward'], next_state=g['next_state'],
                          mask=g['mask'])

    def construct_transition(self, index):
        if not self.valid_index(index):
            return None
        g = self.gather(np.asarray([index], dtype=np.int64))
        return Transition(state=g['state'][0], action=g['action'][0], reward=g['reward'][0],
                          next_state=g['next_state'][0], mask=g['mask'][0])

    def update_priorities(self, info):
        raise NotImplementedError

    # -- true resume: the ring's contents and the cursor ---------------------------------------
    _RING_SHARD = 1 << 28       # bytes of frames per shard file

    def save_full(self, prefix, ahead=0):
        """Writes <prefix>.replay (cursor, shapes, small arrays) and <prefix>.frames.<k> (the frame array in 256 MB shards:
        7 GB at 10^6 slots -- only the filled part is written).  ahead: slots beyond the logical size that already hold
        data (the device actor runs one agent step ahead of the replay's cursor)."""
        import pickle
        torch.cuda.synchronize()
        meta = dict(pos=self.pos, size=self._size, cls=type(self).__name__, memory_size=self.memory_size, shards=0,
                    state_shape=self._state_shape, state_dtype=str(self._state_dtype), action_dtype=str(self._action_dtype))
        if self._ring is not None:
            frames, actions, rewards, masks = self._ring.arrays()
            n = min(self.memory_size, self._size + int(ahead))
            meta["slots"] = n
            fb = self._ring.frame_bytes
            meta.update(actions=actions[:n * self._ring.action_bytes].cpu().numpy(), rewards=rewards[:n].cpu().numpy(),
                        masks=masks[:n].cpu().numpy(), frame_bytes=fb)
            per = max(1, self._RING_SHARD // fb)
            k = 0
            for lo in range(0, n, per):
                frames[lo * fb:min(n, lo + per) * fb].cpu().numpy().tofile("%s.frames.%d" % (prefix, k))
                k += 1
            meta["shards"], meta["slots_per_shard"] = k, per
        meta.update(self._extra_state())
        with open(prefix + ".replay", "wb") as f:
            pickle.dump(meta, f)

    def load_full(self, prefix):
        import pickle
        with open(prefix + ".replay", "rb") as f:
            meta = pickle.load(f)
        if meta["cls"] != type(self).__name__ or meta["memory_size"] != self.memory_size:
            raise DraError("%s.replay holds a %s of %d slots" % (prefix, meta["cls"], meta["memory_size"]))
        if meta["shards"] or meta["size"]:
            if self._ring is None:
                sd = meta["state_dtype"]
                self.device_ring(meta["state_shape"], np.uint8 if "uint8" in sd else (np.float64 if "64" in sd else np.float32),
                                 np.int64 if "int" in meta["action_dtype"] else np.float64)
            frames, actions, rewards, masks = self._ring.arrays()
            if "frame_bytes" not in meta:
                raise DraError("%s.replay records %d transitions but no device ring (it was written before the first feed "
                               "reached the HBM ring): nothing to restore them from" % (prefix, meta["size"]))
            n, fb, per = meta.get("slots", meta["size"]), meta["frame_bytes"], meta.get("slots_per_shard", 1)
            for k in range(meta["shards"]):
                part = torch.from_numpy(np.fromfile("%s.frames.%d" % (prefix, k), dtype=np.uint8))
                frames[k * per * fb:k * per * fb + part.numel()].copy_(part)
            actions[:n * self._ring.action_bytes].copy_(torch.from_numpy(meta["actions"]))
            rewards[:n].copy_(torch.from_numpy(meta["rewards"]))
            masks[:n].copy_(torch.from_numpy(meta["masks"]))
        self.pos, self._size = meta["pos"], meta["size"]
        self._load_extra_state(meta)
        torch.cuda.synchronize()

    def _extra_state(self):
        return {}

    def _load_extra_state(self, meta):
        pass

    def close(self):
        if self._ring is not None:
            self._ring.close()
            self._ring = None


class PrioritizedReplay(UniformReplay):
    TransitionCLS = PrioritizedTransition

    def __init__(self, memory_size, batch_size, n_step=1, discount=1, history_length=1, keys=None):
        super(PrioritizedReplay, self).__init__(memory_size, batch_size, n_step, discount, history_length, keys)
        self.tree = None
        self._max_priority = 1
        self._min_priority = 1.0  # smallest priority ever offered (the exactness bound of the parallel tree update)
        self._pending = set()   # sum_tree.py:13 pending_idx
        self._write = 0         # sum_tree.py:7 write cursor
        self.ordered_updates = False   # True: always replay the reference's incremental walk (slow, exact for any values)
        self._u_up = self._leaf_up = self._prio_up = self._pos_up = None
        # {max_priority, min priority offered} on the DEVICE once the fused learner writes priorities back without a
        # host round trip (commit_device); the host attributes are then refreshed on demand
        self._stat = None
        self._stat_on_device = False
        self._draw_out = None   # pinned (leaf, priority, total) the sample kernel writes directly

    def _lazy_tree(self):
        if self.tree is None:
            dev = self._device()
            with torch.cuda.device(dev):
                self.tree = ops.SumTree(self.memory_size)
            self._u_up = _PinnedUploader(torch.float64, 1024, dev)
            self._leaf_up = _PinnedUploader(torch.int64, 1024, dev)
            self._prio_up = _PinnedUploader(torch.float64, 1024, dev)
            self._pos_up = _PinnedUploader(torch.int32, 1024, dev)
            self._stat = torch.tensor([float(self._max_priority), float(self._min_priority)], dtype=torch.float64, device=dev)

    def _extra_state(self):
        self._lazy_tree()
        mx = self.max_priority      # (refreshes the host attributes from the device pair when the learner owns them)
        return dict(tree=self.tree.as_tensor().cpu().numpy().copy(), max_priority=mx, min_priority=self._min_priority,
                    pending=sorted(self._pending), write=self._write, stat_on_device=self._stat_on_device)

    def _load_extra_state(self, meta):
        self._lazy_tree()
        self.tree.as_tensor().copy_(torch.from_numpy(meta["tree"]))
        self._max_priority, self._min_priority = meta["max_priority"], meta["min_priority"]
        self._pending, self._write = set(meta["pending"]), meta["write"]
        self._stat.copy_(torch.tensor([float(self._max_priority), float(self._min_priority)], dtype=torch.float64))
        self._stat_on_device = meta["stat_on_device"]

    @property
    def max_priority(self):
        """replay.py:159,195.  Lives on the device while the fused learner commits priorities there (one D2H to read it)."""
        if self._stat_on_device:
            self._max_priority, self._min_priority = self._stat.cpu().tolist()
        return self._max_priority

    @max_priority.setter
    def max_priority(self, value):
        self._max_priority = value
        if self._stat_on_device:
            self._stat[0] = float(value)

    def _exact_parallel(self):
        """The level-parallel tree update recomputes ancestors as left + right; that equals the reference's incremental
        `+= change` (sum_tree.py:16-20) exactly when every partial sum is representable: all leaves are multiples of
        u = ulp_f32(smallest priority) and capacity * max_priority / u <= 2^53."""
        lo, hi = float(self._min_priority), float(self._max_priority)
        if not (lo > 0.0) or not np.isfinite(hi):
            return False
        return self.memory_size * hi <= np.ldexp(1.0, 53 + int(np.floor(np.log2(lo))) - 23)

    def feed(self, data):
        super().feed(data)
        self._add_leaf()

    def _add_leaf(self):
        """SumTree.add (sum_tree.py:39-51) for the transition just written: the leaf at the write cursor gets max_priority;
        a leaf that was sampled and is overwritten before its priority came back is no longer pending."""
        self._lazy_tree()
        leaf = self._write + self.memory_size - 1
        self._pending.discard(leaf)
        with torch.cuda.device(self._device()):
            if self._stat_on_device:
                self.tree.set_from(leaf, self._stat)
            else:
                self.tree.set(leaf, float(self._max_priority))
        self._write += 1
        if self._write >= self.memory_size:
            self._write = 0

    def advance(self, n=1, stream=None, tree=True):
        """UniformReplay.advance + the tree side of feed() for transitions a DEVICE producer wrote into the ring.
        stream: where the tree kernels go (default: torch's current stream).  tree=False: cursor / size only -- the adds were
        issued inside the previous update's chain kernel (chain_fill)."""
        n = int(n)
        super().advance(n)
        if not tree:
            return
        self._lazy_tree()
        if not self._stat_on_device or n > 64 or n > self.memory_size:
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                for _ in range(n):
                    self._add_leaf()
            return
        for i in range(n):
            self._pending.discard((self._write + i) % self.memory_size + self.memory_size - 1)
        with self._on_device():
            self.tree.set_many_from(self._write, n, self._stat, stream=stream)   # one launch for the whole agent step's adds
        self._write = (self._write + n) % self.memory_size

    def draw_begin(self, batch_size=None, stream=None):
        """First half of draw(): B uniforms from python `random` (replay.py:169-172) and the tree descent enqueued on
        `stream` (default: the current one), its results going straight into pinned host memory.  draw_end() waits for them;
        anything enqueued in between (the device actor's forward passes) overlaps the host round trip."""
        if batch_size is None:
            batch_size = self.batch_size
        self._lazy_tree()
        u = [random.random() for _ in range(batch_size)]
        with self._on_device():
            if self._draw_out is None or self._draw_out[0].numel() < batch_size:
                self._draw_out = (torch.empty(batch_size, dtype=torch.int64).pin_memory(),
                                  torch.empty(batch_size, dtype=torch.float64).pin_memory(),
                                  torch.empty(1, dtype=torch.float64).pin_memory())
                self._draw_np = tuple(t.numpy() for t in self._draw_out)
                self._draw_ev = [torch.cuda.Event(), torch.cuda.Event()]
                self._draw_k = 0
            oi, op, ot = self._draw_out
            # the descent kernel READS its uniforms from pinned host memory and WRITES leaves / priorities / total into pinned
            # host memory: no packing kernels, no copy command -- one event wait is the whole host round trip of a draw
            self.tree.sample_into(self._u_up.upload(u, device_copy=False, stream=stream), oi, op, ot, stream=stream)
            self._u_up.consumed(stream)
            self._draw_k ^= 1
            ev = self._draw_ev[self._draw_k]
            ev.record(stream)
        return batch_size, ev

    def draw_end(self, pending_draw):
        """Second half of draw(): validity check, pending marks and padding on the host (replay.py:173-186), consuming python
        `random` exactly as the reference: one random.choice per padded slot.  Returns (tree_idx, sampling_prob, data_idx)."""
        batch_size, ev = pending_draw
        ev.synchronize()
        oi, op, ot = self._draw_np
        return self._finish_draw(oi[:batch_size].copy(), op[:batch_size].copy(), float(ot[0]), batch_size)

    def _finish_draw(self, tree_idx, p, total, batch_size):
        # the common case without the per-sample python loop: every drawn transition is valid -> nothing is skipped or padded
        di_all = tree_idx - (self.memory_size - 1)
        lo, hi = di_all - self.history_length + 1, di_all + self.n_step
        if bool((((lo >= 0) & (hi < self.pos)) | ((lo >= self.pos) & (hi < self.size()))).all()):
            self._pending.update(tree_idx.tolist())     # sum_tree.py:66
            return tree_idx, p / total, di_all
        picked = []
        for i in range(batch_size):
            ti = int(tree_idx[i])
            self._pending.add(ti)  # sum_tree.py:66 (before the validity check)
            di = ti - self.memory_size + 1
            if not self.valid_index(di):
                continue
            picked.append((ti, p[i] / total, di))
        while len(picked) < batch_size:
            picked.append(random.choice(picked))  # "This should rarely happen" (replay.py:184-186)
        return (np.asarray([t[0] for t in picked], dtype=np.int64), np.asarray([t[1] for t in picked], dtype=np.float64),
                np.asarray([t[2] for t in picked], dtype=np.int64))

    def draw(self, batch_size=None):
        """replay.py:164-186.  Returns (tree_idx, sampling_prob, data_idx) as numpy arrays."""
        return self.draw_end(self.draw_begin(batch_size))

    def sample(self, batch_size=None):
        tree_idx, prob, data_idx = self.draw(batch_size)
        g = self.gather(data_idx)
        dev = g['state'].device
        return PrioritizedTransition(state=g['state'], action=g['action'], reward=g['reward'],
                                     next_state=g['next_state'], mask=g['mask'],
                                     sampling_prob=torch.from_numpy(prob).to(dev),
                                     idx=torch.from_numpy(tree_idx).to(dev))

    def update_priorities(self, info):
        """replay.py:193-196 + sum_tree.py:54-60: max_priority tracks every offered priority; a tree
        update happens only for pending leaves, first occurrence wins."""
        if self._stat_on_device:        # back to host bookkeeping: fetch what the device accumulated
            _ = self.max_priority
            self._stat_on_device = False
        leaves, prios = [], []
        for idx, priority in info:
            self._max_priority = max(self._max_priority, priority)
            self._min_priority = min(self._min_priority, float(priority))
            idx = int(idx)
            if idx in self._pending:
                self._pending.remove(idx)
                leaves.append(idx)
                prios.append(float(priority))
        if leaves:
            with torch.cuda.device(self._device()):
                self.tree.update(self._leaf_up.upload(np.asarray(leaves, dtype=np.int64)),
                                 self._prio_up.upload(np.asarray(prios, dtype=np.float64)),
                                 ordered=self.ordered_updates or not self._exact_parallel())

    def commit_device(self, tree_idx, prio_f32, stream=None):
        """update_priorities(zip(tree_idx, prio)) with the priorities still on the device (f32 tensor, one per sampled
        transition, in sample order): the host applies pending_idx gating / first-writer-wins (which need no priority
        value) and the kernel writes the chosen leaves, keeps max_priority and falls back to the ordered walk by itself
        when the parallel update would not be exact.  No host synchronisation."""
        self._lazy_tree()
        if not self._stat_on_device:
            self._stat.copy_(torch.tensor([float(self._max_priority), float(self._min_priority)], dtype=torch.float64))
            self._stat_on_device = True
        leaves, pos = [], []
        for j, idx in enumerate(tree_idx):
            idx = int(idx)
            if idx in self._pending:
                self._pending.remove(idx)
                leaves.append(idx)
                pos.append(j)
        with self._on_device():
            if leaves:
                self.tree.commit_f32(self._leaf_up.upload(leaves, device_copy=False, stream=stream),
                                     self._pos_up.upload(pos, device_copy=False, stream=stream), prio_f32, self._stat,
                                     force_ordered=self.ordered_updates, stream=stream)
                self._leaf_up.consumed(stream)
                self._pos_up.consumed(stream)
            else:
                self.tree.commit_f32(None, None, prio_f32, self._stat, force_ordered=self.ordered_updates, stream=stream)

    def close(self):
        super().close()
        if self.tree is not None:
            self.tree.close()
            self.tree = None


class DeviceDraw:
    """Host side of PrioritizedReplay.sample() ON THE DEVICE (csrc/sumtree.hip dra_sumtree_per_chain2, captured into the
    learner's prioritized update behind its loss kernel): priorities of update t -> tree, adds of agent step t+1, descent +
    valid_index filter + random.choice padding of draw t+1, whose indices / sampling probabilities the NEXT update reads from
    device memory.  The host never waits for an update's priorities any more.  What stays here:

      * python's `random` stream.  The kernel consumes raw Mersenne-Twister words from a pinned ring in exactly the order
        replay.py:169-186 would (two per random.uniform, k-bit rejection draws for random.choice); the host generates them
        ahead with random.getrandbits and keeps (words produced, random.getstate()) checkpoints, so that release() can put
        the module's generator at exactly the word the device stopped at;
      * the bookkeeping of sum_tree.py's pending_idx and of the actor / update ring-slot hazard, ONE launch late: collect()
        reads the pinned block of the previous launch (spinning on its launch number only if the device is behind)."""
    REFILL = 4096

    def __init__(self, replay, learner):
        import collections
        import ctypes
        self.rp, self.L = replay, learner
        rp = replay
        rp._lazy_tree()
        if not rp._stat_on_device:
            rp._stat.copy_(torch.tensor([float(rp._max_priority), float(rp._min_priority)], dtype=torch.float64))
            rp._stat_on_device = True
        C = ops.PerChain2IO
        self.blocks = []
        for _ in range(4):
            t = torch.zeros(ctypes.sizeof(C), dtype=torch.uint8).pin_memory()
            io = C.from_address(t.data_ptr())
            raw = t.numpy()

            def view(field, dtype, count, raw=raw):
                off = getattr(C, field).offset
                return raw[off:off + count * np.dtype(dtype).itemsize].view(dtype)
            v = dict(head=view("add_n", np.int32, 6), i64=view("add_write0", np.int64, 4), produced=view("rng_produced", np.uint64, 1),
                     beta=view("beta_next", np.float32, 1), raw=view("out_raw_idx", np.int64, 1024), idx=view("out_idx", np.int64, 1024),
                     p=view("out_p", np.float64, 1024), total=view("out_total", np.float64, 1), tail=view("out_n_valid", np.int32, 2),
                     cursor=view("out_rng_cursor", np.uint64, 1), seq=view("out_seq", np.uint64, 1))
            self.blocks.append((io, t, v))
        self._words_t = torch.zeros(ops.PER_RNG_WORDS, dtype=torch.int32).pin_memory()
        self.words = self._words_t.numpy().view(np.uint32)
        self.produced = self.consumed = 0          # words generated / known to be consumed
        self.ckpt = collections.deque()            # (words produced before this refill, random.getstate())
        self.seq = 0                               # launches issued
        self.fifo = collections.deque()            # issued, not collected: (slot, launch number, leaves its adds overwrite)
        self.active = False
        # the newest draw the host has seen: after issued_idx() the minibatch of the update issued last, after release() the one
        # of the update to be issued next
        self.next_tree_idx = self.next_p = self.next_total = self.next_beta = None
        self._collect_leaves = []                  # leaves of the minibatch the next collected launch committed
        learner.set_per_chain2(rp.tree, rp._stat, [b for b, _, _ in self.blocks], self._words_t)

    # -- python `random`, generated ahead -------------------------------------------------------------------------------
    def _ensure_words(self, need):
        if self.produced - self.consumed >= need:
            return
        n = max(self.REFILL, int(need))
        R = ops.PER_RNG_WORDS
        if self.produced + n - self.consumed > R:
            raise DraError("DeviceDraw: the word ring is too small for this batch size")
        self.ckpt.append((self.produced, random.getstate()))
        w = np.frombuffer(random.getrandbits(32 * n).to_bytes(4 * n, "little"), dtype="<u4")
        pos = self.produced & (R - 1)
        first = min(n, R - pos)
        self.words[pos:pos + first] = w[:first]
        if first < n:
            self.words[:n - first] = w[first:]
        self.produced += n
        while len(self.ckpt) > 2 and self.ckpt[1][0] <= self.consumed:
            self.ckpt.popleft()

    def release(self):
        """Everything issued is collected and the module-level generator stands exactly where the reference's would: after
        the last word the device consumed.  The next fill() generates ahead again from there."""
        while self.fifo:
            self.collect()
        if self.ckpt:
            base, state = [c for c in self.ckpt if c[0] <= self.consumed][-1]
            random.setstate(state)
            if self.consumed > base:
                random.getrandbits(32 * (self.consumed - base))
            self.ckpt.clear()
        self.produced = self.consumed

    # -- one launch -----------------------------------------------------------------------------------------------------
    def start(self, tree_idx, p, total, beta, seq=0, cursor=0):
        """The first prioritized minibatch comes from the host (a classic draw, or a resumed run)."""
        rp = self.rp
        tree_idx = np.asarray(tree_idx, dtype=np.int64)
        p = np.asarray(p, dtype=np.float64)
        self.release()
        self.seq = int(seq)
        self.produced = self.consumed = int(cursor)
        self.L.per_chain2_seed(tree_idx, tree_idx - (rp.memory_size - 1), p / total, beta, cursor, seq)
        self.next_tree_idx, self.next_p, self.next_total, self.next_beta = tree_idx, p, float(total), float(beta)
        self._collect_leaves = tree_idx.tolist()
        self.active = True

    def fill(self, slot, add_n, beta_next):
        """Inputs of the launch inside the update about to be issued: the next agent step's adds and everything valid_index
        needs once they are in.  Nothing here depends on a draw."""
        rp = self.rp
        B, mem = rp.batch_size, rp.memory_size
        self._ensure_words(8 * B + 256)
        v = self.blocks[slot][2]
        pos, size = rp.pos, rp.size()
        for _ in range(int(add_n)):
            if pos >= size:
                size += 1
            pos = (pos + 1) % mem
        v["head"][:] = (int(add_n), B, B, int(bool(rp.ordered_updates)), rp.history_length, rp.n_step)
        v["i64"][:] = (int(rp._write), mem, pos, size)
        v["produced"][0] = self.produced
        v["beta"][0] = beta_next
        adds = [(rp._write + i) % mem + mem - 1 for i in range(int(add_n))]
        rp._write = (rp._write + int(add_n)) % mem
        self.seq += 1
        self.fifo.append((slot, self.seq, adds, float(beta_next)))

    def issued_idx(self):
        """Ring indices of the minibatch of the update issued last (fill() + learner.step_update()): the draw of the launch
        BEFORE the one just issued -- normally long complete."""
        while len(self.fifo) > 1:
            self.collect()
        return self.next_tree_idx - (self.rp.memory_size - 1)

    def collect(self):
        """The oldest uncollected launch: its draw becomes `next_*`; pending_idx in the order the device worked
        (commit of the launch's own minibatch, the adds, every leaf of the new draw)."""
        slot, seq, adds, beta = self.fifo.popleft()
        self.L.per_chain2_wait(slot, seq)
        v = self.blocks[slot][2]
        B = self.rp.batch_size
        flags = int(v["tail"][1])
        if flags:
            raise DraError("dra_sumtree_per_chain2: %s" % ("the word ring ran dry" if flags & 1 else "no valid transition in a draw"))
        pend = self.rp._pending
        pend.difference_update(self._collect_leaves)
        for leaf in adds:
            pend.discard(leaf)
        idx = v["idx"][:B].copy()
        if int(v["tail"][0]) == B:
            pend.update(idx.tolist())
        else:
            pend.update(v["raw"][:B].tolist())
        self._collect_leaves = idx.tolist()
        self.consumed = int(v["cursor"][0])
        self.next_tree_idx, self.next_p, self.next_total, self.next_beta = idx, v["p"][:B].copy(), float(v["total"][0]), beta


class ReplayWrapper:
    """Same surface as replay.py:199-278.  The reference runs the replay in a separate process so
    that sampling and the host->GPU copy overlap the learner; with the ring resident in HBM there
    is nothing left to hide, so the wrapper is a thin in-process delegate for both values of the
    `async` flag (third positional argument, or `async_=` / `**{'async': ...}`)."""
    FEED = 0
    SAMPLE = 1
    EXIT = 2
    UPDATE_PRIORITIES = 3

    def __init__(self, replay_cls, replay_kwargs, async_=True, **kw):
        if 'async' in kw:
            async_ = kw.pop('async')
        if kw:
            raise TypeError('unexpected arguments %s' % sorted(kw))
        self.replay_kwargs = replay_kwargs
        self.replay_cls = replay_cls
        self.async_ = async_
        self.cache_len = 2
        self.replay = replay_cls(**replay_kwargs)
        self.sample = self.replay.sample
        self.feed = self.replay.feed
        self.update_priorities = self.replay.update_priorities

    def size(self):
        return self.replay.size()

    def close(self):
        self.replay.close()
