"""Flat parameter storage + the fused clip/optimiser step (K11).

`config.optimizer_fn(params)` in examples.py builds a torch.optim.RMSprop / Adam
(examples.py:67-68,139,204,370,508-509,534).  `FusedOptimizer.adopt` keeps that object as the
carrier of hyper-parameters (lr schedules keep working) but replaces its ~100 tiny ATen kernels
per step (deep_rl/agent/DQN_agent.py:130-134) with two HIP launches over ONE flat fp32 buffer that
all parameters (and their .grad views) alias.
"""
import torch

from . import ops
from ._lib import DraError


# the three NatureConvBody weights (network_bodies.py:14-20): stored [(c,kh,kw)][oc] so that the one-round-trip
# convolution kernels read them without a per-step layout conversion
NATURE_CONV_SHAPES = {(32, 4, 8, 8), (64, 32, 4, 4), (64, 64, 3, 3)}


def nature_conv_weights(params):
    return [p for p in params if p.dim() == 4 and tuple(p.shape) in NATURE_CONV_SHAPES]


class FlatParams:
    """Re-homes parameters into one contiguous fp32 buffer (each tensor 16-byte aligned) and gives
    every parameter a .grad view into a matching flat gradient buffer."""

    def __init__(self, params, align=4, koc=()):
        """`koc`: parameters (4-D conv weights [OC,C,KH,KW]) to store in the [(c,kh,kw)][oc] layout
        the one-round-trip conv kernels read; the module keeps seeing a [OC,C,KH,KW] (strided) view."""
        koc_ids = {id(p) for p in koc}
        self.params = []
        seen = set()
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise DraError("no parameters")
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise DraError("FlatParams supports float32 parameters")
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            if id(p) in koc_ids:
                oc, c, kh, kw = p.shape
                self.flat[o:o + n].copy_(p.data.permute(1, 2, 3, 0).reshape(-1))
                p.data = self.flat[o:o + n].view(c, kh, kw, oc).permute(3, 0, 1, 2)
                p.grad = self.grad[o:o + n].view(c, kh, kw, oc).permute(3, 0, 1, 2)
            else:
                self.flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()

    def offset_of(self, p):
        for q, o in zip(self.params, self.offsets):
            if q is p:
                return o
        raise KeyError("parameter not in this flat buffer")

    def view(self, buf, p):
        o = self.offset_of(p)
        return buf[o:o + p.numel()].view(p.shape)


class FusedOptimizer:
    """clip_grad_norm_ + optimizer.step() as two launches (dra_grad_sqnorm, dra_*_step)."""

    def __init__(self, flat, kind, hyper, torch_optimizer=None):
        self.flat = flat
        self.kind = kind
        self.hyper = hyper
        self.torch_optimizer = torch_optimizer
        dev = flat.flat.device
        self.state1 = torch.zeros_like(flat.flat)
        self.state2 = torch.zeros_like(flat.flat)
        self.n_partials = ops.norm_partials()
        self.partials = torch.zeros(self.n_partials, dtype=torch.float64, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.steps = 0
        # graph mode (agents capture the whole update into a hipGraph): every kernel argument must be constant
        # across steps, so Adam's step-dependent scalars live in device memory and are refreshed by prepare_step()
        self.graph_mode = False
        self._hyper_dev = None
        self._hyper_up = None
        # conv layers whose slab folds wait for step() (nets.direct_param_grads(defer_folds_to=self)): (offset, floats, slabs,
        # slab stride, slab count) per layer
        self._pending_folds = []
        self._partials_segs = None

    def defer_fold(self, grad_view, floats, slabs, n_slabs):
        """A conv layer's backward hands its unfolded weight-gradient slabs over: step() folds them into
        flat.grad[offset : offset + floats] inside the launch that forms the gradient norm.  False: not this optimizer's
        buffer / not aligned -- the layer folds as before."""
        off = (grad_view.data_ptr() - self.flat.grad.data_ptr()) // 4
        if (grad_view.data_ptr() - self.flat.grad.data_ptr()) % 16 or off < 0 or off + floats > self.flat.numel or floats % 4 \
                or n_slabs < 1 or slabs.data_ptr() % 16:
            return False
        self._pending_folds.append((int(off), int(floats), slabs, int(floats), int(n_slabs)))
        return True

    def _fold_pending(self, want_norm):
        """Folds the registered layers.  -> (partials, count) covering the WHOLE gradient when one launch could do everything
        (segments contiguous from offset 0: the NatureConvBody layers of a FlatParams buffer), else None (folded, no norm)."""
        segs = sorted(self._pending_folds, key=lambda t: t[0])
        self._pending_folds = []
        f = self.flat
        contiguous = segs[0][0] == 0 and all(segs[i][0] == segs[i - 1][0] + segs[i - 1][1] for i in range(1, len(segs)))
        if self._partials_segs is None:
            self._partials_segs = torch.zeros(ops.norm_partials_max(), dtype=torch.float64, device=f.flat.device)
        if contiguous and len(segs) <= ops.MAX_FOLD_SEGS and f.numel % 4 == 0:
            n = ops.grad_sqnorm_segs(f.grad, segs, self._partials_segs)
            return self._partials_segs, n
        for off, cnt, slabs, stride, ns in segs:
            ops.grad_sqnorm_segs(f.grad[off:off + cnt], [(0, cnt, slabs, stride, ns)], self._partials_segs)
        return None

    @classmethod
    def adopt(cls, torch_optimizer, flat=None):
        groups = torch_optimizer.param_groups
        if len(groups) != 1:
            raise DraError("FusedOptimizer supports a single param group")
        g = groups[0]
        if flat is None:
            flat = FlatParams(g['params'], koc=nature_conv_weights(g['params']))
        if isinstance(torch_optimizer, torch.optim.RMSprop):
            if g.get('momentum', 0) != 0 or g.get('weight_decay', 0) != 0:
                raise DraError("RMSprop momentum / weight_decay have no HIP kernel")
            kind = 'rmsprop'
        elif isinstance(torch_optimizer, torch.optim.Adam):
            if g.get('weight_decay', 0) != 0 or g.get('amsgrad', False):
                raise DraError("Adam weight_decay / amsgrad have no HIP kernel")
            kind = 'adam'
        else:
            raise DraError("no HIP kernel for optimizer %s" % type(torch_optimizer).__name__)
        return cls(flat, kind, g, torch_optimizer)

    def zero_grad(self, direct=False):
        """direct: the backward pass that follows runs under nets.direct_param_grads(True, covers=[self]) -- when the previous such
        pass overwrote every parameter's gradient (all_direct) the fill is skipped (6.75 MB per update of the pixel nets; the
        alignment gaps between parameters are zero from construction and never written)."""
        self._pending_folds = []
        if direct and getattr(self, 'all_direct', False):
            return
        self.flat.zero_grad()

    def enable_graph_mode(self):
        if not self.graph_mode:
            dev = self.flat.flat.device
            self._hyper_dev = torch.zeros(2, dtype=torch.float32, device=dev)
            from .replay import _PinnedUploader
            self._hyper_up = _PinnedUploader(torch.float32, 2, dev)
            self.graph_mode = True

    def enable_block_mode(self, k):
        """Graph mode for k consecutive steps inside ONE captured graph: step j of the block reads its Adam scalars from row j of
        a [k, 2] device block that prepare_steps(k) fills with one copy."""
        self.enable_graph_mode()
        if getattr(self, '_hyper_block', None) is None or self._hyper_block.shape[0] != k:
            dev = self.flat.flat.device
            self._hyper_block = torch.zeros((k, 2), dtype=torch.float32, device=dev)
            from .replay import _PinnedUploader
            self._hyper_block_up = _PinnedUploader(torch.float32, 2 * k, dev)
        return self._hyper_block

    def prepare_steps(self, k):
        """prepare_step() for the next k steps at once (same host arithmetic per step: dra_adam_hyper), one upload."""
        vals = []
        if self.kind == 'adam':
            import ctypes
            from ._lib import lib
            b1, b2 = self.hyper['betas']
            hp = (ctypes.c_float * 2)()
            for _ in range(k):
                self.steps += 1
                lib.dra_adam_hyper(float(self.hyper['lr']), float(b1), float(b2), int(self.steps), hp)
                vals += [hp[0], hp[1]]
            self._hyper_block_up.upload_into(self._hyper_block.view(-1), vals)
        else:
            self.steps += k

    def hyper_signature(self):
        """Hyper-parameters that are baked into a captured update (a change invalidates the graph)."""
        h = self.hyper
        return (self.kind, h['lr'], h.get('alpha'), h.get('eps'), h.get('centered'), tuple(h.get('betas', ())))

    def prepare_step(self):
        """Graph mode, OUTSIDE the captured region and before each replay: count the step and refresh Adam's
        bias-corrected scalars in device memory (same host arithmetic as the eager path: dra_adam_hyper)."""
        self.steps += 1
        if self.kind == 'adam':
            import ctypes
            from ._lib import lib
            b1, b2 = self.hyper['betas']
            hp = (ctypes.c_float * 2)()
            lib.dra_adam_hyper(float(self.hyper['lr']), float(b1), float(b2), int(self.steps), hp)
            self._hyper_up.upload_into(self._hyper_dev, [hp[0], hp[1]])

    def step(self, max_norm=None):
        """max_norm None / 0 = no clipping (then the norm pass is skipped)."""
        f, h = self.flat, self.hyper
        clip = bool(max_norm)
        folded = self._fold_pending(clip) if self._pending_folds else None
        if folded is not None:      # the deferred folds' launch formed every partial sum of squares as well
            partials, n_partials = (folded if clip else (None, self.n_partials))
        else:
            if clip:
                ops.grad_sqnorm(f.grad, self.partials)
            partials, n_partials = (self.partials if clip else None), self.n_partials
        if self.graph_mode:   # steps / Adam scalars were advanced by prepare_step()
            if self.kind == 'rmsprop':
                ops.rmsprop_step(f.flat, f.grad, self.state1, self.state2, partials, n_partials, max_norm or 0.0,
                                 h['lr'], h['alpha'], h['eps'], h['centered'], self.norm if clip else None)
            else:
                b1, b2 = h['betas']
                ops.adam_step_dev(f.flat, f.grad, self.state1, self.state2, partials, n_partials, max_norm or 0.0,
                                  b1, b2, h['eps'], self._hyper_dev, self.norm if clip else None)
            return
        self.steps += 1
        if self.kind == 'rmsprop':
            ops.rmsprop_step(f.flat, f.grad, self.state1, self.state2, partials, n_partials, max_norm or 0.0,
                             h['lr'], h['alpha'], h['eps'], h['centered'], self.norm if clip else None)
        else:
            b1, b2 = h['betas']
            ops.adam_step(f.flat, f.grad, self.state1, self.state2, partials, n_partials, max_norm or 0.0, h['lr'],
                          b1, b2, h['eps'], self.steps, self.norm if clip else None)
