"""Import shim for the READ-ONLY reference at /root/reference (TEST TOOLING ONLY).

The reference (ShangtongZhang/DeepRL) does not import on Python 3.10 as-is
(`async` used as an identifier in deep_rl/component/replay.py:205,210;
`collections.Sequence` in deep_rl/utils/misc.py:16; gym/baselines/skimage/
tensorboard/torchvision are not installed).  This module makes the untouched
on-disk sources importable so that golden vectors can be GENERATED from the
reference's own code (tests/golden/make_golden.py) and so that, when
/root/reference is present, the CPU tests can cross-check the oracle live.

Nothing in the product (deeprl_amd/) imports this file.  /root/reference does
not exist on the GPU box: every user of this module must call `available()`
first and skip otherwise.
"""
import collections
import collections.abc
import importlib.abc
import importlib.machinery
import importlib.util
import os
import re
import sys
import types

REFERENCE_ROOT = os.environ.get("DEEPRL_REFERENCE_ROOT", "/root/reference")

_loaded = None


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "deep_rl", "__init__.py"))


class _RunningMeanStd:
    """Stand-in for baselines.common.running_mean_std.RunningMeanStd (third-party,
    not vendored by the reference; public definition: count-weighted parallel
    mean/var merge, count initialised to epsilon=1e-4, population variance)."""

    def __init__(self, epsilon=1e-4, shape=()):
        import numpy as np
        self.mean = np.zeros(shape, "float64")
        self.var = np.ones(shape, "float64")
        self.count = epsilon

    def update(self, x):
        import numpy as np
        batch_mean = np.mean(x, axis=0)
        batch_var = np.var(x, axis=0)
        batch_count = x.shape[0]
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m_a = self.var * self.count
        m_b = batch_var * batch_count
        m2 = m_a + m_b + np.square(delta) * self.count * batch_count / tot_count
        self.mean = new_mean
        self.var = m2 / tot_count
        self.count = tot_count


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _install_stubs():
    if not hasattr(collections, "Sequence"):
        collections.Sequence = collections.abc.Sequence

    class Wrapper:
        def __init__(self, env=None):
            self.env = env

    class ObservationWrapper(Wrapper):
        pass

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete:
        def __init__(self, n):
            self.n = n

    class FrameStack:
        def __init__(self, env, k):
            self.env, self.k = env, k

    class VecEnv:
        def __init__(self, *a, **k):
            pass

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_histogram(self, *a, **k):
            pass

    if "gym" not in sys.modules:
        gym = _stub("gym", Wrapper=Wrapper, ObservationWrapper=ObservationWrapper,
                    make=lambda *a, **k: None)
        _stub("gym.envs")
        spaces = _stub("gym.spaces")
        box = _stub("gym.spaces.box", Box=Box)
        disc = _stub("gym.spaces.discrete", Discrete=Discrete)
        spaces.box, spaces.discrete = box, disc
        gym.spaces = spaces
    if "baselines" not in sys.modules:
        _stub("baselines")
        _stub("baselines.common")
        _stub("baselines.common.running_mean_std", RunningMeanStd=_RunningMeanStd)
        _stub("baselines.common.atari_wrappers", make_atari=lambda *a, **k: None,
              wrap_deepmind=lambda *a, **k: None, FrameStack=FrameStack)
        _stub("baselines.common.vec_env")
        _stub("baselines.common.vec_env.subproc_vec_env", SubprocVecEnv=VecEnv, VecEnv=VecEnv)
    if "skimage" not in sys.modules:
        _stub("skimage")
        _stub("skimage.io", imsave=lambda *a, **k: None)
    if "torchvision" not in sys.modules:
        _stub("torchvision")
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        _stub("tensorboard")
        import torch.utils
        tb = _stub("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
        torch.utils.tensorboard = tb
    for optional in ("matplotlib", "matplotlib.pyplot"):
        try:
            importlib.import_module(optional)
        except Exception:
            _stub(optional)


class _AsyncRenameLoader(importlib.machinery.SourceFileLoader):
    """Loads a reference source file with the token `async` renamed to `async_`."""

    def get_data(self, path):
        data = super().get_data(path)
        if path.endswith(".py"):
            return re.sub(rb"\basync\b", b"async_", data)
        return data

    def source_to_code(self, data, path, *, _optimize=-1):
        return compile(data, path, "exec", dont_inherit=True, optimize=_optimize)


class _RefFinder(importlib.abc.MetaPathFinder):
    """Resolves `deep_rl[.x.y]` from REFERENCE_ROOT; no .pyc is written anywhere."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "deep_rl" and not fullname.startswith("deep_rl."):
            return None
        parts = fullname.split(".")
        base = os.path.join(REFERENCE_ROOT, *parts)
        if os.path.isdir(base):
            fn = os.path.join(base, "__init__.py")
            return importlib.util.spec_from_file_location(
                fullname, fn, loader=_AsyncRenameLoader(fullname, fn),
                submodule_search_locations=[base])
        fn = base + ".py"
        if os.path.isfile(fn):
            return importlib.util.spec_from_file_location(
                fullname, fn, loader=_AsyncRenameLoader(fullname, fn))
        return None


def load():
    """Returns the reference `deep_rl` package (imported from REFERENCE_ROOT)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_stubs()
    sys.dont_write_bytecode = True  # never write into the read-only reference tree
    sys.meta_path.insert(0, _RefFinder())
    import deep_rl  # noqa: E402  (the reference, via _RefFinder)
    _loaded = deep_rl
    return deep_rl


def load_examples():
    """The reference examples.py (uses `async=` keywords, examples.py:116,149,180,214)."""
    load()
    fn = os.path.join(REFERENCE_ROOT, "examples.py")
    spec = importlib.util.spec_from_file_location(
        "ref_examples", fn, loader=_AsyncRenameLoader("ref_examples", fn))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
