"""The reference's own examples.py against the drop-in surface (CPU, no GPU work): with deeprl_amd installed as
`deep_rl`, the file -- modulo the `async` keyword it cannot legally contain on Python >= 3.7 (examples.py:116,149,180,
214) -- executes, and every global name its entry functions for the in-scope algorithms reference is
provided by the package.  Skipped where /root/reference is absent (the GPU box)."""
import builtins
import dis
import os
import re
import types

import pytest

REF = os.environ.get("DEEPRL_REFERENCE_ROOT", "/root/reference")
IN_SCOPE = ["ddpg_continuous", "td3_continuous", "option_critic_feature", "option_critic_pixel", "dqn_feature", "dqn_pixel", "quantile_regression_dqn_feature", "quantile_regression_dqn_pixel",
            "categorical_dqn_feature", "categorical_dqn_pixel", "rainbow_feature", "rainbow_pixel", "a2c_feature",
            "a2c_pixel", "a2c_continuous", "n_step_dqn_feature", "n_step_dqn_pixel", "ppo_continuous", "ppo_pixel"]
OUT_OF_SCOPE_NAMES = set()      # every name examples.py uses is provided


def _global_names(fn):
    names = set()
    todo = [fn.__code__]
    while todo:
        code = todo.pop()
        for ins in dis.get_instructions(code):
            if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME"):
                names.add(ins.argval)
        todo += [c for c in code.co_consts if isinstance(c, types.CodeType)]
    return names


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "examples.py")), reason="reference tree not present")
def test_reference_examples_resolve_against_the_package():
    import sys
    import deeprl_amd
    saved = {k: v for k, v in sys.modules.items() if k == "deep_rl" or k.startswith("deep_rl.")}
    try:
        deeprl_amd.install_as_deep_rl()
        _check_examples(deeprl_amd)
    finally:                                   # other tests import the REAL reference under the same name
        for k in [k for k in sys.modules if k == "deep_rl" or k.startswith("deep_rl.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def _check_examples(deeprl_amd):
    src = re.sub(r"\basync\b", "async_", open(os.path.join(REF, "examples.py")).read())
    mod = types.ModuleType("ref_examples_dropin")
    exec(compile(src, "examples.py", "exec"), mod.__dict__)     # star-imports deep_rl == deeprl_amd; defines the entry points
    missing = {}
    for name in IN_SCOPE:
        fn = getattr(mod, name, None)
        assert callable(fn), "examples.py has no %s" % name
        lacking = [g for g in _global_names(fn) if g not in mod.__dict__ and not hasattr(builtins, g)]
        if lacking:
            missing[name] = lacking
    assert not missing, "names examples.py needs that the package does not export: %s" % missing
    # the whole file: only the out-of-scope agents may be absent
    every = set()
    for v in mod.__dict__.values():
        if isinstance(v, types.FunctionType) and v.__module__ == mod.__name__:
            every |= {g for g in _global_names(v) if g not in mod.__dict__ and not hasattr(builtins, g)}
    assert every <= OUT_OF_SCOPE_NAMES, "unexpected missing names: %s" % sorted(every - OUT_OF_SCOPE_NAMES)
    for name in ("OptionCriticAgent", "DDPGAgent", "TD3Agent"):      # real agents (tests/test_gpu_more_agents.py)
        assert issubclass(getattr(deeprl_amd, name), deeprl_amd.BaseAgent)


def test_replay_wrapper_accepts_every_spelling_of_the_async_flag():
    """replay.py:205: the third argument is called `async` in the reference; callers pass it positionally
    (examples.py:41,84), as `async_=` after the rename, or through **{'async': ...}."""
    from deeprl_amd.replay import ReplayWrapper, UniformReplay
    kw = dict(memory_size=16, batch_size=2)
    for w in (ReplayWrapper(UniformReplay, kw, True), ReplayWrapper(UniformReplay, kw, async_=False),
              ReplayWrapper(UniformReplay, kw, **{"async": True})):
        assert isinstance(w.replay, UniformReplay) and w.size() == 0
    with pytest.raises(TypeError):
        ReplayWrapper(UniformReplay, kw, bogus=1)
