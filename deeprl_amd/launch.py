"""Launcher for deep_rl-style example files on MI355X.

    python -m deeprl_amd.launch <examples.py> <entry> [key=value ...] [--gpu N] [--max-steps K] [--seed S]

<examples.py> is the reference's own examples.py (or any file written against `from deep_rl import *`, e.g.
deeprl_amd/zoo.py).  The file is executed with `deep_rl` bound to this package and with the one token the reference
cannot legally contain on Python >= 3.7 rewritten (`async=` -> `async_=`, examples.py:116,149,180,214); `--max-steps`
caps `config.max_steps` of the agent the entry point hands to run_steps (the reference's functions set it after merging
their kwargs, so it cannot be passed as one).
"""
import argparse
import ast
import re
import sys
import types


def load_examples(path, namespace_name="examples"):
    """Executes a deep_rl example file against deeprl_amd and returns it as a module object."""
    import deeprl_amd
    deeprl_amd.install_as_deep_rl()
    src = re.sub(r"\basync\b", "async_", open(path).read())
    mod = types.ModuleType(namespace_name)
    mod.__file__ = path
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


def run_entry(module, entry, max_steps=None, **kwargs):
    """Calls module.<entry>(**kwargs); with max_steps, run_steps sees config.max_steps = max_steps.  Returns the agent."""
    from . import support
    seen = {}
    real = module.__dict__.get("run_steps", support.run_steps)

    def capped(agent):
        if max_steps is not None:
            agent.config.max_steps = max_steps
        seen["agent"] = agent
        return real(agent)

    had = "run_steps" in module.__dict__
    module.run_steps = capped
    try:
        out = getattr(module, entry)(**kwargs)
    finally:
        if had:
            module.run_steps = real
    return seen.get("agent", out)


def _value(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("file")
    ap.add_argument("entry")
    ap.add_argument("kwargs", nargs="*", help="key=value, passed to the entry point")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--max-steps", type=float, default=None)
    ap.add_argument("--seed", type=int, default=None)
    args = ap.parse_args(argv)
    import os
    import deeprl_amd as d
    d.mkdir("log")
    d.mkdir("tf_log")
    d.set_one_thread()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:       # launched by torch.distributed.run: one rank per GPU, A2C / PPO become data-parallel (dist.py)
        import torch
        from . import dist as dp
        local = int(os.environ.get("LOCAL_RANK", "0"))
        args.gpu = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(args.gpu)
        dp.init("nccl" if torch.cuda.device_count() >= world else "gloo")
        if args.seed is None:
            args.seed = 1 + int(os.environ.get("RANK", "0"))     # distinct exploration per rank unless asked otherwise
    d.random_seed(args.seed)
    d.select_device(args.gpu)
    mod = load_examples(args.file)
    kw = {}
    for item in args.kwargs:
        k, _, v = item.partition("=")
        kw[k] = getattr(d, v) if isinstance(v, str) and hasattr(d, v) and v[:1].isupper() else _value(v)
    return run_entry(mod, args.entry, max_steps=args.max_steps, **kw)


if __name__ == "__main__":
    main()
    sys.exit(0)
