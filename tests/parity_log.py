"""Measured parity errors of the GPU tests, kept as evidence (profiles/*_parity_errors.json)."""
import os


def record_parity(case, **errs):
    """Appends the measured maxima of a parity check to gpurun_out/parity_errors.jsonl (merged back from the GPU box;
    tools/parity_summary.py turns it into profiles/r03_parity_errors.json).  Never fails a test."""
    try:
        import json
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.jsonl"), "a") as f:
            f.write(json.dumps(dict(case=case, **{k: float(v) for k, v in errs.items()})) + "\n")
    except Exception:
        pass


def check_trajectory(name, traj, want, first_atol=1e-6, atol=2e-6, rtol=1e-5, gate_atol=5e-5, gate_frac=0.01, margins=None):
    """Parameters after every update of an UNSYNCHRONISED agent run against the reference run's digests (every 4099th element
    of every tensor).  The first update is the strict per-update check.  Later rows allow what differently gated ReLU units
    do (a pre-activation within fp32 summation noise of zero may be gated differently by two correct
    implementations; measured for ONE such unit: parameter errors 1e-6 .. 5e-5 in the unit's own fan-in / fan-out, and the
    difference then stays in the run): at most max(1, gate_frac x elements) digest elements beyond rtol / atol, none beyond
    gate_atol.  `margins` (optional: the smallest |ReLU input| of every update's differentiated forward, from the CPU oracle
    of the same run) sharpens that: strict until the first update whose margin is below 5e-7, then 4 elements per
    ambiguous update so far.  The per-update maxima go to the parity log; a step in them is the signature of a gate, not drift."""
    import numpy as np
    errs = [float(np.abs(a - b).max()) for a, b in zip(traj, want)]
    record_parity(name, first_update_abs=errs[0], last_update_abs=errs[-1], max_update_abs=max(errs), n_updates=len(errs),
                  **{"update_%02d_abs" % i: e for i, e in enumerate(errs)})
    assert len(traj) == len(want), (len(traj), len(want))
    np.testing.assert_allclose(traj[0], want[0], rtol=rtol, atol=first_atol, err_msg=name + ": first update")
    for i, (a, b) in enumerate(zip(traj, want)):
        if margins is None:
            check_gated(a, b, "%s: update %d" % (name, i), atol, rtol, gate_atol, gate_frac)
            continue
        k = int(sum(1 for m in margins[:i + 1] if m < 5e-7))
        if k == 0:
            np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg="%s: update %d (no ambiguous gate so far)" % (name, i))
        else:
            check_gated(a, b, "%s: update %d (%d ambiguous updates so far)" % (name, i, k), atol, rtol, gate_atol, gate_frac, allowed=4 * k)
    return errs


def check_gated(got, want, what, atol=2e-6, rtol=1e-5, gate_atol=5e-5, gate_frac=0.01, allowed=None):
    import numpy as np
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    off = err > atol + rtol * np.abs(want)
    allowed = max(1, int(gate_frac * off.size), int(allowed or 0))
    assert off.sum() <= allowed and err.max() <= gate_atol + rtol * np.abs(want).max(), \
        "%s: %d of %d elements beyond rtol %g / atol %g (allowed %d), max error %.3g" % (what, off.sum(), off.size, rtol, atol, allowed, err.max())
