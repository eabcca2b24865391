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
