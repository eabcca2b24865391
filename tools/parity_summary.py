#!/usr/bin/env python
"""gpurun_out/parity_errors.jsonl (appended by the GPU parity tests, tests/test_gpu_agents.py::_record_parity) ->
one JSON: per check family the number of cases and the MAXIMUM of every recorded error, plus the unambiguous-step maxima
of the schedule oracle.  Committed as profiles/rNN_parity_errors.json.

    python tools/parity_summary.py gpurun_out/parity_errors.jsonl > profiles/r03_parity_errors.json"""
import json
import sys


def main():
    fam = {}
    for line in open(sys.argv[1]):
        line = line.strip()
        if not line:
            continue
        r = json.loads(line)
        case = r.pop("case")
        name = case.split("[")[0]
        if "ambiguous" in case:
            name += " (steps with a ReLU input within 5e-7 of zero: reported, judged at 100x)"
        f = fam.setdefault(name, {"cases": 0, "max": {}})
        f["cases"] += 1
        for k, v in r.items():
            if k == "relu_margin":
                f["max"]["min_relu_margin"] = min(f["max"].get("min_relu_margin", 1e9), v)
            else:
                f["max"][k] = max(f["max"].get(k, 0.0), v)
    print(json.dumps({"what": "maxima of the errors measured by the GPU parity tests against the CPU oracle (relative unless "
                              "named *_abs; q / td relative to max(|value|, max|q|))", "families": fam}, indent=1))


if __name__ == "__main__":
    main()
