#!/usr/bin/env python
"""Where does a resumed run part ways with the uninterrupted one?  (tests/resume_worker.py outputs)"""
import subprocess
import sys
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = os.path.join(ROOT, "tests", "resume_worker.py")
kind = sys.argv[1] if len(sys.argv) > 1 else "dqn_async"
n1, n2 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (40, 20)
tmp = "/tmp/diag_resume"
os.makedirs(tmp, exist_ok=True)


def run(*a):
    r = subprocess.run([sys.executable, W] + [str(x) for x in a], cwd=ROOT, capture_output=True, text=True)
    if r.returncode:
        print(r.stdout[-2000:], r.stderr[-3000:])
        sys.exit(1)


run(kind, tmp + "/whole.npz", "run", n1 + n2)
run(kind, tmp + "/p1.npz", "save", n1, tmp + "/ck")
run(kind, tmp + "/p2.npz", "load", n2, tmp + "/ck")
run(kind, tmp + "/mid.npz", "run", n1)
a, b, m, p1 = (dict(np.load(tmp + "/" + f)) for f in ("whole.npz", "p2.npz", "mid.npz", "p1.npz"))
print("save-run == plain run of the same length:", all(np.array_equal(m[k], p1[k]) for k in m))
for k in a:
    if not np.array_equal(a[k], b[k]):
        d = np.flatnonzero(a[k].reshape(-1) != b[k].reshape(-1))
        unit = 7056 if k == "frames" else (8 if k == "act" else 1)
        print("%-28s differs: %d elements, first at %d (slot %d), last slot %d" % (k, d.size, d[0], d[0] // unit, d[-1] // unit))
    else:
        print("%-28s equal" % k)
