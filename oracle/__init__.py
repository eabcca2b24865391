"""CPU oracle for the rollout -> replay -> update hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy for byte/integer/fp64 arithmetic, plain
PyTorch fp32 CPU ops for the floating-point kernels) of the algorithms in the
reference ShangtongZhang/DeepRL, each function citing the reference file:line it
follows.  It is the checker for the HIP path, never the thing shipped or
measured: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import it.  deeprl_amd/ never does.

Parity pinning: the reference has NO tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, run in the authoring container through tests/ref_shim.py and committed
as fixtures under tests/golden/ (generator: tests/golden/make_golden.py).
tests/test_oracle_vs_golden.py checks every function here against them.
"""
