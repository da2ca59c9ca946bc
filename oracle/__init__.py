"""CPU oracle for the sgl_kernel hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (sgl-cpu-tests_amd/sgl_kernel) never imports it and has no CPU compute fallback.

What it restates: the reference repo has no implementation of its own — the arithmetic lives in the
absent third-party CPU build of sgl_kernel (sgl-project/sglang `sgl-kernel`, no version pinned
anywhere in the reference).  What the reference *does* hold are the pure-torch oracles embedded in
its test files, which it treats as ground truth.  This package restates those oracles
(`oracle/moe.py`, `oracle/routing.py`, ...; each function cites the file:line it follows) and, for
the fp8 fused_experts hot path, a second independent restatement in plain C (`oracle/c/`) that is
also the `cpu_baseline` timed by bench.py.

Pinning: every restatement is checked (tests/test_oracle_golden.py, `-m "not gpu"`) against golden
vectors under tests/golden/ that were produced by running the reference's own embedded oracle
functions in the build container (tests/golden/make_golden.py).  Parity status: pinned at the
operator boundary; the rounding points *inside* the absent C++ kernels are not observable, so
below the boundary this repo fixes its own.
"""
