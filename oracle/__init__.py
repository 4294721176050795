"""CPU oracle for the pyphysim link-level hot path.  TEST INFRASTRUCTURE ONLY.

This package is a from-scratch NumPy (complex128 / int64) restatement of the
reference's per-realization chain.  Each function cites the
reference file:line it follows (paths relative to the reference checkout).

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker* / reported baseline,
never as the thing measured or shipped.  Nothing under ``pyphysim_amd/``
imports it; the product path fails loudly when the HIP library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference itself
in the build container (read-only, with the two stub modules in
``oracle/ref_shim``), runs every chain below with ``np.random.seed(base + r)``
on both sides, asserts equality (integers exactly, floats <= 1e-12) and writes
the inputs / intermediates / counters to ``tests/golden/*.npz``.  The
``-m "not gpu"`` tests re-check the oracle against those fixtures on every
run; the reference itself never travels to the GPU box.
"""
