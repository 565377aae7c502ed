"""CPU oracle for the NRMS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and there only as the checker (never as the thing measured as the product, never
as a fallback).  ``newsreclib_amd`` must not import it; a test enforces that.

Parity status: PINNED -- ``tests/golden/*.npz`` were generated in the build container by
``tests/golden/make_golden.py``, which imports the reference's own components from
``/root/reference`` (read-only) and records their outputs; ``tests/test_oracle_golden.py``
checks this restatement against those vectors.
"""
