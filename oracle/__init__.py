"""CPU oracle for the NRMS hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and there only as the checker (never as the thing measured as the product, never
as a fallback).  ``newsreclib_amd`` must not import it; a test enforces that.

Parity status: PINNED -- ``tests/golden/*.npz`` were generated in the build container by
``tests/golden/make_golden.py``, which imports the reference's own components from
``/root/reference`` (read-only) and records their outputs; ``tests/test_oracle_golden.py``
checks this restatement against those vectors.

What is NOT pinned against the reference itself (the reference modules behind these rows cannot be imported in the build
container, and the reference holds no test or fixture for them), each stated again in the module's own header:

* ``input_oracle.py`` (the host collate): ``rec_dataset.py`` -> ``mind_dataframe.py`` needs
  ``omegaconf`` -- restatement checked by hand-made known answers only: **parity unpinned**.
* ``metrics_oracle.py`` (row f.3, the aspect-metric half of the evaluation path): ``metrics/functional.py`` imports
  ``torchmetrics`` -- **parity unpinned**.  (The scoring half of row f.3 IS pinned: cached == uncached == nrms_oracle; the
  AUC / MRR / nDCG half is cross-checked against scikit-learn, an independent implementation of the same definitions.)
* ``losses_oracle.py`` (the SupCon half of row f.4): ``pytorch-metric-learning==2.2.0`` is absent and un-vendored -- its
  published ``GenericPairLoss`` / ``AvgNonZeroReducer`` arithmetic is restated: **parity unpinned** for that half.
* ``to_dense_batch`` (row a8): third-party ``torch_geometric==2.3.0``, absent -- pinned by known answers.
* The Lightning / Hydra drop-in route of INTEGRATION.md path A has never executed under real Lightning or Hydra (neither is
  installed in the build container or on the GPU box): ``newsreclib_amd/_lightning.py`` stands in for ``LightningModule``, and
  the constructor / forward / config-key / state_dict contract is pinned by ``tests/golden/reference_contract.json``
  (``tests/golden/make_contract.py``: ``ast`` + ``yaml`` over the reference sources).
"""
