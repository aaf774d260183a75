"""CPU oracle for the Sup3rGan hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a numpy restatement of what NREL/sup3r's Sup3rGan compute
core executes through TensorFlow 2.15 / keras 2.15 / phygnn 0.0.33 (none of
which is installable where this repo is built or run).  It is deliberately
written "as TF executes": every JSON layer is materialised as its own op
(REFLECT pad-3 -> valid conv -> crop-2, un-fused, channels-last fp32/fp64).

PARITY UNPINNED: the reference holds no golden tensors for this path
 and TF/phygnn cannot be imported here, so the oracle is pinned
by (1) agreement with an independent torch-CPU implementation
(tests/test_oracle_vs_torch.py), (2) exact-integer permutation cases and (3)
the reference's shape contracts (tests/training/test_load_configs.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  Nothing under ``sup3r_amd/``
imports it; the product path fails loudly when the HIP library is missing.
"""
