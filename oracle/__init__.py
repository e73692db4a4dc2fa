"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path used solely as the checker.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the product
(``fie_amd``, ``src/``, the CLIs) never does and fails loudly when the HIP library is missing.
See nets.py for the per-function upstream citations and the pinning status ("parity unpinned" parts).
"""
