"""TEST INFRASTRUCTURE ONLY - CPU restatement ("oracle") of the qmps hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path (``qmps_amd``) never imports it
and fails loudly when the HIP library is missing.
"""
