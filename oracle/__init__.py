"""CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a checker: a plain numpy / C restatement of what
tail-unica/hopwise computes on the KGE-scoring, negative-sampling and PGPR
beam-search path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
``hopwise_amd`` never imports from here and has no CPU fallback.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
(``/root/reference``) in the build container, runs the reference functions on
seeded inputs and stores inputs + outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every oracle function against them.
"""
