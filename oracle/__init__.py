"""CPU oracle for the ALD reconstruction hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and there only
as the checker / the timed CPU baseline -- never as a fallback for the HIP path.

Every function is a plain numpy (k-space / integer / resampling logic) or torch-CPU-functional
(score network) restatement of the reference's algorithm and cites the reference file:line it
follows.  Parity is PINNED: ``tests/test_oracle_golden.py`` checks every function here against the
fixtures under ``tests/golden/`` that ``tests/golden/make_golden.py`` produced by importing the
reference itself in the build container.
"""
