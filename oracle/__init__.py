"""CPU oracle for the R(2+1)D / loss / step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
directory: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

Every function restates, in plain fp32 PyTorch-CPU / NumPy, the algorithm of a
reference function (cited as ``file:line`` relative to the reference root).
Parity pinning: the reference holds no golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned by fixtures generated from the
reference itself, run on CPU in the build container by
``tests/golden/make_golden.py`` and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks the oracle against them.
"""
