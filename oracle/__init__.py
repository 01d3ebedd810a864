"""
CPU oracle for the learn-nerf hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU / NumPy, float64 by default) of the
mathematics in the reference's ``learn_nerf/render.py``, ``model.py``,
``instant_ngp.py``, ``ref_nerf.py`` and ``train.py``.  Every function cites the
reference file:line it follows.

PARITY UNPINNED: the reference ships no golden vectors / known-answer tests for
this path and JAX/Flax/optax are not installable in the build container (plain
ModuleNotFoundError, no network), so the oracle cannot be checked against the
reference executing.  It is pinned instead by analytic known-answer identities
that follow from the reference source (tests/test_oracle_known_answers.py) and
by float64 finite-difference gradient checks.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product package (``learn-nerf_amd/learn_nerf``)
never does; it fails loudly when the HIP library is missing.
"""
