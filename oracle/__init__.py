"""CPU oracle for the filtering hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a CPU restatement (NumPy, fp32-faithful; plus a plain-C port under
``oracle/c``) of the per-timestep recursions of the reference package ``gaussfiltax``
(kostastsa/BayesianFiltering).  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker / the timed CPU baseline only.  Nothing under
``bayesianfiltering_amd/`` imports, links or executes anything from here; the product path
raises if the HIP library is missing.

PARITY UNPINNED: the reference cannot be executed in this build environment (``jax``,
``tensorflow_probability`` are not installed; ordinary ModuleNotFoundError) and
the reference's own tests hold no assertions, golden vectors or fixtures
(docs/tests/test_inference.py:74-104 only ``return`` values).  The oracle is therefore pinned
by what can be checked offline: textbook fp64 Kalman recursions, the discrete Riccati
steady state (scipy), public Threefry-2x32 known-answer vectors (Random123), and
finite-difference checks of every analytic Jacobian.  The only reference-generated data this path
can be checked against are the 20 PRNG keys printed in docs/notebooks/BOTExperiment.ipynb (cell 6):
PRNGKey / split reproduce them bit for bit (tests/golden/reference_notebook_keys.json).  The golden fixtures under
``tests/golden`` are produced by this oracle (``tests/golden/make_golden.py``).
"""
