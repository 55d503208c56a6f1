"""CPU oracle for the filtering hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a CPU restatement (NumPy, fp32-faithful; plus a plain-C port under
``oracle/c``) of the per-timestep recursions of the reference package ``gaussfiltax``
(kostastsa/BayesianFiltering).  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker / the timed CPU baseline only.  Nothing under
``bayesianfiltering_amd/`` imports, links or executes anything from here; the product path
raises if the HIP library is missing.

PARITY PIN: the reference cannot be executed in this build environment (``jax``,
``tensorflow_probability`` are not installed; ordinary ModuleNotFoundError) and
its own tests hold no assertions, golden vectors or fixtures (docs/tests/test_inference.py:74-104
only ``return`` values).  What pins this restatement to the real reference are the results the
reference's author recorded in the stored cell outputs of docs/notebooks/*.ipynb
(tests/golden/extract_reference_outputs.py): the 20 PRNG keys of BOTExperiment.ipynb (bit-exact),
the matrix autocov_sims.ipynb computes from 30 normal draws (all 8 printed digits),
the GSF / UGSF RMSEs of Experiment_TSP_2023.ipynb (<= 3e-6 relative, NaN pattern included) and the
GSF-NaN / 100-particle BPF RMSE / weights of test_single_run.ipynb (6e-7) --
tests/test_reference_recorded_outputs.py, tests/test_oracle_rng.py.  PARITY UNPINNED for the
augmented filters (their recorded RMSEs come from an older library state) and the legacy classes.
Beyond that the oracle is checked against offline mathematics: textbook fp64 Kalman recursions, the
discrete Riccati steady state (scipy), public Threefry-2x32 known-answer vectors (Random123), and
finite-difference checks of every analytic Jacobian.  The golden fixtures under ``tests/golden``
(*.npz) are produced by this oracle (``tests/golden/make_golden.py``).
"""
