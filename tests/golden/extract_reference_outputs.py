#!/usr/bin/env python3
"""Extract the numbers the reference itself recorded -- the stored cell outputs of its notebooks -- as fixtures.

The reference's tests assert nothing and JAX is not installed here, but three notebooks under docs/notebooks/ still
carry the printed results of the author's own runs, for seeds and model settings that are spelled out in the same
notebook.  Those printed numbers are reference-generated known answers; only they (and the settings needed to re-run
the computation) are copied here, no source text.

reference_notebook_keys.json     BOTExperiment.ipynb cell 6: ``next_key = jr.PRNGKey(1)``; ten times
                                 ``key0, key, next_key = jr.split(next_key, 3)`` with key0 / key printed: 20 keys.
reference_notebook_outputs.json  Experiment_TSP_2023.ipynb cell 6 (Lorenz-63 + 0.001 |x|^2 emission, T = 100,
                                 PRNGKey(0) chain, the two simulations whose output was stored): RMSE of
                                 gaussian_sum_filter (M = 2), unscented_gaussian_sum_filter (M = 2, ParamsUKF(1,0,0))
                                 and bootstrap_particle_filter (5e5 particles);
                                 test_single_run.ipynb cells 4-6, 9 (manoeuvring target, bearing + range, T = 30, explicit
                                 keys): GSF RMSE (nan), BPF RMSE (100 particles, ess 0.5), weights[:, 16] and the printed
                                 particle cloud particles[:, 16] (100 x 4).

                                 autocov_sims.ipynb cell 2: ``jrandom.multivariate_normal(PRNGKey(0), ones(3), eye(3), (10,))``
                                 pushed through the Hessians of a cubic map and 100 gradient steps; the matrix X the
                                 cell printed (the second of its two stored outputs is the one its current source
                                 produces): pins the 30 normal draws and their (10, 3) layout.

Not usable (the notebook's code cells were edited after the stored run, or the run used an older library): the
augmented-filter RMSEs of both notebooks (AGSF and UAGSF agree to 6 digits there, which the current node operations cannot
produce), every RMSE of BOTExperiment.ipynb (neither FCV gain 1.05 nor 1.0 reproduces them) and of Experiment A.ipynb.

Re-run (needs /root/reference):  python tests/golden/extract_reference_outputs.py
"""
import json
import os
import re

NB = "/root/reference/docs/notebooks/"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stream(nb, cell):
    return "".join("".join(o["text"]) for o in nb["cells"][cell]["outputs"] if o.get("output_type") == "stream")


def _floats(txt, label):
    return [float(v) for v in re.findall(r"\s" + label + r" RMSE:\s*(\S+)", txt)]


def keys():
    txt = _stream(json.load(open(NB + "BOTExperiment.ipynb")), 6)
    key0 = [[int(a), int(b)] for a, b in re.findall(r"key0:\s+\[\s*(\d+)\s+(\d+)\]", txt)]
    key = [[int(a), int(b)] for a, b in re.findall(r"key:\s+\[\s*(\d+)\s+(\d+)\]", txt)]
    assert len(key0) == len(key) == 10
    json.dump({"source": "docs/notebooks/BOTExperiment.ipynb cell 6 (stream output)",
               "recipe": "next_key = PRNGKey(1); repeat 10x: key0, key, next_key = split(next_key, 3)",
               "seed": 1, "key0": key0, "key": key}, open(os.path.join(OUT, "reference_notebook_keys.json"), "w"), indent=1)


def outputs():
    tsp = _stream(json.load(open(NB + "Experiment_TSP_2023.ipynb")), 6)
    n = len(_floats(tsp, "BPF"))               # complete simulations in the stored output
    single_nb = json.load(open(NB + "test_single_run.ipynb"))
    single = _stream(single_nb, 6)
    w16 = "".join(single_nb["cells"][9]["outputs"][-1]["data"]["text/plain"])
    doc = {
        "tsp": {
            "source": "docs/notebooks/Experiment_TSP_2023.ipynb cells 2-4, 6",
            "settings": {"state_dim": 3, "emission_dim": 1, "seq_length": 100, "initial_mean": [0.0, 0.0, 0.0],
                         "initial_covariance_diag": 1.0, "Q_diag": 20.0, "R_diag": 0.1,
                         "dynamics": "lorentz_63(sigma=10, rho=28, beta=2.667, dt=0.01) + q",
                         "emission": "0.001 * dot(x, x) + r", "inputs": "zeros", "seed": 0,
                         "key_chain": "key, next_key = split(next_key)", "gsf_components": 2, "ukf_params": [1, 0, 0],
                         "bpf_particles": 500000, "bpf_key": "key", "bpf_ess": 0.5},
            "rmse": {k: _floats(tsp, k)[:n] for k in ("GSF", "UGSF", "BPF")},
        },
        "single_run": {
            "source": "docs/notebooks/test_single_run.ipynb cells 2-4, 6, 9",
            "settings": {"seq_length": 30, "initial_mean": [-0.05, 0.001, 0.7, -0.05],
                         "initial_covariance_diag": [0.1, 0.005, 0.1, 0.01], "Q_diag": 1e-6, "R_diag": 25e-6,
                         "dynamics": "fManBOT(dt=0.5, acc=0.5)", "emission": "bearing + range",
                         "inputs": "[1]*10 + [0]*10 + [2]*10", "sample_key": [1426702441, 1492789755],
                         "gsf_components": 5, "next_key": [550753349, 3769041584], "key": "split(next_key)[0]",
                         "bpf_particles": 100, "bpf_ess": 0.5},
            "rmse": {"GSF": float(re.search(r"GSF RMSE:\s*(\S+)", single).group(1)),
                     "BPF": float(re.search(r"BPF RMSE:\s*(\S+)", single).group(1))},
            "bpf_weights_t16": [float(v) for v in re.findall(r"\d+\.\d+", w16)],
            # print(posterior_bpf["particles"][:, 16]): the whole cloud (100 x 4) after the 17th step
            "bpf_particles_t16": [float(v) for v in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", _stream(single_nb, 9))],
        },
    }
    auto = _stream(json.load(open(NB + "autocov_sims.ipynb")), 2)
    nums = [float(v) for v in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", auto)]
    assert len(nums) == 18
    doc["autocov"] = {
        "source": "docs/notebooks/autocov_sims.ipynb cells 1-2 (second printed matrix)",
        "settings": {"sample": "multivariate_normal(PRNGKey(0), ones(3), eye(3), (10,))", "eta": 0.01, "L": 0.1, "N": 10,
                     "steps": 100, "map": "x + 0.01 * [10 (x1^3 - x0 x1 x2), 28 x0 - x1 - x0 x2^2, x0 x1 - 2.667 x2]"},
        "X_first_output": [nums[0:3], nums[3:6], nums[6:9]],
        "X": [nums[9:12], nums[12:15], nums[15:18]],
    }
    json.dump(doc, open(os.path.join(OUT, "reference_notebook_outputs.json"), "w"), indent=1)
    print(json.dumps({"tsp": doc["tsp"]["rmse"], "single_run": doc["single_run"]["rmse"],
                      "n_w16": len(doc["single_run"]["bpf_weights_t16"])}))


if __name__ == "__main__":
    keys()
    outputs()
