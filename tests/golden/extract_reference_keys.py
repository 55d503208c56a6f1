#!/usr/bin/env python3
"""Extract the PRNG keys the reference itself recorded, as a fixture.

docs/notebooks/BOTExperiment.ipynb (cell 6) starts from ``next_key = jr.PRNGKey(1)`` and, for each of its 10
simulations, runs ``key0, key, next_key = jr.split(next_key, 3)`` and prints key0 and key.  The stored cell output
therefore holds 20 keys produced by the reference's own JAX run: a known-answer chain for PRNGKey + split (Threefry-2x32
in JAX's counter layout), the one piece of reference-generated data in the repository that this path can be pinned
against bit for bit.  (The RMSE values printed next to them are not usable: the notebook's model cell was edited after the
run -- neither its FCV gain of 1.05 nor 1.0 reproduces them.)

Only the printed numbers are copied.  Re-run (needs /root/reference):  python tests/golden/extract_reference_keys.py
"""
import json
import os
import re

NB = "/root/reference/docs/notebooks/BOTExperiment.ipynb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_notebook_keys.json")


def main():
    nb = json.load(open(NB))
    cell = nb["cells"][6]
    txt = "".join("".join(o["text"]) for o in cell["outputs"] if o.get("output_type") == "stream")
    key0 = [[int(a), int(b)] for a, b in re.findall(r"key0:\s+\[\s*(\d+)\s+(\d+)\]", txt)]
    key = [[int(a), int(b)] for a, b in re.findall(r"key:\s+\[\s*(\d+)\s+(\d+)\]", txt)]
    assert len(key0) == len(key) == 10
    json.dump({"source": "docs/notebooks/BOTExperiment.ipynb cell 6 (stream output)",
               "recipe": "next_key = PRNGKey(1); repeat 10x: key0, key, next_key = split(next_key, 3)",
               "seed": 1, "key0": key0, "key": key}, open(OUT, "w"), indent=1)
    print(OUT, len(key0) + len(key), "keys")


if __name__ == "__main__":
    main()
