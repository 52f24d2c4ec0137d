#!/usr/bin/env python3
"""Golden vectors for the preprocessing row by EXECUTING the reference notebook's own
preprocessing cells (nb:129-283) on a synthetic ratings.dat.

TEST INFRASTRUCTURE ONLY -- runs in the build container (needs /root/reference and pandas); never runs on the GPU box
and is never imported by the product package.

    python oracle/make_golden_prep.py [/root/reference/main_svae.ipynb]

Executed from the notebook (nothing is copied into this repo): code cells 2, 3, 4 (nb:129-283), unmodified, with
n_heldout_users = 750 and max_seq_len = 1000 as written.  Shims needed to run 2018 code on pandas 2.3 (neither changes
what is computed):
  * ``sys`` in the namespace (nb:207 calls sys.stdout.flush(); the notebook never imports sys)
  * ``get_count`` (nb:150-153) re-bound to ``tp.groupby(id).size()``: on pandas 0.2x ``groupby(id, as_index=False)
    .size()`` returned that Series (indexed by id); on pandas 2.x it returns a DataFrame and nb:160 raises IndexError.
The synthetic ratings.dat is rebuilt from a seed by synth_ratings() below (also used by the test), so the fixture
holds only the expected outputs: sha256 + line count + first lines of each of the six files the cells write.
Timestamps are unique within a user and all have 9 digits: the reference's sort is unstable for ties (nb:144,
quicksort) and pandas 2.x keeps the parsed column as objects, so ties / mixed digit counts are outside what this
fixture pins.
"""
import hashlib
import json
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "preprocess_synth.json")
FILES = ["train.csv", "validation_tr.csv", "validation_te.csv", "test_tr.csv", "test_te.csv", "unique_sid.txt"]


def synth_ratings(path, seed=7, n_users=2400, n_movies=1300):
    """userId::movieId::rating::timestamp lines in random order (MovieLens ratings.dat format)."""
    rs = np.random.RandomState(seed)
    pop = 1.0 / np.arange(1, n_movies + 1) ** 0.8
    pop /= pop.sum()
    rows = []
    for u in range(1, n_users + 1):
        if u in (17, 911) and n_movies >= 1250:
            n = 1250                                   # more than max_seq_len liked movies: the user is dropped
        else:
            n = int(np.clip(np.round(rs.lognormal(2.6, 0.9)), 1, 300))
        movies = rs.choice(n_movies, size=n, replace=False, p=pop) + 1
        ratings = rs.choice([1, 2, 3, 4, 5], size=n, p=[0.05, 0.1, 0.2, 0.35, 0.3]) if n < 1000 else \
            rs.choice([4, 5], size=n)
        stamps = 978300000 + rs.permutation(4000)[:n] * 200 + rs.randint(0, 200, size=n)     # unique per user
        rows.extend(zip([u] * n, movies.tolist(), ratings.tolist(), stamps.tolist()))
    order = rs.permutation(len(rows))
    with open(path, "w") as f:
        for i in order:
            f.write("%d::%d::%d::%d\n" % rows[i])
    return len(rows)


def digest(path):
    data = open(path, "rb").read()
    return dict(sha256=hashlib.sha256(data).hexdigest(), lines=data.count(b"\n"),
                head=data.decode().splitlines()[:6])


def main():
    import pandas as pd
    warnings.filterwarnings("ignore")
    nb_path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/main_svae.ipynb"
    nb = json.load(open(nb_path))
    code = ["".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code"]
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, "ml/")
        os.makedirs(base)
        n = synth_ratings(os.path.join(base, "ratings.dat"))
        hp = {"data_base": base}
        ns = dict(hyper_params=hp, os=os, pd=pd, np=np, sys=sys)
        exec(compile(code[2], "nb_cell_2", "exec"), ns)
        exec(compile(code[3], "nb_cell_3", "exec"), ns)
        ns["get_count"] = lambda tp, id: tp.groupby(id).size()
        exec(compile(code[4], "nb_cell_4", "exec"), ns)
        pro = os.path.join(base, "pro_sg")
        out = dict(seed=7, n_users=2400, n_movies=1300, ratings_lines=n, pandas=pd.__version__,
                   files={f: digest(os.path.join(pro, f)) for f in FILES})
        if os.environ.get("KEEP"):
            import shutil
            shutil.copytree(pro, os.environ["KEEP"], dirs_exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, {k: v["lines"] for k, v in out["files"].items()})


if __name__ == "__main__":
    main()
