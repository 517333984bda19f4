"""Generates tests/golden/prf_sklearn.json: known answers of sklearn's
`precision_recall_fscore_support(average="binary", zero_division=1)` -- the call the reference's
MidiEvaluationWrapper makes per step (/root/reference/robopianist/wrappers/evaluation.py:139-141,167-169) --
on the corner cases (no positives at all; predictions but no true positives; misses only; all correct) and on
random 88-key rows.  Run where scikit-learn is importable:  python tests/golden/make_prf_golden.py"""
import json
import os
import warnings

import numpy as np
import sklearn
from sklearn.metrics import precision_recall_fscore_support


def main():
    rng = np.random.default_rng(20260925)
    z, o = np.zeros(88, int), np.ones(88, int)
    one = z.copy(); one[40] = 1
    other = z.copy(); other[41] = 1
    cases = [(z, z), (z, one), (one, z), (one, one), (o, o), (one, other), (one, one | other), (one | other, one),
             (np.zeros(1, int), np.zeros(1, int)), (np.ones(1, int), np.zeros(1, int)), (np.zeros(1, int), np.ones(1, int)),
             (np.ones(1, int), np.ones(1, int))]
    for _ in range(64):
        pt, pp = rng.random() * 0.15, rng.random() * 0.15
        cases.append(((rng.random(88) < pt).astype(int), (rng.random(88) < pp).astype(int)))
    out = []
    for yt, yp in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p, r, f, _ = precision_recall_fscore_support(y_true=yt, y_pred=yp, average="binary", zero_division=1)
        out.append({"y_true": yt.tolist(), "y_pred": yp.tolist(), "precision": float(p), "recall": float(r), "f1": float(f)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prf_sklearn.json")
    with open(path, "w") as fh:
        json.dump({"sklearn": sklearn.__version__, "cases": out}, fh)
    print(path, len(out), "cases")


if __name__ == "__main__":
    main()
