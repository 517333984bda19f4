"""TEST INFRASTRUCTURE: bisect a real-MuJoCo recording over the narrow-phase choices the oracle could not pin from memory.

    python oracle/bisect_golden.py tests/golden/mujoco/config2.npz

A recording (oracle/make_golden.py) is replayed teacher-forced -- every mj_step restarted from MuJoCo's own state --
through the oracle under every combination of `rp_oracle.NARROW_PHASE_VARIANTS` (capsule-box point rule, box-box point
count, MPR stopping rule).  Per combination: the first mj_step whose contact COUNT differs from MuJoCo's, and the worst
relative velocity error over the steps whose counts agree.  The combination with no count mismatch and the smallest
error is the rule MuJoCo follows; if it is not the default (first line), the engine's narrow phase (csrc/rp_narrow.hpp)
needs the same change -- the switches exist in the oracle only.

Works on the synthetic recordings of tests/test_mujoco_golden.py too (the oracle as recorder: the default wins with
error 0), which is how this script is tested where MuJoCo cannot be installed."""
from __future__ import annotations

import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def bisect(path_or_dict, max_steps=None, grid=None):
    from oracle import rp_oracle
    from robopianist_amd import engine
    from robopianist_amd.tools import mjmodel_to_blob as imp
    d = dict(np.load(path_or_dict, allow_pickle=False)) if not isinstance(path_or_dict, dict) else path_or_dict
    model, keys = imp.model_from_npz(d)
    blob = engine.make_blob(model, keys)
    nsub = int(d["n_substeps"]) if "n_substeps" in d else 10
    n = d["qpos"].shape[0] - 1
    if max_steps:
        n = min(n, int(max_steps))
    V = rp_oracle.NARROW_PHASE_VARIANTS
    rows = []
    try:
        for cb, bb, mp in (grid if grid is not None else itertools.product(V["capsule_box"], V["boxbox_max"], V["mpr"])):
            rp_oracle.set_narrow_phase_variant(cb, bb, mp)
            orc = rp_oracle.Oracle(model, blob)
            first_bad, worst, agree = -1, 0.0, 0
            for i in range(n):
                orc.qpos[:] = d["qpos"][i]; orc.qvel[:] = d["qvel"][i]; orc.qacc_warmstart[:] = d["qacc_warmstart"][i]
                orc.ctrl[:] = d["ctrl"][i // nsub]
                orc.step1()   # (the recorded state was written from outside: its position / velocity stage first)
                orc.step(1)
                if orc.ncon != int(d["ncon"][i]):   # (ncon[i]: the contacts of the state step i produced, as make_golden records it)
                    if first_bad < 0:
                        first_bad = i
                    continue
                agree += 1
                den = max(np.abs(d["qvel"][i + 1] - d["qvel"][i]).max(), 1e-9)
                worst = max(worst, float(np.abs(orc.qvel - d["qvel"][i + 1]).max() / den))
            rows.append(dict(capsule_box=cb, boxbox_max=bb, mpr=mp, first_count_mismatch=first_bad, steps_agreeing=agree,
                             worst_rel_dv=worst))
    finally:
        rp_oracle.set_narrow_phase_variant()   # back to the defaults (process-wide switches)
    return rows


def main():
    if len(sys.argv) < 2:
        print(__doc__); sys.exit(2)
    rows = bisect(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    rows.sort(key=lambda r: (r["first_count_mismatch"] >= 0, -r["steps_agreeing"], r["worst_rel_dv"]))
    print(f"{'capsule_box':>11} {'boxbox_max':>10} {'mpr':>10} {'first count mismatch':>20} {'steps agreeing':>15} {'worst rel dv':>13}")
    for r in rows:
        print(f"{r['capsule_box']:>11} {r['boxbox_max']:>10} {r['mpr']:>10} {r['first_count_mismatch']:>20} {r['steps_agreeing']:>15} {r['worst_rel_dv']:>13.3e}")


if __name__ == "__main__":
    main()
