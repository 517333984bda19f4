"""TEST INFRASTRUCTURE: how much of the headline workload is an artefact of the STAND-IN hand.

The Shadow Hand of this repo is re-authored from memory (robopianist_amd/model/shadow_hand.py; the reference loads the
menagerie asset, /root/reference/robopianist/models/hands/shadow_hand.py:91-154, which is an empty submodule here), and
the scripted replay was trained on the real one.  Two reports, both from the CPU oracle:

  contact_residency(...)   along an action stream: per geom pair, the share of mj_steps it is in contact; the share of
                           contacts that are hand SELF-contacts / hand-hand / hand-key.
  joint_range_sweep(...)   every hand joint swept over its own range with all other joints at qpos0: the link pairs of
                           one hand that interpenetrate (two rigid links of one hand cannot, physically).

Used by tests/test_standin_report.py and by bench.py's CPU leg (the shares go into the bench line's `config`)."""
from __future__ import annotations

from collections import Counter

import numpy as np


def _short(name: str) -> str:
    return name.split("/")[-1]


def _side(name: str) -> str:
    n = _short(name)
    return "rh" if n.startswith("rh_") else ("lh" if n.startswith("lh_") else "piano")


def contact_residency(model, blob, ctrl_seq, hold=10, nstep=None, top=12):
    from oracle.rp_oracle import Oracle
    o = Oracle(model, blob)
    gm = model.names["geom"]
    n = int(nstep if nstep is not None else ctrl_seq.shape[0] * hold)
    pair_steps, kinds, ncs = Counter(), Counter(), []
    for i in range(n):
        o.ctrl[:] = ctrl_seq[(i // hold) % ctrl_seq.shape[0]]
        o.step(1)
        con = o.contact.reshape(-1, 16)
        ncs.append(len(con))
        seen = set()
        for c in con:
            a, b = gm[int(c[13])], gm[int(c[14])]
            sa, sb = _side(a), _side(b)
            kinds["hand_key" if "piano" in (sa, sb) else ("hand_self" if sa == sb else "hand_hand")] += 1
            seen.add((_short(a), _short(b)))
        for p in seen:
            pair_steps[p] += 1
    total = max(sum(kinds.values()), 1)
    return {
        "mj_steps": n, "mean_contacts": float(np.mean(ncs)), "max_contacts": int(np.max(ncs)),
        "share_of_contacts": {k: kinds[k] / total for k in ("hand_self", "hand_hand", "hand_key")},
        "pair_residency_top": [{"pair": list(p), "share_of_mj_steps": c / n} for p, c in pair_steps.most_common(top)],
    }


def joint_range_sweep(model, blob, samples=9, depth_floor=1e-4):
    """Returns [{joint, pair, max_depth_m, at_q}] for every (joint, geom pair of one hand) that penetrates by more than
    `depth_floor` somewhere in the joint's range, all other joints at qpos0 (keys untouched)."""
    from oracle.rp_oracle import Oracle
    o = Oracle(model, blob)
    gm, jn = model.names["geom"], model.names["joint"]
    q0 = np.array(model.qpos0, float)
    out = {}
    for j in range(int(model.njnt)):
        if _side(jn[j]) == "piano" or not int(model.jnt_limited[j]):
            continue
        lo, hi = (float(x) for x in model.jnt_range[j])
        for q in np.linspace(lo, hi, samples):
            o.reset()
            o.qpos[:] = q0
            o.qpos[j] = q
            o.step1()
            for c in o.contact.reshape(-1, 16):
                a, b = gm[int(c[13])], gm[int(c[14])]
                if _side(a) != _side(b) or _side(a) == "piano" or -float(c[0]) < depth_floor:
                    continue
                key = (_short(jn[j]), _short(a), _short(b))
                if key not in out or -float(c[0]) > out[key][0]:
                    out[key] = (-float(c[0]), float(q))
    return [{"joint": k[0], "pair": [k[1], k[2]], "max_depth_m": v[0], "at_q": v[1]}
            for k, v in sorted(out.items(), key=lambda kv: -kv[1][0])]
