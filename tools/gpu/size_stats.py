"""Round 3 (reused in round 6 on the new stand-in; argv[1] = capsule for capsule fingertips): distribution of the solver's problem sizes (contacts, Jacobian entries, dense rows, touched keys)
per env and mj_step, on the staggered replay and on the random policy -- sizes the capacity classes of the
lean solver stage.  GPU only."""
import warnings; warnings.simplefilter('ignore')
import sys, json, os; sys.path.insert(0, '.')
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from bench import load_actions
E = 4096
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=(len(sys.argv) > 1 and sys.argv[1] == "capsule"))
m = si.model
ctrl, _ = load_actions(m)
T = ctrl.shape[0]
out = {}
for mode in ("replay_staggered", "random"):
    phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E, precision=64)
    rng = np.random.default_rng(12345)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    phase = np.arange(E) % T
    rows = []
    for t in range(T + 40):
        if mode == "random":
            c = lo + rng.uniform(0, 1, (E, m.nu)) * (hi - lo)
        else:
            # env e replays row (t - phase[e]) once it has started; before that it holds row 0
            idx = np.clip(t - phase, 0, None) % T
            c = ctrl[idx]
        phys.set(engine.CTRL, c)
        if t < T:
            phys.step(10)
            continue
        for s in range(10):
            phys.step(1)
            h = phys.get(engine.DEBUG_HANDOVER_HDR)
            it = phys.get(engine.SOLVER_ITER)
            nd = np.array([bin(int(a) & 0xffffffff).count("1") + bin(int(b) & 0xffffffff).count("1") for a, b in zip(h[:, 2], h[:, 3])])
            rows.append(np.stack([h[:, 0], h[:, 1], nd, h[:, 4], h[:, 5], it & 255], 1))
        if mode == "replay_staggered" and t % 20 == 0:
            # restart finished episodes (physics only)
            pass
    r = np.concatenate(rows, 0)
    names = ["ncon", "nkt", "nd", "nent", "maxm", "iters"]
    d = {}
    for i, n in enumerate(names):
        v = r[:, i]
        d[n] = dict(mean=float(v.mean()), p50=float(np.percentile(v, 50)), p90=float(np.percentile(v, 90)),
                    p99=float(np.percentile(v, 99)), p999=float(np.percentile(v, 99.9)), max=int(v.max()))
    cls = {}
    for (nc, ne, ndm, nk) in [(8, 64, 16, 4), (12, 96, 20, 6), (12, 96, 24, 8), (16, 112, 24, 8), (16, 128, 24, 8), (16, 128, 32, 8), (24, 160, 32, 8), (24, 160, 36, 8), (24, 192, 36, 8), (24, 192, 40, 12), (32, 224, 44, 12), (32, 256, 57, 12)]:
        ok = (r[:, 0] <= nc) & (r[:, 3] <= ne) & (r[:, 2] <= ndm) & (r[:, 1] <= nk)
        cls[f"ncon<={nc},nent<={ne},nd<={ndm},nkt<={nk}"] = float(ok.mean())
    d["light_fraction"] = cls
    d["warn"] = int(phys.warn_flags.max())
    out[mode] = d
    print(mode, json.dumps(d, indent=1))
json.dump(out, open(os.environ.get("RP_STATS_OUT", "gpurun_out/r06_size_stats.json"), "w"), indent=1)
