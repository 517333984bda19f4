"""Contact-count statistics of the oracle on the replay / a random policy, with box-box keeping 3 or 8 points."""
import sys, os, warnings, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from multiprocessing import Pool
from robopianist_amd.model import scene
from robopianist_amd import engine
from oracle.rp_oracle import Oracle, lib
import bench

def work(args):
    bbmax, policy, seed, nstep, prim = args
    L = lib(); L.rpo_debug_set_boxbox_max.argtypes = [ctypes.c_int]; L.rpo_debug_set_boxbox_max(bbmax)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=prim)
    m = si.model
    blob = engine.make_blob(m, si.key_joint_ids)
    o = Oracle(m, blob); o.reset()
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    rng = np.random.default_rng(seed)
    ctrl_seq, _ = bench.load_actions(m)
    gt = m.geom_type
    ncons, nbb, nbbpairs_gt3 = [], [], 0
    for i in range(nstep):
        if i % 10 == 0:
            if policy == "replay":
                o.ctrl[:] = ctrl_seq[((i // 10) + 7 * seed) % ctrl_seq.shape[0]]
            else:
                o.ctrl[:] = lo + rng.uniform(0, 1, m.nu) * (hi - lo)
        o.step(1)
        c = o.contact.reshape(-1, 16)
        ncons.append(len(c))
        bb = [(int(x[13]), int(x[14])) for x in c if gt[int(x[13])] == 6 and gt[int(x[14])] == 6]
        nbb.append(len(bb))
        from collections import Counter
        nbbpairs_gt3 += sum(1 for v in Counter(bb).values() if v > 3)
    return np.array(ncons), np.array(nbb), nbbpairs_gt3

if __name__ == "__main__":
    nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    with Pool(8) as pool:
        for policy in ("replay", "random"):
            for bbmax in (3, 8):
                res = pool.map(work, [(bbmax, policy, s, nstep, False) for s in range(8)])
                nc = np.concatenate([r[0] for r in res]); nb = np.concatenate([r[1] for r in res])
                print(f"{policy} boxbox_max={bbmax}: ncon mean {nc.mean():.1f} p50 {np.percentile(nc,50):.0f} p99 {np.percentile(nc,99):.0f} "
                      f"p99.9 {np.percentile(nc,99.9):.0f} max {nc.max()}  >24: {(nc>24).mean():.4f} >32: {(nc>32).mean():.4f} >48: {(nc>48).mean():.4f} >64: {(nc>64).mean():.5f}; "
                      f"box-box contacts mean {nb.mean():.1f} max {nb.max()}; pairs with >3 points: {sum(r[2] for r in res)}")
