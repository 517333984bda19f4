"""(ncon, Jacobian entries, touched keys) statistics of the oracle on a random policy (config 3's), box-box 8 points."""
import sys, os, warnings, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from multiprocessing import Pool
from robopianist_amd.model import scene
from robopianist_amd import engine
from oracle.rp_oracle import Oracle, lib

def work(args):
    bbmax, seed, nstep = args
    L = lib(); L.rpo_debug_set_boxbox_max.argtypes = [ctypes.c_int]; L.rpo_debug_set_boxbox_max(bbmax)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False)
    m = si.model
    blob = engine.make_blob(m, si.key_joint_ids)
    o = Oracle(m, blob); o.reset()
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    rng = np.random.default_rng(seed)
    # dofs in the support of a body = number of joints on its path to the root
    nanc = np.zeros(m.nbody, int)
    for b in range(1, m.nbody):
        nanc[b] = nanc[m.body_parentid[b]] + m.body_jntnum[b]
    gb = m.geom_bodyid
    out = []
    keys = set(int(k) for k in si.key_joint_ids)
    for i in range(nstep):
        if i % 10 == 0:
            o.ctrl[:] = lo + rng.uniform(0, 1, m.nu) * (hi - lo)
        if i % 1580 == 0: o.reset()
        o.step(1)
        c = o.contact.reshape(-1, 16)
        ne = sum(nanc[gb[int(x[13])]] + nanc[gb[int(x[14])]] for x in c)
        out.append((len(c), ne))
    return np.array(out)

if __name__ == "__main__":
    nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 1580
    nenv = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    with Pool(8) as pool:
        for bbmax in (3, 8):
            res = np.concatenate(pool.map(work, [(bbmax, s, nstep) for s in range(nenv)]))
            nc, ne = res[:, 0], res[:, 1]
            print(f"boxbox_max={bbmax}: {len(nc)} mj_steps; ncon p99 {np.percentile(nc,99):.0f} p99.9 {np.percentile(nc,99.9):.0f} max {nc.max()}; "
                  f"entries p99 {np.percentile(ne,99):.0f} p99.9 {np.percentile(ne,99.9):.0f} max {ne.max()}")
            for cn, ce in ((32, 256), (48, 512), (64, 768), (64, 1024)):
                print(f"   overflow of ({cn} contacts, {ce} entries): {((nc > cn) | (ne > ce)).mean():.5f} of mj_steps (contacts {(nc > cn).mean():.5f}, entries {(ne > ce).mean():.5f})")
