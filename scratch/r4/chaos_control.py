"""Chaos control: oracle vs oracle with qpos0 perturbed by `eps`, on the Twinkle replay (hull / capsule fingertips);
also: oracle with MPR tolerance 0.99e-6 vs 1e-6, and the same in discrete-termination mode."""
import sys, os, warnings, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from robopianist_amd.model import scene
from robopianist_amd import engine
from oracle.rp_oracle import Oracle, lib
import bench

L = lib()
L.rpo_debug_set_mpr.argtypes = [ctypes.c_double, ctypes.c_int]

def traj(m, blob, ctrl_seq, nstep, eps=0.0, tol=1e-6, discrete=0, seed=0):
    L.rpo_debug_set_mpr(tol, discrete)
    o = Oracle(m, blob)
    o.reset()
    if eps:
        rng = np.random.default_rng(seed)
        o.qpos[:] += eps * rng.standard_normal(m.nv)
    out = np.zeros((nstep, m.nv))
    for i in range(nstep):
        o.ctrl[:] = ctrl_seq[(i // 10) % ctrl_seq.shape[0]]
        o.step(1)
        out[i] = o.qpos
    return out

def rel(a, b):
    r = np.abs(a - b) / np.maximum(np.abs(b), 1e-2)
    return r.max(axis=1)

if __name__ == "__main__":
    nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    for prim in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=prim)
        m = si.model
        blob = engine.make_blob(m, si.key_joint_ids)
        ctrl_seq, _ = bench.load_actions(m)
        print("fingertips:", "capsule" if prim else "hull")
        for disc in ((0, 1) if not prim else (0,)):
            base = traj(m, blob, ctrl_seq, nstep, discrete=disc)
            for eps in (1e-15, 1e-14, 1e-13):
                for seed in range(3):
                    t = traj(m, blob, ctrl_seq, nstep, eps=eps, discrete=disc, seed=seed)
                    r = rel(t, base)
                    print(f"  discrete={disc} eps={eps:g} seed={seed}: max rel {r.max():.2e}; at 100/300/500/700/1000: "
                          + " ".join(f"{r[:k].max():.1e}" for k in (100, 300, 500, 700, nstep)))
            if not prim:
                t = traj(m, blob, ctrl_seq, nstep, tol=0.99e-6, discrete=disc)
                r = rel(t, base)
                print(f"  discrete={disc} tol 0.99e-6 vs 1e-6: max rel {r.max():.2e}; " + " ".join(f"{r[:k].max():.1e}" for k in (100, 300, 500, 700, nstep)))
