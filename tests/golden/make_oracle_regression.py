"""Regression fixture of the CPU oracle ITSELF (not of MuJoCo -- parity stays unpinned): the oracle's own
trajectory on the scripted Twinkle replay, hull and capsule fingertips, at a few mj_steps inside the window where the
trajectory is still smooth (the first 300), plus the contact counts along the way.  A change to the
oracle or to the model builders that alters the physics of the benchmark scene shows up here.

    python tests/golden/make_oracle_regression.py        # rewrites tests/golden/oracle_regression.npz
"""
import os, sys, warnings
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

MARKS = (1, 10, 50, 100, 200, 300)


def rollout(primitive: bool):
    from robopianist_amd.model import scene
    from robopianist_amd import engine
    from oracle.rp_oracle import Oracle
    import bench
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=primitive)
    m = si.model
    o = Oracle(m, engine.make_blob(m, si.key_joint_ids)); o.reset()
    ctrl, _ = bench.load_actions(m)
    q, ncon = [], []
    for i in range(max(MARKS)):
        o.ctrl[:] = ctrl[i // 10]
        o.step(1)
        ncon.append(o.ncon)
        if i + 1 in MARKS:
            q.append(o.qpos.copy())
    return np.array(q), np.array(ncon, np.int32)


if __name__ == "__main__":
    out = {"marks": np.array(MARKS, np.int32)}
    for name, prim in (("hull", False), ("capsule", True)):
        q, n = rollout(prim)
        out[f"qpos_{name}"] = q; out[f"ncon_{name}"] = n
        print(name, "contacts over 300 mj_steps:", int(n.sum()), "max", int(n.max()))
    np.savez_compressed(os.path.join(HERE, "oracle_regression.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_regression.npz"))
