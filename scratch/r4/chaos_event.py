import sys, os, warnings
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from chaos_control import *
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=False)
m = si.model
blob = engine.make_blob(m, si.key_joint_ids)
ctrl_seq, _ = bench.load_actions(m)
def run(eps):
    o = Oracle(m, blob); o.reset()
    if eps: o.qpos[:] += eps * np.random.default_rng(0).standard_normal(m.nv)
    rec = []
    for i in range(436):
        o.ctrl[:] = ctrl_seq[(i // 10) % ctrl_seq.shape[0]]
        o.step(1)
        if i >= 424:
            c = o.contact.reshape(-1, 16).copy()
            rec.append((i, o.qpos.copy(), c, o.solver_iter))
    return rec
a = run(0); b = run(1e-15)
names = m.names["geom"]
for (i, qa, ca, ia), (_, qb, cb, ib) in zip(a, b):
    print("step", i, "dq max", np.abs(qa-qb).max(), "ncon", len(ca), len(cb), "iters", ia, ib)
    if len(ca) == len(cb):
        for x, y in zip(ca, cb):
            d = np.abs(x - y).max()
            if d > 1e-9:
                print("   ", names[int(x[13])], names[int(x[14])], "dist", x[0], y[0], "dn", np.abs(x[4:7]-y[4:7]).max(), "dpos", np.abs(x[1:4]-y[1:4]).max())
    else:
        print("   A:", [(names[int(x[13])], names[int(x[14])]) for x in ca])
        print("   B:", [(names[int(x[13])], names[int(x[14])]) for x in cb])
