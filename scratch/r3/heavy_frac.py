"""Share of envs outside the light capacity class in the staggered config-2 replay, and why."""
import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0, '.')
import numpy as np
from robopianist_amd import engine
from robopianist_amd.model import scene
from bench import load_actions
hull = len(sys.argv) > 1 and sys.argv[1] == 'hull'
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=not hull)
m = si.model
E = 4096
phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E, precision=64)
ctrl, _ = load_actions(m)
T = len(ctrl)
off = np.arange(E) % T
heavy = []; why = np.zeros(5)
mx = np.zeros(5)
for t in range(160):
    phys.set(engine.CTRL, ctrl[(off + t) % T])
    phys.step(10)
    if t >= 20 and t % 4 == 0:
        h = phys.get(engine.DEBUG_HANDOVER_HDR)
        nd = np.array([bin((int(a) & 0xffffffff) | ((int(b) & 0xffffffff) << 32)).count('1') for a, b in zip(h[:, 2], h[:, 3])])
        hv = h[:, 6] == 0
        heavy.append(hv.mean())
        why += [(h[:, 0] > 24).sum(), (h[:, 4] > 160).sum(), (nd > 36).sum(), (h[:, 1] > 12).sum(), hv.sum()]
        mx = np.maximum(mx, [h[:, 0].max(), h[:, 4].max(), nd.max(), h[:, 1].max(), 0])
        if t == 40:
            print('ncon pct 50/90/99/max', np.percentile(h[:, 0], [50, 90, 99, 100]), 'nent', np.percentile(h[:, 4], [50, 90, 99, 100]), 'nd', np.percentile(nd, [50, 90, 99, 100]), 'nkt', np.percentile(h[:, 1], [50, 90, 99, 100]))
print('heavy share mean %.4f max %.4f' % (np.mean(heavy), np.max(heavy)))
print('reasons (ncon>24, nent>160, nd>36, nkt>12, total heavy) per sample:', why / len(heavy), 'max seen', mx)
