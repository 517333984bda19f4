import warnings; warnings.simplefilter('ignore')
import sys, os; sys.path.insert(0,'.')
import numpy as np
from robopianist_amd.model import scene
from robopianist_amd import engine
from oracle.rp_oracle import Oracle
from bench import load_actions
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=True)
m = si.model
o = Oracle(m, engine.make_blob(m, si.key_joint_ids))
ctrl,_ = load_actions(m)
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for th in (1, 8, 32, 64, 128, 256):
    n = max(th, 8)
    secs,_ = o.bench(n, 300, np.tile(ctrl[40],(n,1)), th)
    print(th, 'threads: %.0f mj_steps/s total, %.0f per thread' % (n*300/secs, n*300/secs/th))
