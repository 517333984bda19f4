import warnings; warnings.simplefilter('ignore')
import sys; sys.path.insert(0,'.')
import numpy as np, time
from robopianist_amd import engine
from robopianist_amd.model import scene
from bench import load_actions
prec = int(sys.argv[1]) if len(sys.argv)>1 else 32
E = int(sys.argv[2]) if len(sys.argv)>2 else 4096
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=(len(sys.argv) <= 3 or sys.argv[3] != "hull"))
m = si.model
phys = engine.BatchedPhysics(m, si.key_joint_ids, n_envs=E, precision=prec)
ctrl,_ = load_actions(m)
names = {0:'load',1:'actuation',2:'M chol+solve',3:'warmstart',4:'H assembly',5:'H chol+solve',6:'mulM/mulJ/quad',7:'linesearch',8:'update+JT',9:'euler',10:'FK',11:'inertia+CRB',18:'geom centres',12:'col: candidate gen',13:'col: narrow geometry',19:'col: hull pairs (MPR)',24:'col: contact emission',14:'slots+jac',15:'cvel+rne',16:'transm+rows',17:'trace',20:'ts: leaves+rows->LDS',21:'ts: trunk levels',22:'ts: dense gather',23:'col: drain rounds / prefilters',25:'ts: dense chol+solve',5:'ts: back-subst + rest of H solve',32:'ts: chain levels',33:'ts: leader part',34:'ts: trunk gather',35:'mulM0',36:'mulJ'}
for t in range(30):
    phys.set(engine.CTRL, ctrl[t][None,:]); phys.step(10)
phys.profile(True)
n=40
its=[]
for t in range(30,30+n):
    phys.set(engine.CTRL, ctrl[t][None,:]); phys.step(10)
    its.append(phys.get(engine.SOLVER_ITER).mean())
p = phys.profile(False)
cand, passes = p[30], p[31]; p[30]=0; p[31]=0
print('drain rounds/mj_step %.2f' % (p[26]/n/10)); p[26]=0
print('capsule-box cands after prefilter/mj_step %.1f' % (p[27]/n/10)); p[27]=0
print('geom-geom cands/mj_step %.1f key cands %.1f' % (p[28]/n/10, p[29]/n/10)); p[28]=0; p[29]=0
pk = p[48:64].copy(); p[48:64] = 0
tot = p.sum()
print('candidates/mj_step %.1f  narrow passes/mj_step %.2f' % (cand/n/10, passes/n/10))
print('total cycles/env-step (env0): %.0f  -> per mj_step %.0f' % (tot/n, tot/n/10), 'mean newton iters', np.mean([int(v)&255 for v in np.array(its).ravel()]) if False else '', 'ncon mean', phys.get(engine.NCON).mean())
for i in sorted(names, key=lambda i:-p[i]):
    print('%-16s %10.0f cyc/mj_step  %5.1f%%' % (names[i], p[i]/n/10, 100*p[i]/tot))
if pk[8:15].sum() > 0:
    print('pooled narrow phase, per list (cc, cb, bb, hull buckets 3..6): chunks per launch / avg cycles per chunk')
    for t in range(7):
        if pk[8 + t]: print('  list %d: %8.1f chunks/launch  %9.0f cycles/chunk' % (t, pk[8 + t] / n / 10, pk[t] / pk[8 + t]))
