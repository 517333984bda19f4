"""env-steps/s on one MI355X for the BASELINE.json configs other than the headline one."""
import warnings; warnings.simplefilter('ignore')
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from robopianist_amd import music, suite
from robopianist_amd.suite import variations
from robopianist_amd.wrappers import CanonicalSpecWrapper

KW = dict(control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True)

def rate(env, E, steps=150, warm=15):
    env.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(12345)
    A = env.action_spec().shape[0]
    for t in range(warm + steps):
        if t == warm:
            torch.cuda.synchronize(); t0 = time.time()
        env.step(torch.rand((E, A), generator=g, device='cuda', dtype=torch.float64) * 2 - 1)
    torch.cuda.synchronize()
    w = env.physics.warn
    return E * steps / (time.time() - t0), int(((w & 1) != 0).sum())

E = 4096
env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=12345, n_envs=E,
                                      task_kwargs=dict(trim_silence=True, **KW)))
print("config 3 (Twinkle, random policy, %d envs): %.0f env-steps/s, bad envs %d" % ((E,) + rate(env, E)))
del env
E = 8192
env = CanonicalSpecWrapper(suite.load("RoboPianist-debug-CMajorScaleTwoHands-v0", seed=12345, n_envs=E, task_kwargs=dict(**KW)))
print("config 4 (CMajorScaleTwoHands, random policy, %d envs): %.0f env-steps/s, bad envs %d" % ((E,) + rate(env, E)))
del env
E = 2048
from robopianist_amd.suite.tasks import PianoWithShadowHands
from robopianist_amd.suite import environment
songs = [music.load(n) for n in music.ALL]
rs = np.random.RandomState(0)
bank = list(songs)
aug = [variations.MidiTemporalStretch(1.0, 0.2), variations.MidiPitchShift(1.0, 5)]
while len(bank) < 150:   # PIG repertoire is licence-gated: in-tree songs + stretch / shift variants
    m = songs[len(bank) % len(songs)]
    for v in aug: m = v(initial_value=m, random_state=rs)
    bank.append(m)
task = PianoWithShadowHands(midi=bank, **KW)
env = CanonicalSpecWrapper(environment.Environment(task, n_envs=E, random_state=1))
print("config 5 (150 distinct goal tables, random policy, %d envs): %.0f env-steps/s, bad envs %d; song lengths %d..%d"
      % ((E,) + rate(env, E) + (int(task._song_len.min()), int(task._song_len.max()))))
