"""Throughput of the env loop with per-episode MIDI augmentations (host regeneration of the
goal tables of every env that starts an episode) vs the same loop without them."""
import sys, time, warnings
warnings.simplefilter("ignore")
sys.path.insert(0, ".")
import numpy as np, torch
from robopianist_amd import suite
from robopianist_amd.suite import variations
from robopianist_amd.wrappers import CanonicalSpecWrapper

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
actions = np.load("tests/golden/twinkle_twinkle_actions.npy")
for aug in (False, True, "prefetch"):
    augs = [variations.MidiTemporalStretch(prob=1.0, stretch_range=0.2),
            variations.MidiPitchShift(prob=1.0, shift_range=3)] if aug else None
    env = CanonicalSpecWrapper(suite.load(
        "RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=1, n_envs=E, precision=64,
        task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                         primitive_fingertip_collisions=True, augmentations=augs,
                         augmentation_prefetch=(aug == "prefetch"))))
    t0 = time.time(); env.reset(); torch.cuda.synchronize(); t_reset = time.time() - t0
    a = torch.as_tensor(actions, device=env.physics.device, dtype=torch.float64)
    n = 600
    torch.cuda.synchronize(); t0 = time.time()
    firsts = 0
    for t in range(n):
        ts = env.step(a[t % len(a)].expand(E, -1))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"augmentations={aug}: full reset {t_reset:.2f} s, {n} steps of {E} envs in {dt:.2f} s "
          f"-> {E * n / dt:,.0f} env-steps/s, song lengths {int(env.task._song_len.min())}..{int(env.task._song_len.max())}"
          + (f", refills {env.task.prefetch_refills}" if aug == "prefetch" else ""))
