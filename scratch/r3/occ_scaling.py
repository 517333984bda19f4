"""How the two-waves-per-SIMD kernels scale with the number of resident waves: lockstep replay at E = 1024 (one wave
per SIMD), 2048 (two), 4096 (two rounds of two) envs; solver / step-sequence time from the engine's HIP events."""
import warnings; warnings.simplefilter('ignore')
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from robopianist_amd import engine
from robopianist_amd.model import scene
from bench import load_actions
ft = sys.argv[1] if len(sys.argv) > 1 else "primitive"
si = scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=(ft == "primitive"))
ctrl, _ = load_actions(si.model)
for E in (512, 1024, 2048, 4096, 8192):
    p = engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=E, precision=64, self_check=False)
    p.set_stream_slices(1)
    for t in range(40):
        p.set(engine.CTRL, ctrl[t][None, :]); p.step(10)
    p.sync(); p.solver_kernel_time(); p.kernel_time()
    t0 = time.perf_counter()
    for t in range(40, 100):
        p.set(engine.CTRL, ctrl[t][None, :]); p.step(10)
    p.sync(); dt = time.perf_counter() - t0
    sms, _ = p.solver_kernel_time(); kms, _ = p.kernel_time()
    print(f"{ft} E={E}: {E*60/dt:9.0f} env-steps/s, step sequence {kms:.3f} ms, solver launch {sms:.4f} ms, position ~{(kms-10*sms)/11:.4f} ms")
