"""Batched, headless counterpart of the reference's examples/self_actuated_piano_env.py: the
self-actuated piano played by the oracle policy of that example (ctrl = ctrlrange max on the
goal keys, min elsewhere; last action entry = goal sustain), which reaches F1 = 1."""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robopianist_amd import music  # noqa: E402
from robopianist_amd.suite import environment  # noqa: E402
from robopianist_amd.suite.tasks import SelfActuatedPiano  # noqa: E402
from robopianist_amd.wrappers import MidiEvaluationWrapper  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--midi", default="TwinkleTwinkleLittleStar", help="library name or .mid path")
    ap.add_argument("--control_timestep", type=float, default=0.05)
    ap.add_argument("--n_envs", type=int, default=16)
    ap.add_argument("--precision", type=int, default=64, choices=(32, 64))
    args = ap.parse_args()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        task = SelfActuatedPiano(midi=music.load(args.midi), n_steps_lookahead=1,
                                 control_timestep=args.control_timestep, change_color_on_activation=True)
        env = MidiEvaluationWrapper(environment.Environment(task, n_envs=args.n_envs, precision=args.precision))
    spec = env.action_spec()
    lo = torch.as_tensor(spec.minimum, device=env.physics.device)
    hi = torch.as_tensor(spec.maximum, device=env.physics.device)
    timestep = env.reset()
    n, ret = 0, 0.0
    while True:
        goal = timestep.observation["goal"][:, :89]           # current goal: 88 keys + sustain
        action = torch.where(goal > 0, hi, lo)
        timestep = env.step(action)
        ret += float(timestep.reward.mean())
        n += 1
        if bool(timestep.last().all()):
            break
    print(f"{n} control steps x {args.n_envs} envs, mean return {ret:.3f}")
    for k, v in env.get_musical_metrics().items():
        print(f"\t{k}: {v:.4f}")


if __name__ == "__main__":
    main()
