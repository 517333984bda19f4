"""RoboPianist suite, vectorised (mirror of robopianist/suite/__init__.py:27-93).

`load()` keeps the reference's signature and adds `n_envs`, `device_id`,
`precision` (64 by default: the reference computes in float64, and only the fp64
engine tracks the CPU path to the teacher-forced 1e-9; 32 selects the fp32 build, which is NOT faster -- it has
no lean solver / split-stage schedule, measured 664 k against 705 k env-steps/s -- and exists for memory-bound batch
sizes).  It returns a batched dm_env-style Environment whose physics is the HIP engine."""

from pathlib import Path
from typing import Any, Dict, Mapping, Optional, Union

from robopianist_amd import music
from robopianist_amd.suite import environment
from robopianist_amd.suite.tasks import piano_with_shadow_hands, self_actuated_piano

_BASE_REPERTOIRE_NAME = "RoboPianist-repertoire-150-{}-v0"
REPERTOIRE_150 = [_BASE_REPERTOIRE_NAME.format(name) for name in music.PIG_MIDIS]
_REPERTOIRE_150_DICT = dict(zip(REPERTOIRE_150, music.PIG_MIDIS))
_BASE_ETUDE_NAME = "RoboPianist-etude-12-{}-v0"
ETUDE_12 = [_BASE_ETUDE_NAME.format(name) for name in music.ETUDE_MIDIS]
_ETUDE_12_DICT = dict(zip(ETUDE_12, music.ETUDE_MIDIS))
_DEBUG_BASE_NAME = "RoboPianist-debug-{}-v0"
DEBUG = [_DEBUG_BASE_NAME.format(name) for name in music.DEBUG_MIDIS]
_DEBUG_DICT = dict(zip(DEBUG, music.DEBUG_MIDIS))

ALL = REPERTOIRE_150 + ETUDE_12 + DEBUG
_ALL_DICT: Dict[str, Union[Path, str]] = {**_REPERTOIRE_150_DICT, **_ETUDE_12_DICT, **_DEBUG_DICT}


def load(
    environment_name: str,
    midi_file: Optional[Path] = None,
    seed: Optional[int] = None,
    stretch: float = 1.0,
    shift: int = 0,
    recompile_physics: bool = False,
    legacy_step: bool = True,
    task_kwargs: Optional[Mapping[str, Any]] = None,
    n_envs: int = 1,
    device_id: int = 0,
    precision: int = 64,
) -> environment.Environment:
    """Loads a (batched) RoboPianist environment; raises ValueError for unknown names."""
    del recompile_physics  # the model is compiled once and uploaded to the GPU
    if midi_file is not None:
        midi = music.load(midi_file, stretch=stretch, shift=shift)
    else:
        if environment_name not in ALL:
            raise ValueError(
                f"Unknown environment {environment_name}. Available environments: {ALL}")
        midi = music.load(_ALL_DICT[environment_name], stretch=stretch, shift=shift)
    task_kwargs = dict(task_kwargs or {})
    task = piano_with_shadow_hands.PianoWithShadowHands(midi=midi, **task_kwargs)
    return environment.Environment(task, n_envs=n_envs, random_state=seed, device_id=device_id,
                                   precision=precision, legacy_step=legacy_step)


__all__ = ["ALL", "DEBUG", "ETUDE_12", "REPERTOIRE_150", "load"]
