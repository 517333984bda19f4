"""Vectorised tasks (mirror of robopianist/suite/tasks/__init__.py:15-27)."""

from robopianist_amd.suite.tasks.piano_with_one_shadow_hand import PianoWithOneShadowHand
from robopianist_amd.suite.tasks.piano_with_shadow_hands import PianoWithShadowHands
from robopianist_amd.suite.tasks.self_actuated_piano import SelfActuatedPiano

__all__ = ["PianoWithOneShadowHand", "PianoWithShadowHands", "SelfActuatedPiano"]
