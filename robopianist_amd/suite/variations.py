"""MIDI augmentations applied at episode boundaries.

Mirror of robopianist/suite/variations.py:27-184: callables with dm_control's
`Variation` calling convention `var(initial_value=, current_value=, random_state=)`
that map a MidiFile to a (possibly) different MidiFile.  Draw order on the numpy
RandomState is the reference's (one `uniform(0, 1)` gate, then the parameter draw), so a
seeded run selects the same augmentations.

The vectorised tasks call them once per env whose episode starts (host side; the goal
tables of that env are then regenerated and uploaded, see
`PianoWithShadowHands._maybe_change_midi`).
"""

from __future__ import annotations

from typing import Sequence

import numpy as np

from robopianist_amd import music
from robopianist_amd.music import constants, midi_file


class Variation:
    """Stand-in for dm_control.composer.variation.Variation (the calling convention only)."""

    def __call__(self, initial_value=None, current_value=None, random_state=None):
        raise NotImplementedError


def _require_midi(initial_value) -> midi_file.MidiFile:
    if initial_value is None or not isinstance(initial_value, midi_file.MidiFile):
        raise ValueError("Expected `initial_value` to be provided and be a midi_file.MidiFile.")
    return initial_value


class MidiSelect(Variation):
    """Uniformly picks one of `midi_names` (keys of `music.load`)  (variations.py:27-46)."""

    def __init__(self, midi_names: Sequence[str] = ()) -> None:
        self._midi_names = list(midi_names)

    def __call__(self, initial_value=None, current_value=None, random_state=None) -> midi_file.MidiFile:
        del initial_value, current_value
        random = random_state or np.random
        return music.load(str(random.choice(self._midi_names)))


class MidiTemporalStretch(Variation):
    """With probability `prob`, stretches time by a factor drawn from
    U[1 - stretch_range, 1 + stretch_range]  (variations.py:49-83)."""

    def __init__(self, prob: float, stretch_range: float) -> None:
        self._prob = prob
        self._stretch_range = stretch_range

    def __call__(self, initial_value=None, current_value=None, random_state=None) -> midi_file.MidiFile:
        del current_value
        random = random_state or np.random
        gate = random.uniform(0.0, 1.0)
        midi = _require_midi(initial_value)
        if gate > self._prob:
            return midi
        factor = 1.0 + random.uniform(-self._stretch_range, self._stretch_range)
        return midi.stretch(factor)


def _shift_bounds(midi: midi_file.MidiFile, max_semitones: int):
    """Largest downward / upward shift that keeps every note on the 88 keys."""
    min_pitch, max_pitch = midi.pitch_range()
    low = max(constants.MIN_MIDI_PITCH_PIANO - min_pitch, -max_semitones)
    high = min(constants.MAX_MIDI_PITCH_PIANO - max_pitch, max_semitones)
    return low, high


class MidiPitchShift(Variation):
    """With probability `prob`, transposes by an integer number of semitones drawn
    uniformly from [-shift_range, shift_range], truncated to the piano's range
    (variations.py:86-133)."""

    def __init__(self, prob: float, shift_range: int) -> None:
        self._prob = prob
        if not isinstance(shift_range, int):
            raise ValueError("`shift_range` must be an integer.")
        self._shift_range = shift_range

    def __call__(self, initial_value=None, current_value=None, random_state=None) -> midi_file.MidiFile:
        del current_value
        random = random_state or np.random
        gate = random.uniform(0.0, 1.0)
        midi = _require_midi(initial_value)
        if gate > self._prob or self._shift_range == 0:
            return midi
        low, high = _shift_bounds(midi, self._shift_range)
        shift = random.randint(low, high + 1)
        return midi if shift == 0 else midi.transpose(int(shift))


class MidiOctaveShift(Variation):
    """As MidiPitchShift, in whole octaves  (variations.py:136-184)."""

    def __init__(self, prob: float, octave_range: int) -> None:
        self._prob = prob
        if not isinstance(octave_range, int):
            raise ValueError("`octave_range` must be an integer.")
        self._octave_range = octave_range

    def __call__(self, initial_value=None, current_value=None, random_state=None) -> midi_file.MidiFile:
        del current_value
        random = random_state or np.random
        gate = random.uniform(0.0, 1.0)
        midi = _require_midi(initial_value)
        if gate > self._prob or self._octave_range == 0:
            return midi
        low, high = _shift_bounds(midi, self._octave_range * 12)
        shift = random.randint(low // 12, high // 12 + 1)
        return midi if shift == 0 else midi.transpose(int(shift) * 12)
