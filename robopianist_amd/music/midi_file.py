"""MIDI abstractions: MidiFile, PianoNote, NoteTrajectory.

Mirror of robopianist/music/midi_file.py (same names, argument meaning and error
behaviour) on top of `sequence.NoteSequence` instead of note_seq protos.
Synthesis / playback (FluidSynth, PyAudio) is out of scope.
"""

from __future__ import annotations

import dataclasses
from pathlib import Path
from typing import List, Tuple, Union

import numpy as np

from robopianist_amd.music import constants as consts
from robopianist_amd.music import sequence as seqlib
from robopianist_amd.music.piano_roll import sequence_to_pianoroll
from robopianist_amd.music.sequence import NoteSequence

MIN_MIDI_VELOCITY = 1  # note_seq.constants
MAX_MIDI_VELOCITY = 127


def note_name_to_midi_number(name: str) -> int:
    return consts.NOTE_NAME_TO_MIDI_NUMBER[name]


def midi_number_to_note_name(number: int) -> str:
    return consts.MIDI_NUMBER_TO_NOTE_NAME[number]


def key_number_to_midi_number(key_number: int) -> int:
    if not 0 <= key_number < consts.NUM_KEYS:
        raise ValueError(f"Key number should be in [0 {consts.NUM_KEYS}], got {key_number}.")
    return key_number + consts.MIN_MIDI_PITCH_PIANO


def midi_number_to_key_number(midi_number: int) -> int:
    if not consts.MIN_MIDI_PITCH_PIANO <= midi_number <= consts.MAX_MIDI_PITCH_PIANO:
        raise ValueError(
            f"MIDI pitch number should be in [{consts.MIN_MIDI_PITCH_PIANO}, "
            f"{consts.MAX_MIDI_PITCH_PIANO}], got {midi_number}.")
    return midi_number - consts.MIN_MIDI_PITCH_PIANO


def key_number_to_note_name(key_number: int) -> str:
    return consts.KEY_NUMBER_TO_NOTE_NAME[key_number]


def note_name_to_key_number(note_name: str) -> int:
    return consts.NOTE_NAME_TO_KEY_NUMBER[note_name]


@dataclasses.dataclass(frozen=True)
class PianoNote:
    """midi_file.py:124-168."""

    number: int
    velocity: int
    key: int
    name: str
    fingering: int = -1

    @staticmethod
    def create(number: int, velocity: int, fingering: int = -1) -> "PianoNote":
        if not MIN_MIDI_VELOCITY <= velocity <= MAX_MIDI_VELOCITY:
            raise ValueError(
                f"Velocity should be in [{MIN_MIDI_VELOCITY}, {MAX_MIDI_VELOCITY}], got {velocity}.")
        if not consts.MIN_MIDI_PITCH_PIANO <= number <= consts.MAX_MIDI_PITCH_PIANO:
            raise ValueError(
                f"MIDI pitch number should be in [{consts.MIN_MIDI_PITCH_PIANO}, "
                f"{consts.MAX_MIDI_PITCH_PIANO}], got {number}.")
        return PianoNote(number=number, velocity=velocity, key=midi_number_to_key_number(number),
                         name=midi_number_to_note_name(number), fingering=fingering)


@dataclasses.dataclass(frozen=True)
class MidiFile:
    """midi_file.py:171-286 (synthesize/play omitted)."""

    seq: NoteSequence

    @classmethod
    def from_file(cls, filename: Union[str, Path]) -> "MidiFile":
        filename = Path(filename)
        if filename.suffix == ".mid":
            try:
                seq = seqlib.read_midi_file(filename)
            except (RuntimeError, IndexError, KeyError) as e:
                raise RuntimeError(f"Could not parse MIDI file {filename}.") from e
        elif filename.suffix == ".proto":
            raise ValueError("NoteSequence .proto files need note_seq, which is unavailable.")
        else:
            raise ValueError(f"Unsupported file extension {filename.suffix}.")
        return cls(seq=seq)

    def stretch(self, factor: float) -> "MidiFile":
        if factor <= 0:
            raise ValueError("factor must be positive.")
        return MidiFile(seq=seqlib.stretch_note_sequence(self.seq, factor))

    def transpose(self, amount: int, transpose_chords: bool = True) -> "MidiFile":
        del transpose_chords
        seq, _ = seqlib.transpose_note_sequence(
            self.seq, amount=amount, min_allowed_pitch=consts.MIN_MIDI_PITCH_PIANO,
            max_allowed_pitch=consts.MAX_MIDI_PITCH_PIANO)
        return MidiFile(seq=seq)

    def trim_silence(self) -> "MidiFile":
        seq = seqlib.extract_subsequence(
            self.seq, start_time=self.seq.notes[0].start_time,
            end_time=self.seq.notes[-1].end_time)
        return MidiFile(seq=seq)

    def has_fingering(self) -> bool:
        fingerings = set(note.part for note in self.seq.notes)
        non_zero = [f for f in fingerings if f != 0]
        return len(fingerings) > 1 and len(non_zero) > 0

    @property
    def duration(self) -> float:
        return self.seq.total_time

    @property
    def n_notes(self) -> int:
        return len(self.seq.notes)

    @property
    def title(self) -> str:
        return self.seq.sequence_metadata.title

    @property
    def artist(self) -> str:
        return self.seq.sequence_metadata.artist


@dataclasses.dataclass
class NoteTrajectory:
    """midi_file.py:289-414."""

    dt: float
    notes: List[List[PianoNote]]
    sustains: List[int]

    def __post_init__(self) -> None:
        if self.dt <= 0:
            raise ValueError("dt must be positive.")
        if len(self.notes) != len(self.sustains):
            raise ValueError("notes and sustains must have the same length.")

    @classmethod
    def from_midi(cls, midi: MidiFile, dt: float) -> "NoteTrajectory":
        notes, sustains = NoteTrajectory.seq_to_trajectory(midi.seq, dt)
        return cls(dt=dt, notes=notes, sustains=sustains)

    @staticmethod
    def seq_to_trajectory(seq: NoteSequence, dt: float) -> Tuple[List[List[PianoNote]], List[int]]:
        piano_roll = sequence_to_pianoroll(
            seq, frames_per_second=1 / dt, min_pitch=consts.MIN_MIDI_PITCH,
            max_pitch=consts.MAX_MIDI_PITCH, onset_window=0)
        notes: List[List[PianoNote]] = []
        for t, timestep in enumerate(piano_roll.active_velocities):
            notes_in_timestep: List[PianoNote] = []
            for index in np.nonzero(timestep)[0]:
                if (t > 0 and piano_roll.active_velocities[t - 1][index]
                        and piano_roll.onset_velocities[t][index]):
                    # repeated note: force a release between consecutive presses
                    continue
                velocity = int(round(timestep[index] * consts.MAX_VELOCITY))
                fingering = int(piano_roll.fingerings[t, index])
                notes_in_timestep.append(PianoNote.create(int(index), velocity, fingering))
            notes.append(notes_in_timestep)
        sustains: List[int] = []
        prev_sustain = 0
        for timestep in piano_roll.control_changes:
            event = timestep[consts.SUSTAIN_PEDAL_CC_NUMBER]
            if 1 <= event <= consts.SUSTAIN_PEDAL_CC_NUMBER:
                sustain = 0
            elif consts.SUSTAIN_PEDAL_CC_NUMBER + 1 <= event <= consts.MAX_CC_VALUE + 1:
                sustain = 1
            else:
                sustain = prev_sustain
            sustains.append(sustain)
            prev_sustain = sustain
        return notes, sustains

    def __len__(self) -> int:
        return len(self.notes)

    def add_initial_buffer_time(self, initial_buffer_time: float) -> "NoteTrajectory":
        if initial_buffer_time < 0.0:
            raise ValueError("initial_buffer_time must be non-negative.")
        for _ in range(int(round(initial_buffer_time / self.dt))):
            self.notes.insert(0, [])
            self.sustains.insert(0, 0)
        return self

    def to_piano_roll(self) -> np.ndarray:
        frames = np.zeros((len(self.notes), consts.MAX_MIDI_PITCH), dtype=np.int32)
        for t, timestep in enumerate(self.notes):
            for note in timestep:
                frames[t, note.number] = 1
        return frames

    # ---- batched-engine view -------------------------------------------------
    def to_goal_tables(self):
        """Dense tables consumed by the vectorised env: goal[T, 89] (keys + sustain),
        finger[T, 88] (fingering id of each goal key or -1)."""
        T = len(self.notes)
        goal = np.zeros((T, consts.NUM_KEYS + 1), dtype=np.float32)
        finger = np.full((T, consts.NUM_KEYS), -1, dtype=np.int32)
        for t, timestep in enumerate(self.notes):
            for note in timestep:
                goal[t, note.key] = 1.0
                finger[t, note.key] = note.fingering
            goal[t, -1] = self.sustains[t]
        return goal, finger
