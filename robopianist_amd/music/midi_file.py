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


class NoteArrays:
    """Flat numpy view of a NoteSequence (notes in list order, control changes in list
    order): what the vectorised goal-table builder consumes.  Time stretches and
    transpositions act on it with the same IEEE operations as
    sequence.stretch_note_sequence / transpose_note_sequence."""

    __slots__ = ("start", "end", "pitch", "velocity", "part", "cc_time", "cc_num", "cc_val", "total_time")

    @classmethod
    def from_sequence(cls, seq: NoteSequence) -> "NoteArrays":
        a = cls()
        n = seq.notes
        a.start = np.array([x.start_time for x in n], np.float64)
        a.end = np.array([x.end_time for x in n], np.float64)
        a.pitch = np.array([x.pitch for x in n], np.int64)
        a.velocity = np.array([x.velocity for x in n], np.int64)
        a.part = np.array([x.part for x in n], np.int64)
        c = seq.control_changes
        a.cc_time = np.array([x.time for x in c], np.float64)
        a.cc_num = np.array([x.control_number for x in c], np.int64)
        a.cc_val = np.array([x.control_value for x in c], np.int64)
        a.total_time = float(seq.total_time)
        return a

    def stretched(self, factor: float) -> "NoteArrays":
        if factor == 1.0:
            return self
        a = NoteArrays()
        a.start, a.end, a.cc_time = self.start * factor, self.end * factor, self.cc_time * factor
        a.pitch, a.velocity, a.part, a.cc_num, a.cc_val = self.pitch, self.velocity, self.part, self.cc_num, self.cc_val
        a.total_time = self.total_time * factor
        return a

    def transposed(self, amount: int) -> "NoteArrays":
        a = NoteArrays()
        pitch = self.pitch + amount
        keep = (pitch >= consts.MIN_MIDI_PITCH_PIANO) & (pitch <= consts.MAX_MIDI_PITCH_PIANO)
        a.start, a.end, a.pitch = self.start[keep], self.end[keep], pitch[keep]
        a.velocity, a.part = self.velocity[keep], self.part[keep]
        a.cc_time, a.cc_num, a.cc_val, a.total_time = self.cc_time, self.cc_num, self.cc_val, self.total_time
        return a


class MidiFile:
    """midi_file.py:171-286 (synthesize/play omitted).

    `stretch` / `transpose` are recorded as pending operations on the source sequence:
    `.seq` materialises them (sequence.stretch_note_sequence / transpose_note_sequence, in
    order) on first access, `note_arrays()` applies them to the flat numpy view instead,
    which is what the per-episode MIDI augmentations of the vectorised tasks use.  The
    source sequence must not be edited once either view has been requested."""

    def __init__(self, seq: NoteSequence = None, *, _base: NoteSequence = None, _ops: tuple = ()):
        if (seq is None) == (_base is None):
            raise TypeError("MidiFile(seq=NoteSequence)")
        self._base = seq if _base is None else _base
        self._ops = tuple(_ops)
        self._seq = seq
        self._arrays = None

    def __repr__(self) -> str:
        return f"MidiFile(title={self.title!r}, n_notes={self.n_notes}, duration={self.duration:.3f})"

    @property
    def seq(self) -> NoteSequence:
        if self._seq is None:
            seq = self._base
            for op, arg in self._ops:
                if op == "stretch":
                    seq = seqlib.stretch_note_sequence(seq, arg)
                else:
                    seq, _ = seqlib.transpose_note_sequence(
                        seq, amount=arg, min_allowed_pitch=consts.MIN_MIDI_PITCH_PIANO,
                        max_allowed_pitch=consts.MAX_MIDI_PITCH_PIANO)
            self._seq = seq
        return self._seq

    def note_arrays(self) -> NoteArrays:
        if self._arrays is None:
            if self._seq is not None and self._ops:
                a = NoteArrays.from_sequence(self._seq)  # already materialised: take it as is
            else:
                cache = getattr(self._base, "_note_arrays_cache", None)
                if cache is None or len(cache.pitch) != len(self._base.notes):
                    cache = NoteArrays.from_sequence(self._base)
                    self._base._note_arrays_cache = cache
                a = cache
                for op, arg in self._ops:
                    a = a.stretched(arg) if op == "stretch" else a.transposed(arg)
            self._arrays = a
        return self._arrays

    @classmethod
    def from_file(cls, filename: Union[str, Path]) -> "MidiFile":
        filename = Path(filename)
        if filename.suffix == ".mid":
            try:
                seq = seqlib.read_midi_file(filename)
            except (RuntimeError, IndexError, KeyError) as e:
                raise RuntimeError(f"Could not parse MIDI file {filename}.") from e
        elif filename.suffix == ".proto":
            # serialized note_seq NoteSequence (the PIG repertoire; fingering in note.part)
            from robopianist_amd.music import note_seq_proto
            with open(filename, "rb") as f:
                try:
                    seq = note_seq_proto.parse(f.read())
                except ValueError as e:
                    raise RuntimeError(f"Could not parse NoteSequence file {filename}.") from e
        else:
            raise ValueError(f"Unsupported file extension {filename.suffix}.")
        return cls(seq=seq)

    def save(self, filename: Union[str, Path]) -> None:
        """Saves the song as a serialized NoteSequence (`.proto`); midi_file.py:191-201.  Writing
        Standard MIDI files is not implemented (nothing on the path consumes them)."""
        filename = Path(filename)
        if filename.suffix == ".proto":
            from robopianist_amd.music import note_seq_proto
            with open(filename, "wb") as f:
                f.write(note_seq_proto.serialize(self.seq))
        elif filename.suffix == ".mid":
            raise NotImplementedError("writing .mid files is not supported; save as .proto")
        else:
            raise ValueError(f"Unsupported file extension {filename.suffix}.")

    def stretch(self, factor: float) -> "MidiFile":
        if factor <= 0:
            raise ValueError("factor must be positive.")
        return MidiFile(_base=self._base, _ops=self._ops + (("stretch", float(factor)),))

    def transpose(self, amount: int, transpose_chords: bool = True) -> "MidiFile":
        del transpose_chords
        return MidiFile(_base=self._base, _ops=self._ops + (("transpose", int(amount)),))

    def trim_silence(self) -> "MidiFile":
        seq = seqlib.extract_subsequence(
            self.seq, start_time=self.seq.notes[0].start_time,
            end_time=self.seq.notes[-1].end_time)
        return MidiFile(seq=seq)

    def has_fingering(self) -> bool:
        fingerings = set(note.part for note in self._base.notes)  # ops never touch `part`
        non_zero = [f for f in fingerings if f != 0]
        return len(fingerings) > 1 and len(non_zero) > 0

    def pitch_range(self) -> Tuple[int, int]:
        """(lowest, highest) MIDI pitch of the notes."""
        p = self.note_arrays().pitch
        return int(p.min()), int(p.max())

    @property
    def duration(self) -> float:
        return self.note_arrays().total_time if self._seq is None else self._seq.total_time

    @property
    def n_notes(self) -> int:
        return len(self.note_arrays().pitch) if self._seq is None else len(self._seq.notes)

    @property
    def title(self) -> str:
        return self._base.sequence_metadata.title

    @property
    def artist(self) -> str:
        return self._base.sequence_metadata.artist


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
        act = np.asarray(piano_roll.active_velocities)
        ons = np.asarray(piano_roll.onset_velocities)
        # a key that was already down in the previous frame and is struck again in this one is left out of this frame:
        # consecutive presses of one key need a release in between (reference: music/midi_file.py:335-345)
        held = np.zeros_like(act, dtype=bool)
        held[1:] = (act[:-1] != 0) & (ons[1:] != 0)
        keep = (act != 0) & ~held
        vel = np.rint(act * consts.MAX_VELOCITY).astype(np.int64)
        fing = np.asarray(piano_roll.fingerings).astype(np.int64)
        notes: List[List[PianoNote]] = [
            [PianoNote.create(int(k), int(vel[t, k]), int(fing[t, k])) for k in np.flatnonzero(keep[t])]
            for t in range(act.shape[0])]
        # sustain pedal per frame: control value + 1 in 1..64 releases it, 65..128 presses it, 0 (no event in the
        # frame) keeps the last state -- a forward fill of the frames that carry an event
        ev = np.asarray(piano_roll.control_changes)[:, consts.SUSTAIN_PEDAL_CC_NUMBER].astype(np.int64)
        has = (ev >= 1) & (ev <= consts.MAX_CC_VALUE + 1)
        state = (ev > consts.SUSTAIN_PEDAL_CC_NUMBER).astype(np.int64)
        last = np.maximum.accumulate(np.where(has, np.arange(len(ev)), -1))
        sustains: List[int] = [int(v) for v in np.where(last >= 0, state[np.maximum(last, 0)], 0)]
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
    @staticmethod
    def goal_tables_from_arrays(arrays: "NoteArrays", dt: float, initial_buffer_time: float = 0.0):
        """`from_midi(midi, dt).add_initial_buffer_time(b).to_goal_tables()` computed from
        the flat note arrays (same frame arithmetic as piano_roll.sequence_to_pianoroll and
        seq_to_trajectory above, incl. the repeated-note gap and the sustain latch), without
        building note objects: goal[T, 89] float32, finger[T, 88] int32.  Returns None when
        the generic path has to decide (it raises for notes off the 88 keys)."""
        if initial_buffer_time < 0.0:
            raise ValueError("initial_buffer_time must be non-negative.")
        if dt <= 0:
            raise ValueError("dt must be positive.")
        a = arrays
        if len(a.pitch) and (a.pitch.min() < consts.MIN_MIDI_PITCH_PIANO or a.pitch.max() > consts.MAX_MIDI_PITCH_PIANO
                             or a.velocity.max() > consts.MAX_VELOCITY):
            return None
        fps = 1 / dt
        T = int(a.total_time * fps + 1)
        nk = consts.NUM_KEYS
        vel = np.zeros((T, nk), np.float32)
        onset = np.zeros((T, nk), bool)
        fing = np.full((T, nk), -1, np.int32)
        s_frame = (a.start * fps).astype(np.int64)               # int() truncation
        e_frame = np.maximum(s_frame + 1, np.ceil(a.end * fps).astype(np.int64))
        key = a.pitch - consts.MIN_MIDI_PITCH_PIANO
        v32 = a.velocity.astype(np.float64) / consts.MAX_VELOCITY
        # rasterise all notes at once: one (frame, key) index pair per covered frame, notes in
        # start order -- NumPy assigns repeated indices in order, so later notes overwrite
        # earlier ones exactly like the per-note slice assignments of the piano roll
        order = np.argsort(a.start, kind="stable")
        s_o = np.minimum(s_frame[order], T)
        e_o = np.minimum(e_frame[order], T)
        n_fr = np.maximum(e_o - s_o, 0)
        total = int(n_fr.sum())
        if total:
            note = np.repeat(np.arange(len(order)), n_fr)
            first = np.cumsum(n_fr) - n_fr
            frame = np.arange(total) - first[note] + s_o[note]
            kcol = key[order][note]
            vel[frame, kcol] = v32[order][note].astype(np.float32)
            fing[frame, kcol] = a.part[order][note].astype(np.int32)
        starts_in = s_frame[order] < T
        onset[s_frame[order][starts_in], key[order][starts_in]] = True
        active = vel != 0
        repeated = np.zeros_like(active)
        repeated[1:] = active[:-1] & active[1:] & onset[1:]      # seq_to_trajectory's `continue`
        on = active & ~repeated
        goal = np.zeros((T, nk + 1), np.float32)
        goal[:, :nk] = on
        finger = np.where(on, fing, -1).astype(np.int32)
        # sustain: last CC64 event of a frame decides, otherwise the previous state holds
        sel = a.cc_num == consts.SUSTAIN_PEDAL_CC_NUMBER
        frames = (a.cc_time[sel] * fps).astype(np.int64)
        vals = a.cc_val[sel] + 1
        ok = frames < T
        ev = np.zeros(T, np.int64)
        ev[frames[ok]] = vals[ok]                                 # in list order: later wins
        state = np.where(ev >= consts.SUSTAIN_PEDAL_CC_NUMBER + 1, 1, 0)
        has = (ev >= 1) & (ev <= consts.MAX_CC_VALUE + 1)
        idx = np.where(has, np.arange(T), -1)
        np.maximum.accumulate(idx, out=idx)
        goal[:, nk] = np.where(idx >= 0, state[np.maximum(idx, 0)], 0)
        nbuf = int(round(initial_buffer_time / dt))
        if nbuf:
            goal = np.concatenate([np.zeros((nbuf, nk + 1), np.float32), goal])
            finger = np.concatenate([np.full((nbuf, nk), -1, np.int32), finger])
        return goal, finger

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
