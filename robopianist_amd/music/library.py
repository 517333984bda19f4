"""Library of debug songs with fingering (mirror of robopianist/music/library.py:28-553).

The note/fingering tables are the reference's song DATA; the construction code is
written against `sequence.NoteSequence`.
"""

from __future__ import annotations

from pathlib import Path
from typing import Callable, Dict

from robopianist_amd.music import midi_file
from robopianist_amd.music.sequence import NoteSequence

_HERE = Path(__file__).parent
_DATA_PATH = _HERE / "data"


def _seq(title: str) -> NoteSequence:
    seq = NoteSequence()
    seq.sequence_metadata.title = title
    seq.sequence_metadata.artist = "robopianist"
    return seq


def toy(right_finger: int = 1, left_finger: int = 6) -> midi_file.MidiFile:
    """library.py:28-66."""
    seq = NoteSequence()
    n = midi_file.note_name_to_midi_number
    seq.notes.add(start_time=0.0, end_time=0.5, velocity=80, pitch=n("C6"), part=right_finger)
    seq.notes.add(start_time=0.5, end_time=1.0, velocity=80, pitch=n("G5"), part=right_finger)
    seq.notes.add(start_time=0.0, end_time=0.5, velocity=80, pitch=n("C3"), part=left_finger)
    seq.notes.add(start_time=0.5, end_time=1.0, velocity=80, pitch=n("C4"), part=left_finger)
    seq.total_time = 1.0
    seq.tempos.add(qpm=60)
    return midi_file.MidiFile(seq=seq)


def twinkle_twinkle_little_star_one_hand() -> midi_file.MidiFile:
    """library.py:69-97."""
    seq = _seq("Twinkle Twinkle (one hand)")
    rows = [(60, 0.0, 0.5, 0), (60, 0.5, 1.0, 0), (67, 1.0, 1.5, 2), (67, 1.5, 2.0, 2),
            (69, 2.0, 2.5, 3), (69, 2.5, 3.0, 3), (67, 3.0, 4.0, 2),
            (65, 4.0, 4.5, 3), (65, 4.5, 5.0, 3), (64, 5.0, 5.5, 2), (64, 5.5, 6.0, 2),
            (62, 6.0, 6.5, 1), (62, 6.5, 7.0, 1), (60, 7.0, 8.0, 0)]
    for pitch, s, e, part in rows:
        seq.notes.add(pitch=pitch, start_time=s, end_time=e, velocity=80, part=part)
    seq.total_time = 8.0
    seq.tempos.add(qpm=60)
    return midi_file.MidiFile(seq=seq)


def _scale(title, pitches, right_octave, note_duration, left_octave=None) -> midi_file.MidiFile:
    seq = _seq(title)
    rh_f, lh_f = [0, 1, 2, 0, 1, 2, 3, 4], [9, 8, 7, 6, 5, 7, 6, 5]
    for i in range(8):
        seq.notes.add(pitch=12 * right_octave + pitches[i], start_time=i * note_duration,
                      end_time=(i + 1) * note_duration, velocity=80, part=rh_f[i])
        if left_octave is not None:
            seq.notes.add(pitch=12 * left_octave + pitches[i], start_time=i * note_duration,
                          end_time=(i + 1) * note_duration, velocity=80, part=lh_f[i])
    rh_b, lh_b = [3, 2, 1, 0, 2, 1, 0], [6, 7, 5, 6, 7, 8, 9]
    for i in range(7):
        seq.notes.add(pitch=12 * right_octave + pitches[7 - i - 1],
                      start_time=(8 + i) * note_duration, end_time=(9 + i) * note_duration,
                      velocity=80, part=rh_b[i])
        if left_octave is not None:
            seq.notes.add(pitch=12 * left_octave + pitches[7 - i - 1],
                          start_time=(8 + i) * note_duration, end_time=(9 + i) * note_duration,
                          velocity=80, part=lh_b[i])
    seq.total_time = 15 * note_duration
    seq.tempos.add(qpm=60)
    return midi_file.MidiFile(seq=seq)


_C_MAJOR = [0, 2, 4, 5, 7, 9, 11, 12]
_D_MAJOR = [2, 4, 6, 7, 9, 11, 13, 14]


def c_major_scale_one_hand(right_octave: int = 6, note_duration: float = 0.5):
    """library.py:100-137."""
    return _scale("C major scale (one hand)", _C_MAJOR, right_octave, note_duration)


def d_major_scale_one_hand(right_octave: int = 6, note_duration: float = 0.5):
    """library.py:140-176."""
    return _scale("D major scale (one hand)", _D_MAJOR, right_octave, note_duration)


def c_major_scale_two_hands(left_octave: int = 4, right_octave: int = 6, note_duration: float = 0.5):
    """library.py:179-237."""
    return _scale("C major scale", _C_MAJOR, right_octave, note_duration, left_octave)


def d_major_scale_two_hands(left_octave: int = 4, right_octave: int = 6, note_duration: float = 0.5):
    """library.py:240-297."""
    return _scale("D major scale", _D_MAJOR, right_octave, note_duration, left_octave)


def c_major_chord_progression_two_hands() -> midi_file.MidiFile:
    """library.py:300-334."""
    seq = _seq("C major chord progression")
    chords = [(48, 5, (60, 64, 67)), (41, 8, (65, 69, 72)), (43, 7, (67, 71, 74)),
              (48, 5, (60, 64, 67))]
    for t, (lp, lf, rps) in enumerate(chords):
        seq.notes.add(pitch=lp, start_time=t, end_time=t + 1, velocity=80, part=lf)
        for rp, rf in zip(rps, (0, 2, 4)):
            seq.notes.add(pitch=rp, start_time=t, end_time=t + 1, velocity=80, part=rf)
    seq.total_time = 4
    seq.tempos.add(qpm=60)
    return midi_file.MidiFile(seq=seq)


_TWINKLE_ROUSSEAU_FINGERING = [1, 9, 5, 0, 3, 6, 3, 8, 4, 5, 4, 8, 3, 6, 4, 8, 3, 5, 3, 6, 5, 2, 6, 2, 1, 8, 2, 1, 2, 1, 6, 2, 0, 9]
_NOCTURNE_ROUSSEAU_FINGERING = [0, 8, 4, 9, 6, 8, 6, 5, 9, 9, 6, 2, 8, 6, 5, 3, 8, 2, 9, 6, 8, 6, 5, 7, 1, 9, 6, 8, 6, 5, 0, 7, 4, 9, 6, 8, 6, 5, 0, 3, 2, 1, 0, 7, 4, 9, 6, 8, 6, 5, 0, 7, 2, 9, 6, 7, 6, 5, 7, 2, 9, 6, 7, 6, 5, 2, 7, 1, 9, 6, 8, 6, 5, 9, 4, 9, 5, 6, 5, 5, 0, 9, 2, 9, 6, 7, 6, 5, 9, 1, 9, 6, 7, 6, 5, 8, 0, 9, 6, 4, 8, 6, 5, 3, 8, 2, 1, 9, 6, 0, 3, 8, 5, 0, 1, 7, 2]


def _rousseau(filename, title, fingering, key) -> midi_file.MidiFile:
    midi = midi_file.MidiFile.from_file(_DATA_PATH / "rousseau" / filename)
    midi.seq.sequence_metadata.artist = "Rousseau"
    midi.seq.sequence_metadata.title = title
    sorted_notes = sorted(midi.seq.notes, key=key)
    assert len(fingering) == len(sorted_notes)
    for i, note in enumerate(sorted_notes):
        note.part = fingering[i]
    return midi


def twinkle_twinkle_rousseau() -> midi_file.MidiFile:
    """library.py:337-396 (notes sorted by start time)."""
    return _rousseau("twinkle-twinkle-trimmed.mid", "Twinkle Twinkle (YouTube)",
                     _TWINKLE_ROUSSEAU_FINGERING, lambda n: n.start_time)


def nocturne_rousseau() -> midi_file.MidiFile:
    """library.py:399-541 (notes sorted by (start time, pitch))."""
    return _rousseau("nocturne-trimmed.mid", "Nocturne (YouTube)",
                     _NOCTURNE_ROUSSEAU_FINGERING, lambda n: (n.start_time, n.pitch))


MIDI_NAME_TO_CALLABLE: Dict[str, Callable[[], midi_file.MidiFile]] = {
    "TwinkleTwinkleLittleStar": twinkle_twinkle_little_star_one_hand,
    "CMajorScaleOneHand": c_major_scale_one_hand,
    "CMajorScaleTwoHands": c_major_scale_two_hands,
    "DMajorScaleOneHand": d_major_scale_one_hand,
    "DMajorScaleTwoHands": d_major_scale_two_hands,
    "CMajorChordProgressionTwoHands": c_major_chord_progression_two_hands,
    "TwinkleTwinkleRousseau": twinkle_twinkle_rousseau,
    "NocturneRousseau": nocturne_rousseau,
}
