"""Minimal NoteSequence replacement + Standard MIDI File reader.

The reference uses `note_seq.NoteSequence` protos parsed by `note_seq.midi_io`
(on top of pretty_midi); neither library exists in this image.  Only the
fields the hot path's goal tables need are kept (robopianist/music/
midi_file.py:315-362, piano_roll.py:59-204): notes (pitch, velocity, start,
end, part), control changes, total_time, tempos, title/artist.
"""

from __future__ import annotations

import copy
import dataclasses
import struct
from typing import List, Optional


@dataclasses.dataclass
class Note:
    pitch: int
    velocity: int
    start_time: float
    end_time: float
    part: int = 0


@dataclasses.dataclass
class ControlChange:
    time: float
    control_number: int
    control_value: int


@dataclasses.dataclass
class Tempo:
    time: float = 0.0
    qpm: float = 120.0


class _NoteList(list):
    def add(self, **kw) -> Note:
        n = Note(pitch=kw["pitch"], velocity=kw.get("velocity", 0),
                 start_time=float(kw.get("start_time", 0.0)),
                 end_time=float(kw.get("end_time", 0.0)), part=kw.get("part", 0))
        self.append(n)
        return n


class _TempoList(list):
    def add(self, **kw) -> Tempo:
        t = Tempo(time=kw.get("time", 0.0), qpm=kw.get("qpm", 120.0))
        self.append(t)
        return t


class _CCList(list):
    def add(self, **kw) -> ControlChange:
        c = ControlChange(time=kw["time"], control_number=kw["control_number"],
                          control_value=kw["control_value"])
        self.append(c)
        return c


@dataclasses.dataclass
class SequenceMetadata:
    title: str = ""
    artist: str = ""


class NoteSequence:
    def __init__(self):
        self.notes = _NoteList()
        self.control_changes = _CCList()
        self.tempos = _TempoList()
        self.total_time = 0.0
        self.sequence_metadata = SequenceMetadata()

    def copy(self) -> "NoteSequence":
        return copy.deepcopy(self)


# --------------------------------------------------------------------------- SMF
def _read_varlen(data: bytes, pos: int):
    v = 0
    while True:
        b = data[pos]
        pos += 1
        v = (v << 7) | (b & 0x7F)
        if not b & 0x80:
            return v, pos


def read_midi_file(path) -> NoteSequence:
    """Parses a Standard MIDI File the way note_seq.midi_io does (via pretty_midi):
    tick->second conversion through the tempo map, notes created at note-off in track
    order, control changes kept, total_time = latest note end."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"MThd":
        raise RuntimeError(f"Could not parse MIDI file {path}.")
    hlen, fmt, ntrk, division = struct.unpack(">IHHH", data[4:14])
    if division & 0x8000:
        raise RuntimeError("SMPTE time division is not supported")
    pos = 8 + hlen
    tracks = []
    for _ in range(ntrk):
        if data[pos:pos + 4] != b"MTrk":
            raise RuntimeError(f"Could not parse MIDI file {path}.")
        (tlen,) = struct.unpack(">I", data[pos + 4:pos + 8])
        tracks.append(data[pos + 8:pos + 8 + tlen])
        pos += 8 + tlen
    # pass 1: raw events with absolute ticks
    tempo_events = []  # (tick, microseconds per quarter)
    raw = []  # per track: list of (tick, kind, a, b)
    for tr in tracks:
        p = 0
        tick = 0
        status = 0
        ev = []
        while p < len(tr):
            dt, p = _read_varlen(tr, p)
            tick += dt
            b = tr[p]
            if b == 0xFF:
                mtype = tr[p + 1]
                ln, p2 = _read_varlen(tr, p + 2)
                payload = tr[p2:p2 + ln]
                p = p2 + ln
                if mtype == 0x51 and ln == 3:
                    tempo_events.append((tick, int.from_bytes(payload, "big")))
                elif mtype == 0x2F:
                    break
                continue
            if b in (0xF0, 0xF7):
                ln, p2 = _read_varlen(tr, p + 1)
                p = p2 + ln
                continue
            if b & 0x80:
                status = b
                p += 1
            kind = status & 0xF0
            if kind in (0x80, 0x90, 0xA0, 0xB0, 0xE0):
                a, c = tr[p], tr[p + 1]
                p += 2
                ev.append((tick, kind, a, c))
            elif kind in (0xC0, 0xD0):
                p += 1
            else:
                raise RuntimeError(f"Could not parse MIDI file {path}.")
        raw.append(ev)
    tempo_events.sort(key=lambda t: t[0])
    if not tempo_events or tempo_events[0][0] != 0:
        tempo_events.insert(0, (0, 500000))
    # tick -> seconds
    seg = []  # (tick0, sec0, sec_per_tick)
    sec = 0.0
    for i, (tk, us) in enumerate(tempo_events):
        if i > 0:
            ptk, _, pspt = seg[-1]
            sec = seg[-1][1] + (tk - ptk) * pspt
        seg.append((tk, sec, us * 1e-6 / division))

    def t2s(tk):
        lo = 0
        for i in range(len(seg)):
            if seg[i][0] <= tk:
                lo = i
            else:
                break
        tk0, s0, spt = seg[lo]
        return s0 + (tk - tk0) * spt

    seq = NoteSequence()
    for tk, us in tempo_events:
        seq.tempos.add(time=t2s(tk), qpm=60e6 / us)
    for ev in raw:
        last_on = {}
        for tk, kind, a, c in ev:
            if kind == 0x90 and c > 0:
                last_on.setdefault(a, []).append((tk, c))
            elif kind == 0x80 or (kind == 0x90 and c == 0):
                if a in last_on and last_on[a]:
                    # pretty_midi closes every open note-on of this pitch at this note-off
                    opens = last_on[a]
                    keep = []
                    for (stk, vel) in opens:
                        if stk < tk:
                            seq.notes.append(Note(pitch=a, velocity=vel, start_time=t2s(stk),
                                                  end_time=t2s(tk)))
                        else:
                            keep.append((stk, vel))
                    last_on[a] = keep
            elif kind == 0xB0:
                seq.control_changes.add(time=t2s(tk), control_number=a, control_value=c)
    seq.total_time = max((n.end_time for n in seq.notes), default=0.0)
    return seq


# ----------------------------------------------------- sequences_lib equivalents
def stretch_note_sequence(seq: NoteSequence, factor: float) -> NoteSequence:
    """note_seq.sequences_lib.stretch_note_sequence (no-op for factor == 1)."""
    if factor == 1.0:
        return seq.copy()
    out = seq.copy()
    for n in out.notes:
        n.start_time *= factor
        n.end_time *= factor
    for c in out.control_changes:
        c.time *= factor
    for t in out.tempos:
        t.time *= factor
        t.qpm /= factor
    out.total_time *= factor
    return out


def transpose_note_sequence(seq: NoteSequence, amount: int, min_allowed_pitch: int,
                            max_allowed_pitch: int):
    """Out-of-range notes are deleted (sequences_lib.transpose_note_sequence)."""
    out = seq.copy()
    kept = _NoteList()
    deleted = 0
    for n in out.notes:
        n.pitch += amount
        if n.pitch < min_allowed_pitch or n.pitch > max_allowed_pitch:
            deleted += 1
        else:
            kept.append(n)
    out.notes = kept
    return out, deleted


def extract_subsequence(seq: NoteSequence, start_time: float, end_time: float,
                        preserve_control_numbers=(64, 66, 67)) -> NoteSequence:
    """note_seq.sequences_lib.extract_subsequence: notes starting in [start, end),
    truncated at `end`, shifted to t=0; control changes likewise, with the most
    recent earlier event of the preserved controllers re-inserted at t=0."""
    out = NoteSequence()
    out.sequence_metadata = copy.deepcopy(seq.sequence_metadata)
    out.tempos = copy.deepcopy(seq.tempos)
    for n in seq.notes:
        if n.start_time < start_time or n.start_time >= end_time:
            continue
        m = copy.copy(n)
        m.start_time = n.start_time - start_time
        m.end_time = min(n.end_time, end_time) - start_time
        out.notes.append(m)
    latest: dict = {}
    for c in sorted(seq.control_changes, key=lambda c: c.time):
        if c.time < start_time:
            if c.control_number in preserve_control_numbers:
                latest[c.control_number] = c
            continue
        if c.time >= end_time:
            continue
        out.control_changes.add(time=c.time - start_time, control_number=c.control_number,
                                control_value=c.control_value)
    for num, c in latest.items():
        out.control_changes.insert(0, ControlChange(0.0, num, c.control_value))
    out.total_time = max((n.end_time for n in out.notes), default=0.0)
    return out
