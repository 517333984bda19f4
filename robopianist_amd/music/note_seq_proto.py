"""Wire-format reader / writer for Magenta `NoteSequence` protos (`.proto` files).

The reference's main benchmarks (REPERTOIRE_150, ETUDE_12) ship as serialized
`note_seq.protobuf.music_pb2.NoteSequence` messages with the PIG fingering stored in
`note.part` (robopianist/music/midi_file.py:175-199, music/__init__.py:36-56).  note_seq is not
installable here, so this module speaks the protobuf wire format directly for the fields the
goal tables need.  Field numbers follow note_seq's `music.proto` [MEM: not verifiable in this
image -- no .proto fixture ships with the reference; tests/test_music.py checks wire
compatibility with google.protobuf for exactly these numbers]:

    NoteSequence: ticks_per_quarter = 4, tempos = 7, notes = 8, total_time = 9,
                  control_changes = 11, sequence_metadata = 19
    Note:         pitch = 1, velocity = 2, start_time = 3, end_time = 4, instrument = 7,
                  program = 8, is_drum = 9, part = 10
    Tempo:        time = 1, qpm = 2
    ControlChange: time = 1, control_number = 2, control_value = 3, instrument = 4
    SequenceMetadata: title = 1, artist = 2

Unknown fields are skipped on read (forward compatible) and not written.
"""

from __future__ import annotations

import struct

from robopianist_amd.music import sequence as seqlib

_VARINT, _I64, _LEN, _I32 = 0, 1, 2, 5


def _read_varint(buf: bytes, pos: int):
    shift = v = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _fields(buf: bytes):
    """Yields (field number, wire type, value) of one message; LEN values are bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == _VARINT:
            val, pos = _read_varint(buf, pos)
        elif wt == _I64:
            if pos + 8 > n:
                raise ValueError("truncated fixed64")
            val = buf[pos:pos + 8]; pos += 8
        elif wt == _LEN:
            ln, pos = _read_varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field")
            val = buf[pos:pos + ln]; pos += ln
        elif wt == _I32:
            if pos + 4 > n:
                raise ValueError("truncated fixed32")
            val = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield num, wt, val


def _f64(v) -> float:
    return struct.unpack("<d", v)[0]


def _i32(v: int) -> int:
    """int32 / int64 fields arrive as 64-bit two's complement varints."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def parse(data: bytes) -> seqlib.NoteSequence:
    """bytes of a serialized NoteSequence -> music.sequence.NoteSequence."""
    seq = seqlib.NoteSequence()
    seq.ticks_per_quarter = 220
    for num, wt, val in _fields(data):
        if num == 8 and wt == _LEN:
            kw = dict(pitch=0, velocity=0, start_time=0.0, end_time=0.0, part=0)
            for n2, w2, v2 in _fields(val):
                if n2 == 1 and w2 == _VARINT: kw["pitch"] = _i32(v2)
                elif n2 == 2 and w2 == _VARINT: kw["velocity"] = _i32(v2)
                elif n2 == 3 and w2 == _I64: kw["start_time"] = _f64(v2)
                elif n2 == 4 and w2 == _I64: kw["end_time"] = _f64(v2)
                elif n2 == 10 and w2 == _VARINT: kw["part"] = _i32(v2)
            seq.notes.add(**kw)
        elif num == 11 and wt == _LEN:
            kw = dict(time=0.0, control_number=0, control_value=0)
            for n2, w2, v2 in _fields(val):
                if n2 == 1 and w2 == _I64: kw["time"] = _f64(v2)
                elif n2 == 2 and w2 == _VARINT: kw["control_number"] = _i32(v2)
                elif n2 == 3 and w2 == _VARINT: kw["control_value"] = _i32(v2)
            seq.control_changes.add(**kw)
        elif num == 7 and wt == _LEN:
            kw = dict(time=0.0, qpm=0.0)
            for n2, w2, v2 in _fields(val):
                if n2 == 1 and w2 == _I64: kw["time"] = _f64(v2)
                elif n2 == 2 and w2 == _I64: kw["qpm"] = _f64(v2)
            seq.tempos.add(**kw)
        elif num == 9 and wt == _I64:
            seq.total_time = _f64(val)
        elif num == 4 and wt == _VARINT:
            seq.ticks_per_quarter = _i32(val)
        elif num == 19 and wt == _LEN:
            for n2, w2, v2 in _fields(val):
                if n2 == 1 and w2 == _LEN: seq.sequence_metadata.title = v2.decode("utf-8", "replace")
                elif n2 == 2 and w2 == _LEN: seq.sequence_metadata.artist = v2.decode("utf-8", "replace")
    return seq


# ------------------------------------------------------------------------- writer
def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(num, wt) -> bytes:
    return _varint((num << 3) | wt)


def _put_int(num, v) -> bytes:
    return b"" if v == 0 else _key(num, _VARINT) + _varint(int(v))   # proto3: defaults are omitted


def _put_f64(num, v) -> bytes:
    return b"" if v == 0.0 else _key(num, _I64) + struct.pack("<d", float(v))


def _put_bytes(num, b: bytes) -> bytes:
    return _key(num, _LEN) + _varint(len(b)) + b


def serialize(seq: seqlib.NoteSequence) -> bytes:
    out = bytearray()
    out += _put_int(4, getattr(seq, "ticks_per_quarter", 220))
    for t in seq.tempos:
        out += _put_bytes(7, _put_f64(1, t.time) + _put_f64(2, t.qpm))
    for n in seq.notes:
        out += _put_bytes(8, _put_int(1, n.pitch) + _put_int(2, n.velocity) + _put_f64(3, n.start_time)
                          + _put_f64(4, n.end_time) + _put_int(10, n.part))
    out += _put_f64(9, seq.total_time)
    for c in seq.control_changes:
        out += _put_bytes(11, _put_f64(1, c.time) + _put_int(2, c.control_number) + _put_int(3, c.control_value))
    md = seq.sequence_metadata
    if md.title or md.artist:
        body = (_put_bytes(1, md.title.encode()) if md.title else b"") + \
               (_put_bytes(2, md.artist.encode()) if md.artist else b"")
        out += _put_bytes(19, body)
    return bytes(out)
