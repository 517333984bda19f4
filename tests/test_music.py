"""Port of the reference's known-answer tests for the goal-data pipeline
(robopianist/music/midi_file_test.py:29-211, music_test.py:41-63)."""
import numpy as np
import pytest

from robopianist_amd import music
from robopianist_amd.music import constants as consts
from robopianist_amd.music import midi_file
from robopianist_amd.music.sequence import NoteSequence


@pytest.mark.parametrize("factor", [0.5, 1.0, 2.0])
def test_temporal_stretch(factor):
    midi = music.load("CMajorScaleOneHand")
    stretched = midi.stretch(factor)
    assert stretched.n_notes == midi.n_notes
    assert stretched.duration == pytest.approx(midi.duration * factor)


@pytest.mark.parametrize("factor", [-1.0, 0.0])
def test_temporal_stretch_raises(factor):
    with pytest.raises(ValueError):
        music.load("CMajorScaleOneHand").stretch(factor)


@pytest.mark.parametrize("amount", [-2, -1, 0, 1, 2])
def test_transpose(amount):
    midi = music.load("CMajorScaleOneHand")
    t = midi.transpose(amount)
    assert t.n_notes == midi.n_notes
    assert t.duration == pytest.approx(midi.duration)
    for a, b in zip(midi.seq.notes, t.seq.notes):
        assert b.pitch - a.pitch == amount


def test_trim_silence_first_note_at_zero():
    midi = music.load("TwinkleTwinkleRousseau")
    assert midi.seq.notes[0].start_time > 0
    assert midi.trim_silence().seq.notes[0].start_time == pytest.approx(0.0)


def test_twinkle_rousseau_matches_scripted_replay_length():
    """trimmed Twinkle = 158 control steps = rows of the .npy."""
    midi = music.load("TwinkleTwinkleRousseau").trim_silence()
    traj = midi_file.NoteTrajectory.from_midi(midi, 0.05)
    actions = np.load("tests/golden/twinkle_twinkle_actions.npy")
    assert len(traj) == actions.shape[0] == 158
    assert midi.n_notes == 34


def test_key_pitch_name_round_trips():
    for key in range(consts.NUM_KEYS):
        number = midi_file.key_number_to_midi_number(key)
        assert midi_file.midi_number_to_key_number(number) == key
        name = midi_file.key_number_to_note_name(key)
        assert midi_file.note_name_to_key_number(name) == key
        assert midi_file.note_name_to_midi_number(name) == number
    with pytest.raises(ValueError):
        midi_file.key_number_to_midi_number(88)
    with pytest.raises(ValueError):
        midi_file.midi_number_to_key_number(20)


def test_piano_note_validation():
    with pytest.raises(ValueError):
        midi_file.PianoNote.create(number=60, velocity=0)
    with pytest.raises(ValueError):
        midi_file.PianoNote.create(number=10, velocity=80)
    n = midi_file.PianoNote.create(number=21, velocity=80)
    assert n.key == 0 and n.name == "A0"


def test_note_trajectory_consecutive_notes_get_a_gap():
    """midi_file_test.py:176-195 — a key that is active at t-1 and has an onset at t
    is omitted at t."""
    dt = 0.1
    seq = NoteSequence()
    seq.notes.add(pitch=60, start_time=0.0, end_time=2 * dt, velocity=80)
    seq.notes.add(pitch=60, start_time=2 * dt, end_time=4 * dt, velocity=80)
    seq.total_time = 4 * dt
    traj = midi_file.NoteTrajectory.from_midi(midi_file.MidiFile(seq=seq), dt=dt)
    assert len(traj.notes) == 5
    assert [len(n) for n in traj.notes[:4]] == [1, 1, 0, 1]
    assert traj.notes[2] == []


def test_note_trajectory_sustain_events():
    """midi_file_test.py:197-211 — CC64 >= 64 on, < 64 off, held in between."""
    dt = 0.1
    seq = NoteSequence()
    seq.notes.add(pitch=60, start_time=0.0, end_time=5 * dt, velocity=80)
    seq.control_changes.add(time=1 * dt, control_number=64, control_value=127)
    seq.control_changes.add(time=3 * dt, control_number=64, control_value=0)
    seq.total_time = 5 * dt
    traj = midi_file.NoteTrajectory.from_midi(midi_file.MidiFile(seq=seq), dt=dt)
    assert traj.sustains == [0, 1, 1, 0, 0, 0]


def test_note_trajectory_validation_and_buffer():
    with pytest.raises(ValueError):
        midi_file.NoteTrajectory(dt=0.0, notes=[], sustains=[])
    with pytest.raises(ValueError):
        midi_file.NoteTrajectory(dt=0.1, notes=[[]], sustains=[])
    t = midi_file.NoteTrajectory.from_midi(music.load("CMajorScaleOneHand"), 0.05)
    n = len(t)
    t.add_initial_buffer_time(0.5)
    assert len(t) == n + 10 and t.notes[0] == []
    with pytest.raises(ValueError):
        t.add_initial_buffer_time(-1.0)


def test_library_songs_have_valid_fingering():
    """music_test.py:53-63."""
    for name in music.DEBUG_MIDIS:
        midi = music.load(name)
        assert midi.has_fingering()
        for note in midi.seq.notes:
            assert 0 <= note.part <= 9


def test_load_errors():
    with pytest.raises(KeyError):
        music.load("NotASong")
    with pytest.raises(ValueError):
        music.load("song.txt")


def test_goal_tables_match_notes():
    t = midi_file.NoteTrajectory.from_midi(music.load("CMajorScaleTwoHands"), 0.05)
    goal, finger = t.to_goal_tables()
    assert goal.shape == (151, 89) and finger.shape == (151, 88)
    for i, notes in enumerate(t.notes):
        assert sorted(np.flatnonzero(goal[i, :88])) == sorted(n.key for n in notes)
        for n in notes:
            assert finger[i, n.key] == n.fingering


def test_midi_module_edge_detection_matches_reference_semantics():
    """models/piano/midi_module.py:47-98 on an activation trace: note-ons at rising edges
    (velocity 127), note-offs at falling edges, sustain edges, messages grouped per substep
    in the reference's order."""
    import numpy as np
    from robopianist_amd.music import midi_module as mm
    from robopianist_amd.music import midi_file
    mod = mm.MidiModule()
    act = np.zeros(88, bool)
    mod.after_substep(0.002, act, False)
    assert mod.get_latest_midi_messages() == []
    act2 = act.copy(); act2[[3, 40]] = True
    mod.after_substep(0.004, act2, True)
    msgs = mod.get_latest_midi_messages()
    assert [type(m).__name__ for m in msgs] == ["NoteOn", "NoteOn", "SustainOn"]
    assert msgs[0].note == midi_file.key_number_to_midi_number(3) and msgs[0].velocity == 127
    assert msgs[1].note == midi_file.key_number_to_midi_number(40) and msgs[1].time == 0.004
    act3 = act2.copy(); act3[3] = False; act3[7] = True
    mod.after_substep(0.006, act3, False)
    msgs = mod.get_latest_midi_messages()
    assert [type(m).__name__ for m in msgs] == ["NoteOn", "NoteOff", "SustainOff"]
    assert msgs[0].note == midi_file.key_number_to_midi_number(7)
    assert msgs[1].note == midi_file.key_number_to_midi_number(3)
    assert len(mod.get_all_midi_messages()) == 6
    # the packed device trace decodes to the same events
    from robopianist_amd import engine
    trace = np.zeros((2, 1, 3, 4), np.uint32)      # [steps, envs, substeps, words]
    trace[0, 0, 1, 0] = 1 << 3                      # key 3 pressed from substep 1 of step 0
    trace[0, 0, 2, 0] = 1 << 3
    trace[1, 0, 0, 0] = 1 << 3
    trace[1, 0, 1, 2] = 1 << 0                      # key 64 on, key 3 off
    trace[1, 0, 2, 2] = 1 << 0
    ev = mm.events_from_trace(trace, sustain=[False, True], times=[0.006, 0.012], physics_timestep=0.002)
    kinds = [(type(m).__name__, round(m.time, 6)) for m in ev]
    assert kinds == [("NoteOn", 0.004), ("SustainOn", 0.008), ("NoteOn", 0.01), ("NoteOff", 0.01)]
    assert ev[0].note == midi_file.key_number_to_midi_number(3)
    assert ev[2].note == midi_file.key_number_to_midi_number(64)


def test_goal_tables_from_arrays_equal_the_note_object_path():
    """The vectorised table builder used for per-episode augmentations must reproduce
    NoteTrajectory.from_midi(...).to_goal_tables() exactly: all library songs, random
    stretches / transpositions (lazy MidiFile ops vs materialised sequences), several
    control timesteps, with and without initial buffer time."""
    from robopianist_amd.suite import variations
    rs = np.random.RandomState(0)
    augs = [variations.MidiTemporalStretch(1.0, 0.4), variations.MidiPitchShift(1.0, 7),
            variations.MidiOctaveShift(0.5, 2), variations.MidiTemporalStretch(0.5, 0.1)]
    for name in music.ALL:
        base = music.load(name)
        for dt in (0.05, 0.013):
            for _ in range(4):
                m = base
                for v in augs:
                    m = v(initial_value=m, random_state=rs)
                for buf in (0.0, 0.5):
                    fast = midi_file.NoteTrajectory.goal_tables_from_arrays(m.note_arrays(), dt, buf)
                    t = midi_file.NoteTrajectory.from_midi(m, dt)
                    t.add_initial_buffer_time(buf)
                    g, f = t.to_goal_tables()
                    assert fast is not None
                    np.testing.assert_array_equal(fast[0], g)
                    np.testing.assert_array_equal(fast[1], f)
                assert m.duration == m.seq.total_time and m.n_notes == len(m.seq.notes)


def test_goal_tables_from_arrays_defers_off_piano_notes_to_the_generic_path():
    seq = NoteSequence()
    seq.notes.add(start_time=0.0, end_time=0.1, velocity=80, pitch=10, part=0)
    seq.total_time = 0.1
    m = midi_file.MidiFile(seq=seq)
    assert midi_file.NoteTrajectory.goal_tables_from_arrays(m.note_arrays(), 0.05) is None
    with pytest.raises(ValueError):
        midi_file.NoteTrajectory.from_midi(m, 0.05)
