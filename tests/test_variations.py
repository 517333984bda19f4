"""MIDI augmentations: ports of robopianist/suite/variations_test.py:26-127, plus the
vectorised task's per-env use of them (piano_with_shadow_hands.py:151-157)."""
import warnings

import numpy as np
import pytest
import torch

from fake_physics import FakePhysics
from robopianist_amd import music
from robopianist_amd.music import library, midi_file
from robopianist_amd.suite import environment, variations
from robopianist_amd.suite.tasks import piano_with_shadow_hands

_SEED = 12345
_NUM_SAMPLES = 100


def _same_seq(a, b):
    na = [(n.pitch, n.start_time, n.end_time, n.velocity, n.part) for n in a.seq.notes]
    nb = [(n.pitch, n.start_time, n.end_time, n.velocity, n.part) for n in b.seq.notes]
    return na == nb and a.seq.total_time == b.seq.total_time


def test_midi_select_returns_midi_files():
    var = variations.MidiSelect(midi_names=music.ALL)
    rs = np.random.RandomState(_SEED)
    titles = set()
    for _ in range(_NUM_SAMPLES):
        midi = var(random_state=rs)
        assert isinstance(midi, midi_file.MidiFile)
        titles.add(midi.title)
    assert len(titles) > 1


@pytest.mark.parametrize("make", [lambda p: variations.MidiTemporalStretch(prob=p, stretch_range=0.5),
                                  lambda p: variations.MidiPitchShift(prob=p, shift_range=1),
                                  lambda p: variations.MidiOctaveShift(prob=p, octave_range=1)])
def test_output_type_and_prob_zero_identity(make):
    original = library.toy()
    rs = np.random.RandomState(_SEED)
    for _ in range(_NUM_SAMPLES):
        assert isinstance(make(0.1)(initial_value=original, random_state=rs), midi_file.MidiFile)
    for _ in range(_NUM_SAMPLES):
        assert make(0.0)(initial_value=original, random_state=rs) is original


@pytest.mark.parametrize("var", [variations.MidiTemporalStretch(prob=0.1, stretch_range=0.5),
                                 variations.MidiPitchShift(prob=0.1, shift_range=1),
                                 variations.MidiOctaveShift(prob=0.1, octave_range=1)])
def test_raises_value_error_without_a_midi(var):
    rs = np.random.RandomState(_SEED)
    with pytest.raises(ValueError):
        var(random_state=rs)
    with pytest.raises(ValueError):
        var(initial_value=1, random_state=rs)


def test_stretch_prob_one_changes_and_range_zero_keeps():
    original = library.toy()
    rs = np.random.RandomState(_SEED)
    var = variations.MidiTemporalStretch(prob=1.0, stretch_range=0.5)
    for _ in range(_NUM_SAMPLES):
        assert var(initial_value=original, random_state=rs) is not original
    var = variations.MidiTemporalStretch(prob=0.1, stretch_range=0.0)
    for _ in range(_NUM_SAMPLES):
        new = var(initial_value=original, random_state=rs)
        assert _same_seq(original, new) and new.duration == original.duration


def test_shift_range_zero_is_identity_and_bad_range_raises():
    original = library.toy()
    rs = np.random.RandomState(_SEED)
    for _ in range(_NUM_SAMPLES):
        assert variations.MidiPitchShift(prob=0.1, shift_range=0)(initial_value=original, random_state=rs) is original
        assert variations.MidiOctaveShift(prob=1.0, octave_range=0)(initial_value=original, random_state=rs) is original
    with pytest.raises(ValueError):
        variations.MidiPitchShift(prob=0.1, shift_range=0.5)
    with pytest.raises(ValueError):
        variations.MidiOctaveShift(prob=0.1, octave_range=1.5)


def test_pitch_shift_stays_on_the_piano_and_follows_the_reference_draw_order():
    original = music.load("CMajorScaleTwoHands")
    var = variations.MidiPitchShift(prob=1.0, shift_range=60)
    rs = np.random.RandomState(_SEED)
    for _ in range(50):
        new = var(initial_value=original, random_state=rs)
        pitches = [n.pitch for n in new.seq.notes]
        assert len(pitches) == original.n_notes
        assert min(pitches) >= 21 and max(pitches) <= 108
    # same stream as the reference: uniform(0,1) gate, then randint(low, high+1)
    rs_a, rs_b = np.random.RandomState(3), np.random.RandomState(3)
    new = variations.MidiPitchShift(prob=1.0, shift_range=2)(initial_value=original, random_state=rs_a)
    rs_b.uniform(0.0, 1.0)
    shift = rs_b.randint(-2, 3)
    assert [n.pitch for n in new.seq.notes] == [n.pitch + shift for n in original.seq.notes]
    assert rs_a.uniform() == rs_b.uniform()


# ---- the vectorised task ---------------------------------------------------------------------
def _aug_env(n_envs, augmentations, seed=0, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        task = piano_with_shadow_hands.PianoWithShadowHands(
            midi=music.load("CMajorScaleTwoHands"), augmentations=augmentations,
            change_color_on_activation=True, **kw)
    return environment.Environment(task, n_envs=n_envs, random_state=seed,
                                   physics=FakePhysics(task.scene, n_envs))


def _expected_tables(midi, dt):
    t = midi_file.NoteTrajectory.from_midi(midi, dt)
    return t.to_goal_tables()


def test_task_applies_augmentations_per_env_at_every_episode_start():
    E = 4
    augs = [variations.MidiTemporalStretch(prob=1.0, stretch_range=0.3),
            variations.MidiPitchShift(prob=1.0, shift_range=4)]
    env = _aug_env(E, augs, seed=7)
    task = env.task
    ts = env.reset()
    # replay the task's RandomState: env order, variation order
    rs = np.random.RandomState(7)
    base = music.load("CMajorScaleTwoHands")
    lens = []
    for e in range(E):
        midi = base
        for v in augs:
            midi = v(initial_value=midi, random_state=rs)
        g, f = _expected_tables(midi, task.control_timestep)
        lens.append(len(g))
        np.testing.assert_array_equal(task._goal_bank[e, :len(g)].numpy(), g)
        assert (task._goal_bank[e, len(g):] == 0).all()
        np.testing.assert_array_equal(task._finger_bank[e, :len(g)].numpy(), f)
        np.testing.assert_array_equal(ts.observation["goal"][e].numpy()[:89], g[0])
    assert task._song_len.tolist() == lens
    assert len(set(lens)) > 1, "different stretch factors give different episode lengths"

    # episodes end after exactly their own number of steps; the finished env (only) gets a
    # fresh augmentation at its next step, the others keep their tables
    zero = np.zeros((E,) + env.action_spec().shape)
    first_done = int(np.argmin(lens))
    for _ in range(min(lens)):
        ts = env.step(zero)
    assert bool(ts.last()[first_done])
    assert int(ts.last().sum()) == lens.count(min(lens))
    before = task._goal_bank.clone()
    ts = env.step(zero)
    assert bool(ts.first()[first_done])
    midi = base
    for v in augs:
        midi = v(initial_value=midi, random_state=rs)
    g, _ = _expected_tables(midi, task.control_timestep)
    np.testing.assert_array_equal(task._goal_bank[first_done, :len(g)].numpy(), g)
    assert int(task._song_len[first_done]) == len(g)
    others = [e for e in range(E) if lens[e] != min(lens)]
    assert torch.equal(task._goal_bank[others][:, :before.shape[1]], before[others])


def test_bank_grows_when_an_augmented_song_is_longer():
    env = _aug_env(2, [variations.MidiTemporalStretch(prob=1.0, stretch_range=0.9)], seed=1)
    task = env.task
    n0 = len(_expected_tables(music.load("CMajorScaleTwoHands"), task.control_timestep)[0])
    for _ in range(6):
        env.reset()
    assert task._goal_bank.shape[1] >= int(task._song_len.max())
    assert int(task._song_len.max()) != n0


def test_no_augmentation_draw_keeps_the_cached_tables():
    env = _aug_env(3, [variations.MidiPitchShift(prob=0.0, shift_range=3)])
    env.reset()
    g, _ = _expected_tables(music.load("CMajorScaleTwoHands"), env.task.control_timestep)
    for e in range(3):
        np.testing.assert_array_equal(env.task._goal_bank[e, :len(g)].numpy(), g)


def test_self_actuated_piano_applies_augmentations_per_env():
    """self_actuated_piano.py:119-125 on the vectorised task: one goal-bank slot per env."""
    from robopianist_amd.suite.tasks import SelfActuatedPiano
    E = 3
    augs = [variations.MidiTemporalStretch(prob=1.0, stretch_range=0.4)]
    base = music.load("TwinkleTwinkleLittleStar")
    task = SelfActuatedPiano(midi=base, augmentations=augs, control_timestep=0.05,
                             change_color_on_activation=True)
    env = environment.Environment(task, n_envs=E, random_state=3, physics=FakePhysics(task.scene, E))
    ts = env.reset()
    rs = np.random.RandomState(3)
    lens = []
    for e in range(E):
        midi = augs[0](initial_value=base, random_state=rs)
        g, _ = _expected_tables(midi, 0.05)
        lens.append(len(g))
        np.testing.assert_array_equal(task._goal_bank[e, :len(g)].numpy(), g)
        np.testing.assert_array_equal(ts.observation["goal"][e].numpy()[:89], g[0])
    assert task._len.tolist() == lens and len(set(lens)) > 1
    zero = np.zeros((E,) + env.action_spec().shape)
    for _ in range(min(lens)):
        ts = env.step(zero)
    assert ts.last().tolist() == [l == min(lens) for l in lens]
