"""The vectorised Environment on the real engine (GPU): ports of the reference's
env-level tests (suite/suite_test.py:35-56, piano_with_shadow_hands_test.py:228-242,
examples/self_actuated_piano_env.py:82-109) plus engine-vs-oracle checks at the
env.step boundary."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _quiet():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def _np(x):
    return x.detach().cpu().numpy()


def test_suite_load_smoke_all_debug_envs():
    """suite_test.py:35-56 with n_envs=4."""
    from robopianist_amd import suite
    for name in suite.DEBUG[:4]:
        env = suite.load(name, seed=12345, n_envs=4, task_kwargs=dict(primitive_fingertip_collisions=True))
        spec = env.action_spec()
        assert np.isfinite(spec.minimum).all() and np.isfinite(spec.maximum).all()
        ts = env.reset()
        assert ts.reward is None and ts.discount is None
        rng = np.random.RandomState(12345)
        for _ in range(10):
            a = rng.uniform(spec.minimum, spec.maximum, size=(4,) + spec.shape)
            ts = env.step(a)
            ospec = env.observation_spec()
            for k, v in ts.observation.items():
                assert tuple(v.shape[1:]) == ospec[k].shape
                assert torch.isfinite(v).all()
            assert torch.isfinite(ts.reward).all()
        # random actions may (rarely) exceed the 24-contact cap; nothing else may fire
        assert int(env.physics.warn.max()) & ~2 == 0


def test_unknown_environment_raises():
    from robopianist_amd import suite
    with pytest.raises(ValueError):
        suite.load("RoboPianist-debug-NoSuchSong-v0")


def test_failure_termination_with_applied_force():
    """piano_with_shadow_hands_test.py:228-242."""
    from robopianist_amd.suite import environment
    from robopianist_amd.suite.tasks import piano_with_shadow_hands
    from test_tasks_host import _get_test_midi
    task = piano_with_shadow_hands.PianoWithShadowHands(
        midi=_get_test_midi(0.01), control_timestep=0.01, wrong_press_termination=True,
        primitive_fingertip_collisions=True)
    env = environment.Environment(task, n_envs=3, precision=64)
    env.reset()
    f = np.zeros(task.scene.model.nv)
    f[task.piano.joints] = 3.0
    env.physics.set_qfrc_applied(f)
    ts = env.step(np.zeros((3,) + env.action_spec().shape))
    assert bool(ts.last().all())
    assert bool(task.should_terminate_episode(env.physics).all())
    # "Failure, so discount should be 0.0": the reference asserts the task's discount
    np.testing.assert_array_equal(_np(task.get_discount(env.physics)), 0.0)


def test_env_step_matches_oracle_and_key_trace():
    from oracle.rp_oracle import Oracle
    from robopianist_amd import engine, music
    from robopianist_amd.suite import environment
    from robopianist_amd.suite.tasks import piano_with_shadow_hands
    task = piano_with_shadow_hands.PianoWithShadowHands(
        midi=music.load("CMajorScaleTwoHands"), gravity_compensation=True,
        primitive_fingertip_collisions=True)
    env = environment.Environment(task, n_envs=2, precision=64, record_key_trace=True)
    m = task.scene.model
    orc = Oracle(m, env.physics.engine.blob)
    spec = env.action_spec()
    rng = np.random.RandomState(7)
    env.reset()
    for step in range(12):
        a = spec.minimum + rng.uniform(0.2, 0.9, spec.shape) * (spec.maximum - spec.minimum)
        ts = env.step(np.tile(a, (2, 1)))
        # same ctrl on the oracle: action layout = [right 22, left 22, sustain]
        orc.ctrl[task.right_hand.actuators] = a[:22]
        orc.ctrl[task.left_hand.actuators] = a[22:44]
        acts = []
        for _ in range(10):
            orc.step()
            q = np.clip(orc.qpos[:88], 0, m.jnt_range[:88, 1])
            acts.append(np.abs(q - m.jnt_range[:88, 1]) <= 0.00872665)
        np.testing.assert_allclose(_np(env.physics.qpos)[0], orc.qpos, atol=1e-9)
        # sensors seen by the energy reward: force of the last substep's pre-integration
        # state, velocity of the new state
        np.testing.assert_allclose(_np(env.physics.act_vel)[0], orc.actuator_velocity, atol=1e-8)
        np.testing.assert_allclose(_np(env.physics.act_force)[0], orc.actuator_force, atol=1e-7)
        trace = engine.decode_key_trace(_np(env.key_trace).view(np.uint32))
        np.testing.assert_array_equal(trace[0], np.array(acts))
        np.testing.assert_array_equal(_np(task.piano.activation)[0], acts[-1])
        # fingertip sites
        tips = _np(env.physics.site_xpos(list(task.right_hand.fingertip_sites)))[0]
        np.testing.assert_allclose(
            tips, orc.site_xpos.reshape(-1, 3)[task.right_hand.fingertip_sites], atol=1e-9)


def test_step_after_last_resets_only_finished_envs():
    from robopianist_amd import music
    from robopianist_amd.suite import environment
    from robopianist_amd.suite.tasks import piano_with_shadow_hands
    midis = [music.load("CMajorChordProgressionTwoHands"), music.load("CMajorScaleTwoHands")]
    task = piano_with_shadow_hands.PianoWithShadowHands(midi=midis, primitive_fingertip_collisions=True)
    env = environment.Environment(task, n_envs=2)
    spec = env.action_spec()
    a = np.tile(0.5 * (spec.minimum + spec.maximum), (2, 1))
    env.reset()
    for t in range(81):
        ts = env.step(a)
    assert ts.step_type.tolist() == [2, 1]
    q_before = _np(env.physics.qpos)[1].copy()
    ts = env.step(a)
    assert ts.step_type.tolist() == [0, 1]
    assert float(ts.reward[0]) == 0.0 and float(ts.discount[0]) == 1.0
    # env 0 was reset and NOT simulated; env 1 kept going
    assert np.abs(_np(env.physics.qpos)[0, 88:]).max() == 0.0
    assert np.abs(_np(env.physics.qpos)[1] - q_before).max() > 0
    assert int(task._t_idx[0]) == 0 and int(task._t_idx[1]) == 82


def test_self_actuated_oracle_policy_is_perfect():
    """examples/self_actuated_piano_env.py:82-109: all six musical metrics == 1."""
    from robopianist_amd import music
    from robopianist_amd.suite import environment
    from robopianist_amd.suite.tasks import self_actuated_piano
    from robopianist_amd.wrappers import MidiEvaluationWrapper
    task = self_actuated_piano.SelfActuatedPiano(midi=music.load("TwinkleTwinkleLittleStar"),
                                                 n_steps_lookahead=0)
    env = MidiEvaluationWrapper(environment.Environment(task, n_envs=3))
    spec = env.action_spec()
    ts = env.reset()
    while True:
        goal = _np(ts.observation["goal"])[:, :89]
        act = np.where(goal[:, :88] > 0, spec.maximum[:88], spec.minimum[:88])
        ts = env.step(np.concatenate([act, goal[:, 88:]], axis=1))
        if bool(ts.last().all()):
            break
    for k, v in env.get_musical_metrics().items():
        assert v == pytest.approx(1.0), k


def test_scripted_twinkle_replay_through_canonical_wrapper():
    """BASELINE config #2 plumbing: notebook kwargs + CanonicalSpecWrapper + .npy replay."""
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper, MidiEvaluationWrapper
    env = suite.load(
        "RoboPianist-debug-TwinkleTwinkleRousseau-v0", n_envs=8,
        task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                         primitive_fingertip_collisions=True, reduced_action_space=False,
                         n_steps_lookahead=10))
    env = MidiEvaluationWrapper(CanonicalSpecWrapper(env))
    assert env.action_spec().minimum.min() == -1 and env.action_spec().maximum.max() == 1
    actions = np.load("tests/golden/twinkle_twinkle_actions.npy")
    ts = env.reset()
    assert ts.observation["goal"].shape == (8, 11 * 89)
    total = 0
    for t, a in enumerate(actions):
        ts = env.step(np.tile(a, (8, 1)))
        total += 1
        assert torch.isfinite(ts.reward).all()
        if t < len(actions) - 1:
            assert bool(ts.mid().all())
    assert bool(ts.last().all()) and total == 158
    assert int(env.physics.warn.max()) == 0
    metrics = env.get_musical_metrics()
    assert 0.0 <= metrics["f1"] <= 1.0


def test_scripted_actions_replay_inside_the_prestep_launch_is_bitwise_the_host_loop():
    """Round 6: `ScriptedActions` -- the reference example's `for t: env.step(actions[t])`
    (/root/reference/examples/piano_with_shadow_hands_env.py:110-141) as an action source: the pre-step launch reads
    every env's own row of the table and advances the row index (0 at a FIRST step).  Against the host loop that gathers
    the rows and does the index arithmetic with torch ops: same TimeSteps, same state, same indices, bit for bit, across
    staggered episode ends (envs at different episode times) and through the torch fallback of the wrapper."""
    from robopianist_amd import suite
    from robopianist_amd.suite.scripted import ScriptedActions
    from robopianist_amd.wrappers import CanonicalSpecWrapper
    actions = np.load("tests/golden/twinkle_twinkle_actions.npy")
    T, E = actions.shape[0], 12
    def make():
        return CanonicalSpecWrapper(suite.load(
            "RoboPianist-debug-TwinkleTwinkleRousseau-v0", n_envs=E, seed=1,
            task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                             primitive_fingertip_collisions=True, n_steps_lookahead=1)))
    a, b = make(), make()
    dev, dt = a.physics.device, a.physics.dtype
    table = torch.as_tensor(actions, device=dev, dtype=dt)
    start = (torch.arange(E, device=dev) * 13) % T
    ia, ib = start.clone(), start.clone()
    script = ScriptedActions(table, ib)
    a.reset(); b.reset()
    for t in range(T + 40):
        ta = a.step(table.index_select(0, ia))
        ia.add_(1).clamp_(max=T - 1).masked_fill_(ta.step_type == 0, 0)
        tb = b.step(script)
        assert torch.equal(ta.step_type, tb.step_type) and torch.equal(ia, ib), t
        assert torch.equal(ta.reward, tb.reward) and torch.equal(ta.discount, tb.discount), t
        assert torch.equal(a.physics.qpos, b.physics.qpos) and torch.equal(a.physics.ctrl, b.physics.ctrl), t
        for k in ta.observation:
            assert torch.equal(ta.observation[k], tb.observation[k]), (t, k)
    assert int((ia == 0).sum()) < E   # (the envs are at different episode times)
    # the helper's own torch form (what every non-HIP path uses) against the launch's index arithmetic
    ic = start.clone()
    sc = ScriptedActions(table, ic)
    assert torch.equal(sc.take(), table.index_select(0, start))
    sc.advance(torch.zeros(E, dtype=torch.bool, device=dev))
    assert torch.equal(ic, (start + 1).clamp(max=T - 1))


def test_graphed_step_matches_eager():
    """wrappers.GraphedStepWrapper: a whole env.step replayed from one captured hipGraph
    gives the same TimeSteps and the same physics state as the eager path, across an
    episode boundary (device-side auto reset) as well."""
    import os
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper, GraphedStepWrapper
    actions = np.load(os.path.join(os.path.dirname(__file__), "golden", "twinkle_twinkle_actions.npy"))
    kw = dict(seed=7, n_envs=6, precision=64,
              task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                               primitive_fingertip_collisions=True, n_steps_lookahead=3))
    name = "RoboPianist-debug-TwinkleTwinkleRousseau-v0"
    eager = CanonicalSpecWrapper(suite.load(name, **kw))
    graphed = GraphedStepWrapper(CanonicalSpecWrapper(suite.load(name, **kw)), warmup_steps=2)
    eager.reset(); graphed.reset()
    dev = eager.physics.device
    T = actions.shape[0]
    for t in list(range(8)) + list(range(T - 6, T)) + list(range(3)):
        a = torch.as_tensor(actions[t % T], device=dev, dtype=torch.float64).expand(6, -1)
        ts_e, ts_g = eager.step(a), graphed.step(a)
        if t == T - 6:  # jump both envs close to the end of the song to cross an episode boundary
            for env in (eager, graphed):
                env.task._t_idx.fill_(T - 5)
        assert torch.equal(ts_e.step_type, ts_g.step_type)
        np.testing.assert_allclose(_np(ts_e.reward), _np(ts_g.reward), rtol=0, atol=1e-12)
        np.testing.assert_allclose(_np(ts_e.discount), _np(ts_g.discount), rtol=0, atol=0)
        for k in ts_e.observation:
            np.testing.assert_allclose(_np(ts_e.observation[k]), _np(ts_g.observation[k]), rtol=0, atol=1e-12)
        np.testing.assert_allclose(_np(eager.physics.qpos), _np(graphed.physics.qpos), rtol=0, atol=1e-12)
    assert graphed.graph_captured
    assert (ts_g.step_type == 0).any() or (ts_g.step_type == 1).all()


def test_fused_rewards_match_torch_terms():
    """include/rp_task.h: the one-launch reward kernel against the torch reward functions
    (the restatement of piano_with_shadow_hands.py:251-331), term by term, on the replay
    and on random actions, with and without the fingering / forearm terms."""
    import os
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper
    actions = np.load(os.path.join(os.path.dirname(__file__), "golden", "twinkle_twinkle_actions.npy"))
    name = "RoboPianist-debug-TwinkleTwinkleRousseau-v0"
    for extra in (dict(), dict(disable_fingering_reward=True, disable_forearm_reward=True)):
        for precision, tol in ((64, 1e-12), (32, 2e-5)):
            if extra.get("disable_fingering_reward"):
                # the OT fingering term replaces it in the reference: host path, not fused
                continue
            env = CanonicalSpecWrapper(suite.load(
                name, seed=3, n_envs=5, precision=precision,
                task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                                 primitive_fingertip_collisions=True, **extra)))
            task = env.task
            env.reset()
            rng = np.random.RandomState(0)
            for t in range(25):
                a = actions[t] if t < 15 else rng.uniform(-1, 1, size=actions.shape[1])
                ts = env.step(np.tile(a, (5, 1)))
                fused_total = ts.reward.clone()
                fused_terms = {k: v.clone() for k, v in task.reward_fn.reward_terms.items()}
                assert task._fused_rewards is not None
                ref_total = task.reward_fn.compute(env.physics)  # torch functions, same state
                for k, v in task.reward_fn.reward_terms.items():
                    np.testing.assert_allclose(_np(fused_terms[k]), _np(v), rtol=0, atol=tol, err_msg=k)
                live = _np(ts.step_type) != 0
                np.testing.assert_allclose(_np(fused_total)[live], _np(ref_total)[live], rtol=0, atol=5 * tol)


def _load_pair(n_envs, precision, **task_kwargs):
    """Two identical envs: HIP task layer (rp_task_advance) and the torch task hooks."""
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper
    name = "RoboPianist-debug-TwinkleTwinkleRousseau-v0"
    kw = dict(seed=11, n_envs=n_envs, precision=precision,
              task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                               primitive_fingertip_collisions=True, **task_kwargs))
    fused = CanonicalSpecWrapper(suite.load(name, **kw))
    ref = CanonicalSpecWrapper(suite.load(name, **kw))
    ref.task._use_fused_advance = False
    ref.task._use_fused_rewards = False
    return fused, ref


@pytest.mark.parametrize("wrong_press", [False, True])
def test_fused_task_advance_matches_torch_hooks(wrong_press):
    """include/rp_task.h rp_task_advance against the torch restatement of the reference's
    hooks (after_substep, after_step, observables, rewards, termination, discount, auto
    reset), every TimeStep field and every persistent state array, across episode ends,
    wrong-press terminations and steps past the end of the song."""
    import os
    actions = np.load(os.path.join(os.path.dirname(__file__), "golden", "twinkle_twinkle_actions.npy"))
    T, E = actions.shape[0], 6
    fused, ref = _load_pair(E, 64, n_steps_lookahead=4, wrong_press_termination=wrong_press)
    fused.reset(); ref.reset()
    assert fused.task.fused_advance_for(fused.physics) is not None and fused.task.fused_prestep_for(fused.physics) is not None
    assert ref.task.fused_advance_for(ref.physics) is None
    rng = np.random.RandomState(5)
    dev = fused.physics.device
    seen = dict(first=0, last=0, zero_discount=0)
    for step in range(70):
        if step < 20:
            a = np.tile(actions[step], (E, 1))
        else:  # different envs do different things: some press wrong keys, some finish
            a = rng.uniform(-1, 1, size=(E, actions.shape[1]))
            a[0] = actions[step % T]
        if step == 25:
            for env in (fused, ref):
                env.task._t_idx[1:3] = T - 3   # these envs reach the end of the song soon
        at = torch.as_tensor(a, device=dev, dtype=torch.float64)
        ts_f, ts_r = fused.step(at), ref.step(at)
        assert torch.equal(ts_f.step_type, ts_r.step_type), step
        np.testing.assert_allclose(_np(ts_f.reward), _np(ts_r.reward), rtol=0, atol=1e-12, err_msg=str(step))
        np.testing.assert_allclose(_np(ts_f.discount), _np(ts_r.discount), rtol=0, atol=0)
        assert ts_f.observation.keys() == ts_r.observation.keys()
        for k in ts_f.observation:
            np.testing.assert_allclose(_np(ts_f.observation[k]), _np(ts_r.observation[k]), rtol=0, atol=1e-12,
                                       err_msg=f"{k} @ {step}")
        tf, tr = fused.task, ref.task
        for name in ("_t_idx", "_should_terminate", "_failure_termination", "_discount", "_goal_current",
                     "_finger_current", "_finger_next", "_fingering_state", "_goal_state"):
            assert torch.equal(getattr(tf, name), getattr(tr, name)), f"{name} @ {step}"
        for name in ("_activation", "_sustain_activation", "_state", "_normalized_state", "_sustain_state"):
            assert torch.equal(getattr(tf.piano, name), getattr(tr.piano, name)), f"piano.{name} @ {step}"
        assert torch.equal(fused._needs_reset, ref._needs_reset)
        # the pre-step launch (rp_task_prestep: canonical action -> bounds, masks, ctrl scatter, sustain latch) and
        # rp_step_masked against the wrapper's torch expression, the torch hooks and reset / forward / step
        assert torch.equal(fused.physics.ctrl, ref.physics.ctrl), step
        assert torch.equal(fused.physics.qpos, ref.physics.qpos) and torch.equal(fused.physics.qvel, ref.physics.qvel), step
        seen["first"] += int((ts_f.step_type == 0).sum())
        seen["last"] += int((ts_f.step_type == 2).sum())
        seen["zero_discount"] += int(((tf._discount == 0) & (ts_f.step_type == 2)).sum())
    # the scenario did exercise the interesting branches
    assert seen["last"] >= 2 and seen["first"] >= 2, seen
    if wrong_press:
        assert seen["zero_discount"] >= 1, seen


def test_subclass_before_step_is_not_bypassed_and_nan_actions_propagate():
    """(a) a task subclass that overrides before_step must not be bypassed by the pre-step launch
    (which restates the STOCK hook); (b) with clip=True a NaN action stays NaN on the fused path, as torch.clamp
    keeps it on the torch path (it then reaches ctrl and the engine flags the env)."""
    from robopianist_amd import music
    from robopianist_amd.suite import environment
    from robopianist_amd.suite.tasks import piano_with_shadow_hands as pw
    from robopianist_amd.wrappers import CanonicalSpecWrapper

    class Halved(pw.PianoWithShadowHands):
        def before_step(self, physics, action):
            super().before_step(physics, torch.as_tensor(action) * 0.5)

    midi = music.load("TwinkleTwinkleRousseau")
    kw = dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True)
    env = environment.Environment(Halved(midi=midi, **kw), n_envs=2, random_state=3, precision=64)
    env.reset()
    assert env.task.fused_advance_for(env.physics) is not None and env.task.fused_prestep_for(env.physics) is None
    spec = env.action_spec()
    a = torch.as_tensor(np.tile(0.5 * (spec.minimum + spec.maximum), (2, 1)), device=env.physics.device)
    env.step(a)
    ctrl = _np(env.physics.ctrl)
    hand = np.concatenate([_np(env.task._rh_act).reshape(-1), _np(env.task._lh_act).reshape(-1)]).astype(int)
    np.testing.assert_allclose(ctrl[0, hand], 0.5 * _np(a)[0, :-1], rtol=0, atol=0)
    # (b) NaN through the canonical wrapper with clipping, fused vs torch path
    fused, ref = _load_pair(2, 64)
    fused.reset(); ref.reset()
    assert fused.task.fused_prestep_for(fused.physics) is not None
    an = np.zeros((2, fused.action_spec().shape[0])); an[1, 3] = np.nan
    for env2 in (fused, ref):
        env2._clip = True
        env2.step(torch.as_tensor(an, device=env2.physics.device, dtype=torch.float64))
    cf, cr = _np(fused.physics.ctrl), _np(ref.physics.ctrl)
    assert np.isnan(cf[1]).sum() == 1 and (np.isnan(cf) == np.isnan(cr)).all()
    np.testing.assert_array_equal(cf[0], cr[0])


def test_suite_load_legacy_step_false_runs_mj_step_order():
    """suite.load(legacy_step=False) (robopianist/suite/__init__.py:55,91): physics.step() = mj_step.  Same state
    trajectory as the legacy order on the same actions; the fingertip sites the task reads lag the state by one mj_step
    (what mjData holds after mj_step); TimeSteps stay finite and episodes end at the same step."""
    import os
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper
    actions = np.load(os.path.join(os.path.dirname(__file__), "golden", "twinkle_twinkle_actions.npy"))
    name = "RoboPianist-debug-TwinkleTwinkleRousseau-v0"
    kw = dict(seed=3, n_envs=4, precision=64, task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                                                                primitive_fingertip_collisions=True))
    a_env = CanonicalSpecWrapper(suite.load(name, **kw))
    b_env = CanonicalSpecWrapper(suite.load(name, legacy_step=False, **kw))
    a_env.reset(); b_env.reset()
    lag = 0.0
    for t in range(40):
        at = torch.as_tensor(np.tile(actions[t], (4, 1)), device=a_env.physics.device, dtype=torch.float64)
        ta, tb = a_env.step(at), b_env.step(at)
        assert torch.equal(a_env.physics.qpos, b_env.physics.qpos) and torch.equal(a_env.physics.qvel, b_env.physics.qvel), t
        assert torch.equal(ta.step_type, tb.step_type) and bool(torch.isfinite(tb.reward).all())
        lag = max(lag, float((a_env.physics.site_xpos_eng - b_env.physics.site_xpos_eng).abs().max()))
    assert 1e-6 < lag < 5e-2, lag   # (one 5 ms mj_step of fingertip motion: centimetres at most, the replay flails at metres per second)


@pytest.mark.parametrize("precision", [64, 32])
def test_fused_launch_writes_the_trajectory_record_of_the_gather(precision):
    """Round 5: the multi-GPU gather's per-env record (qpos | reward | discount | step type | 88
    activation bits) comes out of the rp_task_advance launch, into two preallocated buffers in turn -- byte for byte what
    distributed.pack_trajectory_record builds from the TimeStep with half a dozen torch launches and two allocations."""
    import os
    from robopianist_amd import distributed as rpd
    actions = np.load(os.path.join(os.path.dirname(__file__), "golden", "twinkle_twinkle_actions.npy"))
    fused, _ = _load_pair(3, precision)
    fused.reset()
    fa = fused.task.fused_advance_for(fused.physics)
    fa.enable_trajectory_record(2)
    nv = fused.physics.qpos.shape[1]
    seen, ptrs = 0, set()
    for t in range(45):
        a = np.tile(actions[t], (3, 1)); a[1] = actions[(t + 30) % len(actions)]
        ts = fused.step(torch.as_tensor(a, device=fused.physics.device, dtype=fused.physics.dtype))
        rec = fa.trajectory_record
        ref = rpd.pack_trajectory_record(fused.physics.qpos, ts.reward, ts.discount, ts.step_type, fused.task.piano.activation)
        assert rec.dtype == ref.dtype and rec.shape == ref.shape == (3, nv + (5 if precision == 64 else 6))
        assert torch.equal(rec.view(torch.uint8), ref.view(torch.uint8)), t
        assert torch.equal(rpd.unpack_key_activation(rec, nv), fused.task.piano.activation)
        seen += int(fused.task.piano.activation.sum()); ptrs.add(rec.data_ptr())
    assert seen > 0 and len(ptrs) == 2   # keys were pressed; exactly the two preallocated buffers were used


def test_midi_augmentations_fused_path_matches_torch_hooks():
    """MIDI augmentations (suite/variations.py) on the HIP task layer: per-env goal bank
    slots are regenerated on the host at every episode start; the fused launch must hand
    out the same TimeSteps as the torch hooks driven by an identically seeded RandomState."""
    from robopianist_amd import music, suite
    from robopianist_amd.suite import variations
    E = 5
    def make(fused):
        augs = [variations.MidiTemporalStretch(prob=0.8, stretch_range=0.4),
                variations.MidiPitchShift(prob=0.8, shift_range=5)]
        env = suite.load("RoboPianist-debug-CMajorScaleTwoHands-v0", seed=11, n_envs=E, precision=64,
                         task_kwargs=dict(control_timestep=0.05, gravity_compensation=True,
                                          primitive_fingertip_collisions=True, n_steps_lookahead=3,
                                          augmentations=augs))
        if not fused:
            env.task._use_fused_advance = False
            env.task._use_fused_rewards = False
        return env
    fused, ref = make(True), make(False)
    ts_f, ts_r = fused.reset(), ref.reset()
    assert fused.task.fused_advance_for(fused.physics) is not None
    assert torch.equal(fused.task._song_len, ref.task._song_len)
    assert len(set(fused.task._song_len.tolist())) > 1
    rng = np.random.RandomState(2)
    spec = fused.action_spec()
    n_first = 0
    for step in range(int(fused.task._song_len.max()) + 40):
        a = torch.as_tensor(rng.uniform(spec.minimum, spec.maximum, size=(E, spec.shape[0])),
                            device=fused.physics.device)
        ts_f, ts_r = fused.step(a), ref.step(a)
        assert torch.equal(ts_f.step_type, ts_r.step_type), step
        np.testing.assert_allclose(_np(ts_f.reward), _np(ts_r.reward), rtol=0, atol=1e-12)
        for k in ts_f.observation:
            np.testing.assert_allclose(_np(ts_f.observation[k]), _np(ts_r.observation[k]), rtol=0, atol=1e-12,
                                       err_msg=f"{k} @ {step}")
        assert torch.equal(fused.task._song_len, ref.task._song_len)
        assert torch.equal(fused.task._goal_bank[:, :10], ref.task._goal_bank[:, :10])
        if step > 0:
            n_first += int((ts_f.step_type == 0).sum())
    assert n_first >= E, "every env restarted (with a fresh augmentation) at least once"


@pytest.mark.parametrize("side", ["right", "left"])
def test_one_hand_task_on_the_engine(side):
    """PianoWithOneShadowHand on the HIP engine (single 26-link tree): protocol, observables,
    and the `position` observable (root body xpos, hands/base.py:111-114) against the
    closed form for slide joints: body_pos + sum_j R axis_j q_j."""
    from robopianist_amd import music
    from robopianist_amd.model import spec
    from robopianist_amd.suite import environment
    from robopianist_amd.suite.tasks import PianoWithOneShadowHand
    E = 4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        task = PianoWithOneShadowHand(midi=music.load("CMajorScaleTwoHands"), hand_side=side,
                                      control_timestep=0.05, gravity_compensation=True,
                                      primitive_fingertip_collisions=True, n_steps_lookahead=2)
    env = environment.Environment(task, n_envs=E, random_state=3, precision=64)
    ts = env.reset()
    hand = task._hand
    m = task.scene.model
    name = hand.name
    spec_a = env.action_spec()
    assert spec_a.shape == (23,)
    rng = np.random.RandomState(0)
    R = spec.quat_to_mat(spec.quat_normalize(m.body_quat[hand.root_body_id]))
    slide = [int(j) for j in hand.joints[-hand.n_forearm_dofs:]]
    for step in range(40):
        a = rng.uniform(spec_a.minimum, spec_a.maximum, size=(E, 23))
        a[:, :22] = 0.3 * a[:, :22] + 0.7 * np.clip(0.0, spec_a.minimum[:22], spec_a.maximum[:22])
        ts = env.step(torch.as_tensor(a, device=env.physics.device))
        q = _np(env.physics.qpos)
        want = m.body_pos[hand.root_body_id][None, :] + sum(
            q[:, j:j + 1] * (R @ m.jnt_axis[j])[None, :] for j in slide)
        np.testing.assert_allclose(_np(ts.observation[f"{name}/position"]), want, rtol=0, atol=1e-12)
        assert np.isfinite(_np(ts.reward)).all()
        assert ts.observation["fingering"].shape == (E, 5)
        assert ts.observation["goal"].shape == (E, 3 * 89)
    assert int(env.physics.warn.max()) == 0
    assert np.abs(q[:, slide]).max() > 1e-3, "the forearm did move"
    terms = task.reward_fn.reward_terms
    assert set(terms) == {"key_press_reward", "sustain_reward", "energy_reward", "fingering_reward"}
    assert float(terms["energy_reward"].max()) <= 0.0


@pytest.mark.parametrize("augment", [False, True])
def test_checkpoint_resume_is_bitwise(augment):
    """Environment.state_dict / load_state_dict (SURVEY.md §5 checkpoint/resume): a second env
    restored from a mid-episode snapshot continues the rollout bit for bit -- physics state
    (qpos, qvel, warm start), rewards, observations, across episode ends, with and without
    per-episode MIDI augmentations (the snapshot carries the per-env goal bank and the host
    RandomState)."""
    import os
    from robopianist_amd import suite
    from robopianist_amd.suite import variations
    from robopianist_amd.wrappers import CanonicalSpecWrapper
    actions = np.load(os.path.join(os.path.dirname(__file__), "golden", "twinkle_twinkle_actions.npy"))
    E = 6

    def make():
        augs = [variations.MidiTemporalStretch(prob=1.0, stretch_range=0.3),
                variations.MidiPitchShift(prob=0.5, shift_range=4)] if augment else None
        return CanonicalSpecWrapper(suite.load(
            "RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=4, n_envs=E, precision=64,
            task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                             primitive_fingertip_collisions=True, n_steps_lookahead=2,
                             augmentations=augs)))
    a = make()
    a.reset()
    if augment:
        a.task._t_idx[1] = int(a.task._song_len[1]) - 50   # this env ends (and re-draws) after the snapshot
    else:
        a.task._t_idx[1] = 158 - 50
    dev = a.physics.device
    act = lambda t: torch.as_tensor(actions[t % len(actions)], device=dev, dtype=torch.float64).expand(E, -1)
    for t in range(30):
        a.step(act(t))
    snap = a.state_dict()
    b = make()
    b.reset()
    b.load_state_dict(snap)
    n_first = 0
    for t in range(30, 70):
        ta, tb = a.step(act(t)), b.step(act(t))
        assert torch.equal(ta.step_type, tb.step_type), t
        assert torch.equal(ta.reward, tb.reward), t
        assert torch.equal(ta.discount, tb.discount), t
        for k in ta.observation:
            assert torch.equal(ta.observation[k], tb.observation[k]), (k, t)
        assert torch.equal(a.physics.qpos, b.physics.qpos) and torch.equal(a.physics.qvel, b.physics.qvel), t
        n_first += int((ta.step_type == 0).sum())
    assert n_first >= 1, "an episode boundary was crossed after the restore"
    if augment:
        assert torch.equal(a.task._song_len, b.task._song_len)


@pytest.mark.parametrize("side", ["right", "left"])
def test_one_hand_fused_task_kernels_match_torch_hooks(side):
    """rp_task_advance in its one-hand mode (rp_task_reward_args.hand_filter) against the torch
    restatement of piano_with_one_shadow_hand.py: every TimeStep field and the persistent task
    state, across episode ends and wrong-press terminations."""
    from robopianist_amd import music
    from robopianist_amd.suite import environment
    from robopianist_amd.suite.tasks import PianoWithOneShadowHand
    E = 5
    def make(fused):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            task = PianoWithOneShadowHand(midi=music.load("CMajorScaleTwoHands"), hand_side=side,
                                          control_timestep=0.05, gravity_compensation=True,
                                          primitive_fingertip_collisions=True, n_steps_lookahead=3,
                                          wrong_press_termination=True)
        if not fused:
            task._use_fused_advance = False
            task._use_fused_rewards = False
        return environment.Environment(task, n_envs=E, random_state=3, precision=64)
    fused, ref = make(True), make(False)
    fused.reset(); ref.reset()
    assert fused.task.fused_advance_for(fused.physics) is not None
    assert ref.task.fused_advance_for(ref.physics) is None
    spec = fused.action_spec()
    rng = np.random.RandomState(1)
    seen = dict(first=0, last=0, fingering=0.0)
    kj = torch.as_tensor(fused.task.piano.joints, device=fused.physics.device, dtype=torch.long)
    for step in range(220):
        a = rng.uniform(spec.minimum, spec.maximum, size=(E, spec.shape[0]))
        a[:, :22] = 0.3 * a[:, :22]
        if step % 37 == 20:   # a wrong key goes down in env 2: failure termination
            for env in (fused, ref):
                f = torch.zeros((E, env.physics.model.nv), device=env.physics.device, dtype=torch.float64)
                f[2, kj[3]] = 3.0
                env.physics.set_qfrc_applied(f)
        if step % 37 == 24:
            for env in (fused, ref):
                env.physics.set_qfrc_applied(torch.zeros((E, env.physics.model.nv), device=env.physics.device,
                                                         dtype=torch.float64))
        at = torch.as_tensor(a, device=fused.physics.device)
        ts_f, ts_r = fused.step(at), ref.step(at)
        assert torch.equal(ts_f.step_type, ts_r.step_type), step
        np.testing.assert_allclose(_np(ts_f.reward), _np(ts_r.reward), rtol=0, atol=1e-12, err_msg=str(step))
        np.testing.assert_allclose(_np(ts_f.discount), _np(ts_r.discount), rtol=0, atol=0)
        assert ts_f.observation.keys() == ts_r.observation.keys()
        for k in ts_f.observation:
            np.testing.assert_allclose(_np(ts_f.observation[k]), _np(ts_r.observation[k]), rtol=0, atol=1e-12,
                                       err_msg=f"{k} @ {step}")
        tf, tr = fused.task, ref.task
        for name in ("_t_idx", "_should_terminate", "_failure_termination", "_discount", "_goal_current",
                     "_finger_current", "_finger_next", "_fingering_state", "_goal_state"):
            assert torch.equal(getattr(tf, name), getattr(tr, name)), f"{name} @ {step}"
        for name in ("fingering_reward", "energy_reward", "key_press_reward", "sustain_reward"):
            np.testing.assert_allclose(_np(tf.reward_fn.reward_terms[name]), _np(tr.reward_fn.reward_terms[name]),
                                       rtol=0, atol=1e-12, err_msg=name)
        seen["first"] += int((ts_f.step_type == 0).sum())
        seen["last"] += int((ts_f.step_type == 2).sum())
        seen["fingering"] = max(seen["fingering"], float(tf.reward_fn.reward_terms["fingering_reward"].max()))
    assert seen["last"] >= E and seen["first"] >= E, seen
    assert seen["fingering"] > 0.0, "this hand's notes did enter the fingering term"


def test_evaluation_wrapper_fused_reduction_matches_torch():
    """MidiEvaluationWrapper (wrappers/evaluation.py:67-177): the reduction inside the
    rp_task_advance launch against the torch definition -- running sums, episode history ring
    and the reported metrics, over several episodes incl. keys pressed by applied torques."""
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper, MidiEvaluationWrapper
    E = 6
    def make(fused):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            env = suite.load("RoboPianist-debug-CMajorChordProgressionTwoHands-v0", seed=2, n_envs=E, precision=64,
                             task_kwargs=dict(control_timestep=0.05, gravity_compensation=True,
                                              primitive_fingertip_collisions=True))
        w = MidiEvaluationWrapper(CanonicalSpecWrapper(env), deque_size=2)
        w._use_fused = fused
        return w
    a, b = make(True), make(False)
    a.reset(); b.reset()
    assert a._fused() and not b._fused()
    rng = np.random.RandomState(0)
    dev = a.physics.device
    kj = torch.as_tensor(a.task.piano.joints, device=dev, dtype=torch.long)
    n_last = 0
    # the reference's own call (evaluation.py:139-141,167-169), per step and env, accumulated like the wrapper does:
    # pins the fused kernel (and the torch definition) to sklearn, corners included (no positives at all on most
    # steps, a wrong key without a goal in env 3, all-correct presses in envs 0-2)
    from sklearn.metrics import precision_recall_fscore_support
    sk_sums, sk_corners = np.zeros((E, 6)), set()
    for step in range(260):
        goal_rows = _np(b.task._goal_state[:, 0]) > 0
        act = torch.as_tensor(0.3 * rng.uniform(-1, 1, size=(E, 45)), device=dev)
        act[:, -1] = float(rng.uniform(-1, 1))   # sustain pedal toggles
        if step % 9 == 0:   # press the goal keys of env 0..2 (and a wrong key in env 3) with torques
            for w in (a, b):
                f = torch.zeros((E, w.physics.model.nv), device=dev, dtype=torch.float64)
                goal = w.task._goal_state[:, 0, :88] > 0
                for e in range(3):
                    f[e, kj[goal[e]]] = 3.0
                f[3, kj[10]] = 3.0
                w.physics.set_qfrc_applied(f)
        ta, tb = a.step(act), b.step(act)
        assert torch.equal(ta.step_type, tb.step_type)
        assert torch.equal(a._count, b._count), step
        keys_pred, sus_pred = _np(b.task.piano.activation), _np(b.task.piano.sustain_activation)
        first, last = _np(tb.first()), _np(tb.last())
        for e in range(E):
            if first[e]:
                continue
            vals = []
            for yt, yp in ((goal_rows[e, :-1], keys_pred[e]), (goal_rows[e, -1:], sus_pred[e].reshape(-1))):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    p, r, f, _ = precision_recall_fscore_support(y_true=yt, y_pred=yp > 0, average="binary", zero_division=1)
                vals += [p, r, f]
                sk_corners.add((bool((yt & (yp > 0)).any()), bool((~yt & (yp > 0)).any()), bool((yt & ~(yp > 0)).any())))
            sk_sums[e] += vals
        np.testing.assert_allclose(np.where(last[:, None], 0.0, sk_sums), _np(a._sums), rtol=0, atol=1e-12, err_msg=f"sklearn @ {step}")
        sk_sums[last] = 0.0
        np.testing.assert_allclose(_np(a._sums), _np(b._sums), rtol=0, atol=1e-12, err_msg=str(step))
        np.testing.assert_allclose(_np(a._hist), _np(b._hist), rtol=0, atol=1e-12, err_msg=str(step))
        assert torch.equal(a._n_finished, b._n_finished)
        n_last += int(ta.last().sum())
    assert n_last >= 2 * E
    ma, mb = a.get_musical_metrics(), b.get_musical_metrics()
    for k in ma:
        assert abs(ma[k] - mb[k]) < 1e-12, k
    assert 0.0 < ma["f1"] < 1.0 and ma["recall"] > 0.0, ma
    assert {(False, False, False), (False, True, False), (True, False, False)} <= sk_corners, sk_corners


def test_augmentation_prefetch_switches_bank_slots_without_host_reads():
    """`augmentation_prefetch=True`: every env owns two goal-bank slots; the fused launch
    switches to the prepared slot when an episode starts (include/rp_task.h next_ready /
    consumed) and the host refills the freed slot from an asynchronous snapshot of the flags.
    Checks: slots stay inside the env's pair, every episode start finds a table the host
    uploaded for that env, songs change from episode to episode, refills keep up."""
    from robopianist_amd import suite
    from robopianist_amd.suite import variations
    E = 16
    augs = [variations.MidiTemporalStretch(prob=1.0, stretch_range=0.3),
            variations.MidiPitchShift(prob=1.0, shift_range=6)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = suite.load("RoboPianist-debug-CMajorScaleOneHand-v0", seed=9, n_envs=E, precision=64,
                         task_kwargs=dict(control_timestep=0.05, gravity_compensation=True,
                                          primitive_fingertip_collisions=True, augmentations=augs,
                                          augmentation_prefetch=True))
    task = env.task
    uploads = {}
    orig = task._fill_slots
    def recording_fill(slots, envs):
        orig(slots, envs)
        for sl, e in zip(slots, envs):   # expected table: the host path on the MIDI just drawn for e
            uploads[int(sl)] = task._tables_for(task._env_midi[e])[0]
    task._fill_slots = recording_fill
    env.reset()
    assert task._device_rasterizer() is not None
    assert task.fused_advance_for(env.physics) is not None and task._goal_bank.shape[0] == 2 * E
    zero = torch.zeros((E, 45), device=env.physics.device, dtype=torch.float64)
    lens_seen = [set() for _ in range(E)]
    starts = np.zeros(E, int)
    stale = 0
    prev_slot = task._song_id.clone().cpu().numpy()
    for step in range(700):
        ts = env.step(zero)
        first = ts.first().cpu().numpy()
        if first.any():
            sid = task._song_id.cpu().numpy()
            assert (sid // 2 == np.arange(E)).all()
            for e in np.flatnonzero(first):
                g = uploads[int(sid[e])]
                n = int(task._song_len[sid[e]])
                assert n == len(g)
                np.testing.assert_array_equal(_np(task._goal_bank[sid[e], :n]), g)
                np.testing.assert_array_equal(_np(ts.observation["goal"][e][:89]), g[0])
                lens_seen[e].add(n)
                starts[e] += 1
                stale += int(sid[e] == prev_slot[e])   # host was late: the env replays its tables
            prev_slot = sid.copy()
    assert starts.min() >= 3
    assert sum(len(s) >= 2 for s in lens_seen) >= E - 2, lens_seen
    assert stale == 0, "the refills did not keep up"
    assert task.prefetch_refills >= starts.sum() - 2 * E


@pytest.mark.parametrize("precision", [64, 32])
def test_device_rasterizer_matches_the_host_tables(precision):
    """rp_task_rasterize against NoteTrajectory.from_midi(...).to_goal_tables() (the note-object
    path that restates midi_file.py:315-362 / piano_roll.py:59-204): every library song, random
    stretch / transpose chains, two control timesteps, with and without initial buffer time --
    goal rows, fingering rows and lengths bit-equal."""
    from robopianist_amd import music, task_kernels
    from robopianist_amd.music import midi_file
    from robopianist_amd.suite import variations
    dev = torch.device("cuda", 0)
    dt_t = torch.float64 if precision == 64 else torch.float32
    bases = [music.load(n) for n in music.ALL]
    rs = np.random.RandomState(1)
    augs = [variations.MidiTemporalStretch(1.0, 0.4), variations.MidiPitchShift(1.0, 9),
            variations.MidiOctaveShift(0.5, 2), variations.MidiTemporalStretch(0.5, 0.1)]
    for dt, buf in ((0.05, 0.0), (0.013, 0.5)):
        r = task_kernels.Rasterizer(dev, dt_t, [b.note_arrays() for b in bases], dt, buf)
        jobs = []
        for rep in range(3):
            for si, base in enumerate(bases):
                m = base
                for v in augs:
                    m = v(initial_value=m, random_state=rs)
                assert m._base is base._base
                jobs.append((si, m))
        want = []
        for si, m in jobs:
            t = midi_file.NoteTrajectory.from_midi(m, dt)
            t.add_initial_buffer_time(buf)
            want.append(t.to_goal_tables())
        cap = max(len(g) for g, _ in want) + 3
        n = len(jobs)
        goal = torch.full((n, cap, 89), 7.0, device=dev, dtype=dt_t)      # garbage: the kernel must overwrite
        finger = torch.full((n, cap, 88), 5, device=dev, dtype=torch.int64)
        lens = torch.zeros(n, dtype=torch.int64, device=dev)
        ops = [list(m._ops[len(bases[si]._ops):]) for si, m in jobs]
        st = r.rasterize(goal, finger, lens, np.arange(n), [si for si, _ in jobs], ops)
        assert st.tolist() == [0] * n
        for j, (g, f) in enumerate(want):
            L = len(g)
            assert int(lens[j]) == L, (j, int(lens[j]), L)
            np.testing.assert_array_equal(_np(goal[j, :L]), g, err_msg=f"goal of job {j}")
            np.testing.assert_array_equal(_np(finger[j, :L]), f, err_msg=f"finger of job {j}")
            assert float(goal[j, L:].abs().sum()) == 0.0 and int((finger[j, L:] != -1).sum()) == 0
        # a bank that is too short is reported, not overrun
        small = torch.zeros((n, 8, 89), device=dev, dtype=dt_t)
        st = r.rasterize(small, torch.zeros((n, 8, 88), device=dev, dtype=torch.int64), lens,
                         np.arange(n), [si for si, _ in jobs], ops)
        assert st.tolist() == [1] * n and float(small.abs().sum()) == 0.0


def test_uniformly_random_actions_do_not_diverge():
    """BASELINE config 3's policy (i.i.d. uniform actions every step) at 8192 envs: hands
    swing into each other at ~18 rad/s and pile up more contacts / Jacobian entries than the
    kernels held in rounds 1-3 (32 / 256).  No env may diverge, and hardly any may overflow."""
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper
    E = 8192
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = CanonicalSpecWrapper(suite.load(
            "RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=E, n_envs=E, precision=64,
            task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                             primitive_fingertip_collisions=True)))
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(E)
    ever = torch.zeros(E, dtype=torch.int32, device="cuda")
    for _ in range(120):
        a = torch.rand((E, 45), generator=g, device="cuda", dtype=torch.float64) * 2 - 1
        ts = env.step(a)
        ever |= env.physics.warn
        assert torch.isfinite(ts.reward).all()
    assert int((ever & 1).sum()) == 0, "diverged envs"
    assert int((ever & 4).sum()) == 0, "clamped Hessian pivots"
    # capacity (round 4: 64 contacts / 640 contact Jacobian entries per env): overflows are rare now -- at most 1e-5 of
    # the env-steps (rounds 2-3: ~3e-4, and those episodes were ended); the overflow path itself is exercised by
    # tests/test_gpu_parity.py::test_contact_capacity_overflow_keeps_the_deepest_contacts
    assert int((ever & 2).sum()) <= 1e-5 * E * 120, int((ever & 2).sum())
    assert torch.isfinite(env.physics.qpos).all()


def test_rollouts_are_bitwise_reproducible_and_batch_invariant():
    """Identical envs stay bitwise identical (the LDS-add reductions of the solver have a
    fixed order), a re-run reproduces the same bits, and an env's trajectory does not depend
    on how many other envs share the launch."""
    import os
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper
    actions = np.load(os.path.join(os.path.dirname(__file__), "golden", "twinkle_twinkle_actions.npy"))

    def run(n_envs, precision):
        env = CanonicalSpecWrapper(suite.load(
            "RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=1, n_envs=n_envs, precision=precision,
            task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                             primitive_fingertip_collisions=True)))
        env.reset()
        a = torch.as_tensor(actions, device=env.physics.device, dtype=env.physics.dtype)
        for t in range(60):
            env.step(a[t].expand(n_envs, -1))
        return env.physics.qpos.clone(), env.physics.qvel.clone()

    for precision in (64, 32):
        q1, v1 = run(96, precision)
        q2, v2 = run(96, precision)
        q3, v3 = run(3, precision)
        assert torch.equal(q1, q1[0:1].expand_as(q1)) and torch.equal(v1, v1[0:1].expand_as(v1))
        assert torch.equal(q1, q2) and torch.equal(v1, v2)
        assert torch.equal(q1[:3], q3) and torch.equal(v1[:3], v3)


def test_build_self_check_runs_once_and_passes(two_hand_scene, capfd, monkeypatch):
    """engine._build_self_check: fp32 and fp64 kernels agree to single precision on the
    shipped library; it is cached per (library, model, device)."""
    from robopianist_amd import engine
    monkeypatch.setenv("RP_SELF_CHECK_VERBOSE", "1")
    monkeypatch.delenv("RP_SKIP_SELF_CHECK", raising=False)
    engine._self_checked.clear()
    si = two_hand_scene
    engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=3)
    out = capfd.readouterr().out
    assert out.count("rp self-check") == 1
    dq = float(out.split("max|dq| = ")[1].split(",")[0])
    assert dq < 2e-6
    engine.BatchedPhysics(si.model, si.key_joint_ids, n_envs=3)   # cached: no second run
    assert "rp self-check" not in capfd.readouterr().out


def test_ot_fingering_reward_on_device_matches_scipy():
    """include/rp_task.h use_fingering = 2: the optimal-transport fingering term
    (piano_with_shadow_hands.py:333-369) as a wave-per-env assignment kernel against the host
    definition (scipy.optimize.linear_sum_assignment per env), incl. more keys than fingers, one
    key, no key, and through the fused advance on a rollout."""
    from robopianist_amd import suite
    from robopianist_amd.wrappers import CanonicalSpecWrapper
    name = "RoboPianist-debug-CMajorChordProgressionTwoHands-v0"
    E = 64
    env = CanonicalSpecWrapper(suite.load(
        name, seed=5, n_envs=E, precision=64,
        task_kwargs=dict(control_timestep=0.05, gravity_compensation=True, primitive_fingertip_collisions=True,
                         disable_fingering_reward=True)))
    task = env.task
    assert "ot_fingering_reward" in task.reward_fn.reward_fns
    env.reset()
    rng = np.random.RandomState(0)
    dev = env.physics.device
    A = env.action_spec().shape[0]
    # (1) through the rollout: fused advance vs the host definition on the same state
    for t in range(12):
        ts = env.step(torch.as_tensor(rng.uniform(-1, 1, size=(E, A)), device=dev))
        assert task._fused_advance is not None
        fused = task.reward_fn.reward_terms["ot_fingering_reward"].clone()
        ref = task._compute_ot_fingering_reward(env.physics)
        np.testing.assert_allclose(_np(fused), _np(ref), rtol=0, atol=1e-12, err_msg=str(t))
    # (2) synthetic goal sets of every size 0..30 keys, random key states / hand poses
    fr = task._fused_rewards_for(env.physics)
    sizes = [0, 1, 2, 3, 5, 9, 10, 11, 12, 17, 30, 88]
    for trial in range(3):
        goal = torch.zeros((E, 89), device=dev, dtype=torch.float64)
        for e in range(E):
            k = sizes[(e + trial) % len(sizes)]
            idx = rng.choice(88, size=k, replace=False)
            goal[e, torch.as_tensor(idx, device=dev, dtype=torch.long)] = 1.0
        task._goal_current.copy_(goal)
        env.step(torch.as_tensor(rng.uniform(-1, 1, size=(E, A)), device=dev))  # new hand poses
        task._goal_current.copy_(goal)
        total, terms = fr.compute(goal_current=task._goal_current, key_norm_state=task.piano.normalized_state,
                                  key_activation=task.piano.activation, sustain_activation=task.piano.sustain_activation,
                                  finger_current=task._finger_current)
        ref = task._compute_ot_fingering_reward(env.physics)
        np.testing.assert_allclose(_np(terms[3]), _np(ref), rtol=0, atol=1e-12)
        assert float(ref[goal[:, :88].sum(1) == 0].min()) == 1.0 if bool((goal[:, :88].sum(1) == 0).any()) else True


def test_capacity_overflow_ends_the_episode():
    """With `overflow_termination=True` (extension) an engine capacity overflow (RP_WARN_CONTACT_FULL etc.) ends
    the episode with reward 0 / discount 0 like a diverged state, on the fused path and on the torch hooks alike;
    by default the env steps on, as the reference's would (piano_with_shadow_hands.py:212-220 names the only
    termination causes), and the episode is counted in `overflow_episodes` when it ends."""
    from robopianist_amd import engine
    fused, ref = _load_pair(4, 64, overflow_termination=True)
    loose, _ = _load_pair(4, 64)
    dev = fused.physics.device
    A = fused.action_spec().shape[0]
    a = torch.zeros((4, A), device=dev, dtype=torch.float64)
    for env in (fused, ref, loose):
        env.reset()
        env.step(a)
        env.physics.warn[2] |= engine.WARN_CONTACT_FULL      # as the position stage would raise it
        env.physics.warn[3] |= engine.WARN_HESSIAN           # not fatal: a clamped pivot is reported only
    ts_f, ts_r, ts_l = fused.step(a), ref.step(a), loose.step(a)
    for ts in (ts_f, ts_r):
        assert ts.step_type.tolist() == [1, 1, 2, 1]
        assert float(ts.reward[2]) == 0.0 and float(ts.discount[2]) == 0.0
    assert ts_l.step_type.tolist() == [1, 1, 1, 1]
    assert fused.task.overflow_terminations() == 1 and ref.task.overflow_terminations() == 1
    assert fused.task.overflow_episodes() == 1 and ref.task.overflow_episodes() == 1
    assert loose.task.overflow_terminations() == 0 and loose.task.overflow_episodes() == 0   # (its episode is still running)
    ts_f = fused.step(a)
    assert ts_f.step_type.tolist() == [1, 1, 0, 1]          # auto-reset clears the flag
    assert int(fused.physics.warn[2]) == 0
