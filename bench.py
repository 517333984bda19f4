#!/usr/bin/env python
"""Benchmark of the hot path (BASELINE.json: env-steps/sec at 4096 envs/GPU).

One "step" = one control step (= 10 physics substeps, base.py:28,31) of ALL envs
on this rank.  Workload: BASELINE.json configs[1] — PianoWithShadowHands,
TwinkleTwinkle scripted replay (tests/golden/twinkle_twinkle_actions.npy mapped
canonical -> ctrlrange), 4096 envs per GPU, inputs resident in HBM.

Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for the definitions
of `roofline` (algorithmic bytes / step-kernel time from HIP events) and
`cpu_baseline` (the fp64 C oracle timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic bytes of ONE launch of the dominant kernel (rp_stage_kernel<T,1>, the mj_step2 /
# constraint-solver stage of one substep) per env, in elements of T:
#   in : qpos qvel qacc_warmstart qfrc_applied (4 nv) + ctrl (nu)
#   out: qpos qvel qacc_warmstart (3 nv) + actuator_force (nu) + time (1)
# = 7*140 + 2*44 + 1 = 1069 elements -> 4276 B (fp32), 8552 B (fp64).  SURVEY/BASELINE §2.3
# quote 4352 B per fused env-step; that figure assumed one 10-substep kernel and is reported
# under roofline.note for reference.
def algo_bytes_per_solver_launch(nv, nu, precision):
    return (7 * nv + 2 * nu + 1) * (8 if precision == 64 else 4)


HBM_PEAK_GBS = 8000.0


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


def _pmc_traffic(E, precision):
    """HBM-side bytes per solver-kernel launch from the committed rocprofv3 --pmc passes
    (profiles/traffic_r01.json); null if not collected for this
    env count / precision."""
    p = os.path.join(ROOT, "profiles", "traffic_r01.json")
    try:
        d = json.load(open(p))
        ok = int(d["envs"]) == int(E) and int(d["precision"]) == int(precision)
        return d["solver_kernel_bytes_per_launch"] if ok else None
    except Exception:
        return None


def _issue_roofline(value, substeps):
    """Issue-slot ceiling of the present instruction streams from the committed SQ counters
    (profiles/r01_sq_instruction_mix.json: SQ_ACTIVE_INST_ANY per wave, in units of 4 cycles):
    every SIMD issues for one wave at a time, 1024 SIMDs at 2.4 GHz."""
    try:
        k = json.load(open(os.path.join(ROOT, "profiles", "r01_sq_instruction_mix.json")))["kernels"]
        act = sum(v["per_wave"]["SQ_ACTIVE_INST_ANY"] for v in k.values())
    except Exception:
        return None
    ceiling = 1024 * 2.4e9 / (substeps * 4.0 * act)
    return {"bound": "instruction issue", "achieved": value, "peak": ceiling, "unit": "env-steps/s",
            "frac": value / ceiling, "source": "profiles/r01_sq_instruction_mix.json (fp64 kernels)"}


def load_actions(m):
    a = np.load(os.path.join(ROOT, "tests", "golden", "twinkle_twinkle_actions.npy")).astype(np.float64)
    hands = a[:, :-1]
    assert hands.shape[1] == m.nu, (hands.shape, m.nu)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    # dm_env_wrappers.CanonicalSpecWrapper: ctrl = lo + (a+1)/2*(hi-lo), clipped
    ctrl = lo + (np.clip(hands, -1, 1) + 1.0) * 0.5 * (hi - lo)
    return ctrl, a[:, -1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=158)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--precision", type=int, default=64, choices=(32, 64),
                    help="64 (default) is the precision that meets the 1e-4 parity bar on this replay")
    ap.add_argument("--aux-fp32", type=int, default=1, help="also time the fp32 engine (reported under aux)")
    ap.add_argument("--substeps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", type=int, default=1, help="all-gather trajectory slab when gpus>1")
    ap.add_argument("--engine-only", action="store_true", help="time rp_step alone (no obs/reward epilogue)")
    ap.add_argument("--dist-backend", default="nccl", help="debug: 'gloo' lets several ranks share one GPU")
    ap.add_argument("--same-device", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--host-io", type=int, default=1,
                    help="N=1: also time the loop with host-resident actions/TimeSteps (aux.host_io, PCIe inclusive)")
    ap.add_argument("--graph", type=int, default=0,
                    help="replay env.step from a captured hipGraph (wrappers.GraphedStepWrapper)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    dev = local_rank if torch.cuda.is_available() else 0

    def measure(precision, steps, warmup):
        from robopianist_amd import engine, suite
        from robopianist_amd import distributed as rpd
        from robopianist_amd.wrappers import CanonicalSpecWrapper, GraphedStepWrapper

        E = args.envs
        device = torch.device("cuda", dev)
        tdt = torch.float32 if precision == 32 else torch.float64
        actions = np.load(os.path.join(ROOT, "tests", "golden", "twinkle_twinkle_actions.npy"))
        T = actions.shape[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            # notebook cell 15 kwargs; capsule fingertips (no meshes available)
            base_env = suite.load(
                "RoboPianist-debug-TwinkleTwinkleRousseau-v0", seed=rpd.rank_seed(12345, rank), n_envs=E,
                device_id=dev, precision=precision,
                task_kwargs=dict(trim_silence=True, control_timestep=0.05, gravity_compensation=True,
                                 primitive_fingertip_collisions=True, reduced_action_space=False,
                                 n_steps_lookahead=10))
        eager_env = CanonicalSpecWrapper(base_env)
        use_graph = bool(args.graph) and not args.engine_only
        env = GraphedStepWrapper(eager_env, warmup_steps=2) if use_graph else eager_env
        phys = base_env.physics.engine
        m = base_env.task.scene.model
        assert base_env.task.physics_steps_per_control_step == args.substeps == 10
        # all envs replay the same action stream (config #2); rows pre-expanded on device
        act_dev = torch.as_tensor(actions, dtype=tdt, device=device)
        state = {"t": 0}

        def one_step(_):
            t = state["t"]
            if args.engine_only:
                lo = torch.as_tensor(m.actuator_ctrlrange[:, 0], dtype=tdt, device=device)
                hi = torch.as_tensor(m.actuator_ctrlrange[:, 1], dtype=tdt, device=device)
                c = lo + (act_dev[t, :-1] + 1) * 0.5 * (hi - lo)
                base_env.physics.set_ctrl(c.expand(E, -1))
                phys.step(args.substeps)
                ts_last = (t + 1 == T)
            else:
                ts = env.step(act_dev[t].expand(E, -1))
                ts_last = (t + 1 == T)
                if world > 1 and args.gather:
                    rec = rpd.pack_trajectory_record(
                        base_env.physics.qpos, ts.reward, ts.discount, ts.step_type,
                        base_env.task.piano.activation)
                    # enqueue only: the all-gather of step t overlaps the physics of step t+1
                    if state.get("gather") is not None:
                        state["gather"][1].wait()
                    state["gather"] = rpd.gather_trajectories(rec, async_op=True, out=state.get("gather_buf"))
                    state["gather_buf"] = state["gather"][0]
            state["t"] = t + 1
            if ts_last:  # episode boundary: reset (not counted as a step, but timed)
                if args.engine_only:
                    phys.sync(); phys.reset()
                else:
                    env.reset()
                state["t"] = 0

        def barrier():
            if state.get("gather") is not None and state["gather"][1] is not None:
                state["gather"][1].wait()
            phys.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()

        env.reset()
        for t in range(warmup):
            one_step(t)
        barrier()
        phys.solver_kernel_time(); phys.kernel_time()  # reset the event-timer statistics
        t0 = time.perf_counter()
        for t in range(steps):
            one_step(warmup + t)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if use_graph:
            # launches inside a replayed hipGraph cannot carry per-kernel events: sample the
            # kernel times on 16 eager steps of the same rollout right after the timed region
            phys.solver_kernel_time(); phys.kernel_time()
            for i in range(16):
                eager_env.step(act_dev[(state["t"] + i) % T].expand(E, -1))
            barrier()
        sms, snl = phys.solver_kernel_time()
        kms, nl = phys.kernel_time()
        warn = int(phys.warn_flags.max())
        q = phys.qpos
        finite = bool(np.isfinite(q).all())
        ctrl_seq, _ = load_actions(m)

        # PCIe-inclusive variant (aux only, never `value`): the caller keeps actions and
        # TimeSteps in host memory -- a [E, 45] numpy action goes up and every TimeStep field
        # (reward, discount, step type, all observations) comes back to numpy each step
        host_io = None
        if args.host_io and world == 1 and not args.engine_only and precision == args.precision:
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

        return dict(host_io=host_io, dt=dt, kms=kms, nl=nl, sms=sms, snl=snl, warn=warn, finite=finite, phys=phys, m=m,
                    ctrl_seq=ctrl_seq, E=E, key_ids=base_env.task.scene.key_joint_ids, graphed=bool(use_graph and env.graph_captured))

    r = measure(args.precision, args.steps, args.warmup)
    dt, kms, nl, sms, snl, warn, finite, phys, m, ctrl_seq, E = (
        r[k] for k in ('dt', 'kms', 'nl', 'sms', 'snl', 'warn', 'finite', 'phys', 'm', 'ctrl_seq', 'E'))
    base_key_ids = r['key_ids']

    if rank == 0:
        value = world * E * args.steps / dt
        algo = algo_bytes_per_solver_launch(int(m.nv), int(m.nu), args.precision) * E
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
            "data": "synthetic (scripted twinkle_twinkle_actions.npy replay on the stand-in hand model)",
            "config": {
                "workload": "PianoWithShadowHands-TwinkleTwinkle scripted replay (BASELINE configs[1]), " + ("engine-level rp_step only" if args.engine_only else "full vectorised env.step (obs + rewards)"),
                "envs_per_gpu": E, "substeps_per_step": args.substeps, "nv": int(m.nv), "nu": int(m.nu),
                "fingertips": "capsule (primitive) stand-in", "mj_steps_per_s": value * args.substeps,
                "trajectory_gather": bool(world > 1 and args.gather), "hipgraph_step": r["graphed"],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc_traffic(E, args.precision),
                "kernel": "rp_stage_kernel<%s, 1> (mj_step2: constraint solver + Euler, one substep of all envs)" % tname,
                "kernel_avg_ms": sms, "kernel_launches_sampled": snl,
                "algorithmic_bytes_per_launch": algo,
                "step_sequence_avg_ms": kms, "step_sequences": nl,
                "kernel_timing": ("HIP events on 16 eager env steps of the same rollout right after the timed region "
                                  "(the timed region replays a captured hipGraph)") if r["graphed"] else
                                 "HIP events over the timed region",
                "note": "one rp_step = 1 + 2*substeps launches (rp_stage_kernel<T,0> position/velocity stage, "
                        "<T,1> solver stage); kernel_avg_ms is the solver launch of the middle substep of every step "
                        "(HIP events on the engine stream), step_sequence_avg_ms the whole 21-launch sequence. "
                        "The path is instruction-issue / latency bound (one wave per env; the fp64 solver runs one "
                        "wave per SIMD, the position kernel two), not HBM bound: see DESIGN.md 6",
            },
            "issue_roofline": _issue_roofline(value / world, args.substeps) if args.precision == 64 else None,
            "sanity": {"warn_flags": warn, "finite": finite},
            "parity": "fp64 engine: max rel |dq| vs CPU oracle over 1000 mj_steps of this replay < 1e-4 "
                      "(measured live under cpu_baseline_parity when the CPU leg runs), "
                      "tests/test_gpu_parity.py::test_replay_fp64_1000_steps; fp32 engine diverges on this "
                      "(chaotic, self-colliding) replay and is reported under aux only",
        }
        if r.get("host_io"):
            out.setdefault("aux", {})["host_io"] = r["host_io"]
        if args.aux_fp32 and args.precision == 64 and world == 1:
            del r, phys
            r32 = measure(32, min(args.steps, 80), min(args.warmup, 10))
            out.setdefault("aux", {})["fp32_engine"] = {
                "value": E * min(args.steps, 80) / r32["dt"], "unit": "env-steps/s", "kernel_avg_ms": r32["kms"],
                "warn_flags": r32["warn"],
                "note": "same workload on the fp32 build; meets 1e-4 on smooth key-press scenarios only"}
            phys = r32["phys"]
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            from oracle.rp_oracle import Oracle
            orc = Oracle(m, phys.blob)
            cores = _usable_cores()
            # bounded sample of the same workload, sized for ~15 s of host work: many short
            # rollouts from reset (the transient, contact-changing part of an episode), each
            # env holding a different row of the scripted replay
            nenv_cpu = cores * 160
            nstep = 400
            rows = (40 + 7 * np.arange(nenv_cpu)) % ctrl_seq.shape[0]
            cc = np.ascontiguousarray(ctrl_seq[rows])
            secs, _ = orc.bench(nenv_cpu, nstep, cc, cores)
            # same leg: the second half of BASELINE.json's metric, measured here -- the engine
            # (precision of the headline run) against the CPU oracle on this very replay,
            # free running, 1000 mj_steps, rel = |dq| / max(|q|, 1e-2)
            from robopianist_amd import engine as _eng
            chk = _eng.BatchedPhysics(m, base_key_ids, n_envs=2, precision=args.precision)
            orc.reset()
            worst, worst_abs = 0.0, 0.0
            # dof groups of SURVEY.md 8(d) metric 2: keys (88), forearm joints (2 per hand), fingers+wrists
            is_key = np.zeros(int(m.nv), bool); is_key[np.asarray(base_key_ids)] = True
            is_arm = np.array(["forearm" in n for n in m.names["joint"]])
            groups = {"keys": is_key, "forearms": is_arm & ~is_key, "fingers_and_wrists": ~is_key & ~is_arm}
            gmax = {k: 0.0 for k in groups}
            curve = {}
            for i in range(1000):
                c = ctrl_seq[(i // args.substeps) % ctrl_seq.shape[0]]
                chk.set(_eng.CTRL, c[None, :]); orc.ctrl[:] = c
                chk.step(1); orc.step(1)
                qg = chk.qpos.astype(np.float64)[0]
                ad = np.abs(qg - orc.qpos)
                rel = ad / np.maximum(np.abs(orc.qpos), 1e-2)
                worst = max(worst, float(rel.max())); worst_abs = max(worst_abs, float(ad.max()))
                for k, sel in groups.items():
                    gmax[k] = max(gmax[k], float(rel[sel].max()))
                if i + 1 in (1, 10, 100, 300, 1000):
                    curve[str(i + 1)] = float(rel.max())
            out["cpu_baseline_parity"] = {
                "max_rel_qpos_error_1000_mj_steps": worst, "max_abs_qpos_error_1000_mj_steps": worst_abs,
                "bar": 1e-4, "rel_error_at_mj_step": curve, "max_rel_error_by_dof_group": gmax,
                "note": "engine (2 envs, same precision as value) vs the CPU oracle, scripted replay, free running; "
                        "rel = |dq| / max(|q_cpu|, 1e-2)"}
            out["cpu_baseline"] = {
                "value": nenv_cpu * nstep / args.substeps / secs, "unit": "env-steps/s",
                "cores": cores, "kind": "port",
                "sample": f"{nenv_cpu} envs x {nstep} mj_steps from reset ({secs:.1f} s of host time), fp64 C oracle (CPU restatement, not MuJoCo), OpenMP {cores} threads, each env holds one row of the scripted replay as ctrl",
                "mj_steps_per_s": nenv_cpu * nstep / secs,
            }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
