"""Runs ON THE GPU BOX after tools/gpu/r06_profiles.sh: reduces the rocprofv3 outputs to the small summaries
that go into profiles/r06_* (the raw traces are > 64 MiB)."""
import glob, json, os, shutil, sys
import pandas as pd
R = sys.argv[1]
OUT = os.path.join(R, "summary"); os.makedirs(OUT, exist_ok=True)
LEAN, FULL, POS = "rp_lean_solver_kernel<double>", "rp_stage_kernel<double, 1", "rp_stage_kernel<double, 0"
out = {"round": 6}
try:
    ks = pd.read_csv(sorted(glob.glob(R + "/stats/*/*kernel_stats.csv"))[-1])
    ks["Name"] = ks["Name"].str.slice(0, 140)
    ks.to_csv(os.path.join(OUT, "r06_kernel_stats.csv"), index=False)
    kt = pd.read_csv(sorted(glob.glob(R + "/stats/*/*kernel_trace.csv"))[-1])
    kt["dur"] = kt.End_Timestamp - kt.Start_Timestamp
    gcol = "Grid_Size_X" if "Grid_Size_X" in kt.columns else ("Grid_Size" if "Grid_Size" in kt.columns else None)
    def sel(pat): return kt[kt.Kernel_Name.str.contains(pat, regex=False)]
    lean, full, pos, task, order = sel(LEAN), sel(FULL), sel(POS), sel("rp_task_"), sel("rp_order_kernel")
    fusedk, cleank = sel("rp_fused_steps_kernel"), sel("rp_cleanup_steps_kernel")
    frontk, narrowk, backk = sel("rp_pos_front_kernel"), sel("rp_narrow_kernel"), sel("rp_pos_back_kernel")
    other = kt[~kt.Kernel_Name.str.contains("rp_stage_kernel|rp_lean_solver|rp_fused_steps|rp_cleanup_steps|rp_task_|rp_reset|rp_order|rp_lead|rp_mark|rp_pos_front|rp_pos_back|rp_narrow|rp_pos_list", regex=True)]
    nstep = max(1, len(task))
    cfg = lambda df: {k: str(df.iloc[0][k]) for k in ["LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"] if k in df.columns} if len(df) else None
    def by_grid(df):
        if gcol is None or not len(df): return None
        res = {}
        for g, x in df.groupby(gcol):
            envs = int(g) // 64 if int(g) >= 64 * 8 else int(g)
            res["%d envs per launch" % envs] = {"launches": int(len(x)), "avg_us": float(x.dur.mean() / 1e3), "share_of_gpu_time": float(x.dur.sum() / kt.dur.sum())}
        return res
    def block(df, big=None):
        d = df if big is None else df[df.dur > big]
        return {"launches": int(len(d)), "avg_us": float(d.dur.mean() / 1e3) if len(d) else None, "min_us": float(d.dur.min() / 1e3) if len(d) else None,
                "max_us": float(d.dur.max() / 1e3) if len(d) else None, "share_of_gpu_time": float(df.dur.sum() / kt.dur.sum()),
                "by_launch_size": by_grid(d), "launch_config": cfg(d)}
    out.update({
        "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 158 --warmup 20  (fp64, config 2, hull fingertips = the reference's default, 4096 envs, staggered episode phases, full env.step, schedule (stream slices / fused substeps) chosen by the engine; the trace also holds the untimed prologue and the lockstep aux leg)",
        "kernels": {
            "rp_lean_solver_kernel<double> (solver stage of the light envs, dominant)": block(lean),
            "rp_stage_kernel<double, 1, 4, 9> (full-capacity solver stage: the envs outside the light class; empty launches exit at once)": block(full),
            "rp_stage_kernel<double, 0, 0, 9, 1> (position/velocity stage, hull build)": dict(block(pos, 30000), masked_forward_launches=int((pos.dur <= 30000).sum())),
            "rp_pos_front_kernel<double, 1> (split position stage, front part: kinematics, CRB, broad phase, prefilters; 13 KB LDS, 155 VGPRs)": block(frontk),
            "rp_narrow_kernel<double, 1> (split position stage: POOLED narrow phase, one wave per 64 candidate pairs of one type)": block(narrowk),
            "rp_pos_back_kernel<double, 1> (split position stage, back part: contacts -> constraint rows, Jacobians, velocity stage)": block(backk),
            "rp_fused_steps_kernel<double, 1> (fused schedule: all substeps of a step in one launch; trial steps of the schedule choice, the spread prologue and the lockstep aux leg)": block(fusedk),
            "rp_cleanup_steps_kernel<double, 1> (envs that left the light class under the fused schedule)": block(cleank),
            "rp_order_kernel (cost-ordered launch + compaction of the envs outside the light class)": {"launches": int(len(order)), "avg_us": float(order.dur.mean() / 1e3) if len(order) else None, "share_of_gpu_time": float(order.dur.sum() / kt.dur.sum())},
            "rp_task_advance_kernel<double> (fused task layer)": {"launches": int(len(task)), "avg_us": float(task.dur.mean() / 1e3) if len(task) else None, "share_of_gpu_time": float(task.dur.sum() / kt.dur.sum())},
            "torch kernels (action gather / scaling, ctrl scatter, masks, output copies)": {"launches_per_step": float(len(other) / nstep), "share_of_gpu_time": float(other.dur.sum() / kt.dur.sum())}},
        "note": "kernel durations overlap when the engine steps two slices on two streams: shares are of the summed kernel time, not of wall time"})
except Exception as e:
    out["kernel_trace_error"] = repr(e)
pm = {}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = sorted(glob.glob(f"{R}/{d}/*/*counter_collection.csv"))
    if not f: continue
    df = pd.read_csv(f[-1]); df = df[df.Counter_Name == name]
    for tag, pat in (("solver", LEAN), ("position", POS)):
        x = df[df.Kernel_Name.str.contains(pat, regex=False)]
        x = x[x.Counter_Value > x.Counter_Value.max() * 0.05]
        pm[f"{name}_KB_per_launch_{tag}"] = float(x.Counter_Value.mean()); pm[f"n_{name}_{tag}"] = int(len(x))
if "FETCH_SIZE_KB_per_launch_solver" in pm and "WRITE_SIZE_KB_per_launch_solver" in pm:
    sol_b = pm["FETCH_SIZE_KB_per_launch_solver"] * 1024 * 2 + pm["WRITE_SIZE_KB_per_launch_solver"] * 1024
    pos_b = pm["FETCH_SIZE_KB_per_launch_position"] * 1024 * 2 + pm["WRITE_SIZE_KB_per_launch_position"] * 1024
    pm["note"] = ("separate --pmc passes (bench.py --stagger 0 --steps 4 --warmup 1, RP_STREAM_SLICES=1: one launch = one substep of all 4096 envs, "
                  "first control steps of the lockstep replay, hull fingertips).  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 "
                  "counts 64 B per 128-B request, so the read side is doubled; WRITE_SIZE is used as reported (uncalibrated).  The counters "
                  "sit on the L2's fabric side: Infinity-Cache hits are included, i.e. an upper bound on HBM traffic.")
    pm["solver_kernel_bytes_per_launch_corrected"] = sol_b; pm["position_kernel_bytes_per_launch_corrected"] = pos_b
    out["pmc"] = pm
    json.dump({"envs": 4096, "envs_per_launch": 4096, "precision": 64, "solver_kernel_bytes_per_launch": sol_b, "position_kernel_bytes_per_launch": pos_b,
               "kernel": LEAN}, open(os.path.join(OUT, "traffic_r06.json"), "w"))
elif pm:
    out["pmc_partial"] = pm
# the split position stage's kernels (separate passes with RP_SPLIT_POS=1)
sp = {}
for name, d in (("FETCH_SIZE", "fetch_split"), ("WRITE_SIZE", "write_split")):
    f = sorted(glob.glob(f"{R}/{d}/*/*counter_collection.csv"))
    if not f: continue
    df = pd.read_csv(f[-1]); df = df[df.Counter_Name == name]
    for tag, pat in (("front", "rp_pos_front_kernel"), ("narrow", "rp_narrow_kernel"), ("back", "rp_pos_back_kernel")):
        x = df[df.Kernel_Name.str.contains(pat, regex=False)]
        if len(x): x = x[x.Counter_Value > x.Counter_Value.max() * 0.05]
        sp[f"{name}_KB_per_launch_{tag}"] = float(x.Counter_Value.mean()) if len(x) else None
if sp and all(v is not None for v in sp.values()):
    tot = {t: sp[f"FETCH_SIZE_KB_per_launch_{t}"] * 1024 * 2 + sp[f"WRITE_SIZE_KB_per_launch_{t}"] * 1024 for t in ("front", "narrow", "back")}
    sp["bytes_per_launch_corrected"] = tot
    sp["split_stage_total_bytes_per_4096_env_mj_step1"] = sum(tot.values())
    out["pmc_split_position_stage"] = sp
    try:
        t5 = json.load(open(os.path.join(OUT, "traffic_r06.json")))
        t5["split_position_stage_bytes_per_launch"] = tot
        json.dump(t5, open(os.path.join(OUT, "traffic_r06.json"), "w"))
    except Exception:
        pass
sq = {}
for d in ("sq1", "sq2", "sq3"):
    f = sorted(glob.glob(f"{R}/{d}/*/*counter_collection.csv"))
    if not f: continue
    df = pd.read_csv(f[-1])
    for tag, pat in (("solver " + LEAN, LEAN), ("position rp_stage_kernel<double, 0, 0, 9, 1>", POS)):
        x = df[df.Kernel_Name.str.contains(pat, regex=False)]
        big = x.groupby("Dispatch_Id").Counter_Value.sum(); x = x[x.Dispatch_Id.isin(big[big > big.max() * 0.05].index)]
        per = x.groupby("Counter_Name").Counter_Value.mean()
        sq.setdefault(tag, {"per_launch": {}, "per_wave": {}})
        for k, v in per.items():
            sq[tag]["per_launch"][k] = float(v); sq[tag]["per_wave"][k] = float(v) / 4096.0
if sq:
    doc = {"round": 6, "note": "rocprofv3 --pmc, separate passes, bench.py --stagger 0 --steps 4 --warmup 1 with one stream slice (config 2, hull fingertips, lockstep, fp64, first five control steps); per wave = per launch / 4096 envs; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY count quad-cycles",
           "kernels": sq}
    k = sq.get("solver " + LEAN, {}).get("per_wave", {})
    if k.get("SQ_INSTS_VALU"):
        f64 = sum(k.get(c, 0.0) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
        res = None
        try:
            res = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lean_resource_usage.json")))
        except Exception:
            pass
        doc["solver_stage_fp64"] = dict({"waves_per_simd": 2, "fp64_math_share_of_valu": f64 / k["SQ_INSTS_VALU"] if f64 else None,
                                         "issue_share_of_wave_cycles": (k.get("SQ_ACTIVE_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"]) if k.get("SQ_WAVE_CYCLES") else None,
                                         "valu_per_wave": k["SQ_INSTS_VALU"], "salu_per_wave": k.get("SQ_INSTS_SALU"), "lds_per_wave": k.get("SQ_INSTS_LDS")},
                                        **(res or {}))
    json.dump(doc, open(os.path.join(OUT, "r06_sq_instruction_mix.json"), "w"), indent=1)
json.dump(out, open(os.path.join(OUT, "r06_step_kernel_summary.json"), "w"), indent=1)
for f in glob.glob(R + "/bench_*.json") + glob.glob(R + "/bench_*.err") + glob.glob(R + "/*.log"):
    shutil.copy(f, OUT)
print(json.dumps({k: v for k, v in out.items() if k != "command"}, indent=1)[:3000])
