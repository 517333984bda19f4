"""Runs ON THE GPU BOX after scratch/collect_profiles_r02b.sh: reduces the rocprofv3 outputs under
gpurun_out/prof_r02b to the small summaries that go into profiles/r02_* (the raw traces are > 64 MiB)."""
import glob, json, os, shutil, sys
import pandas as pd
R = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r02b"
OUT = os.path.join(R, "summary"); os.makedirs(OUT, exist_ok=True)
out = {"round": 2}
try:
    ks = pd.read_csv(sorted(glob.glob(R + "/stats/*/*kernel_stats.csv"))[-1])
    ks["Name"] = ks["Name"].str.slice(0, 140)
    ks.to_csv(os.path.join(OUT, "r02_kernel_stats.csv"), index=False)
    kt = pd.read_csv(sorted(glob.glob(R + "/stats/*/*kernel_trace.csv"))[-1])
    kt["dur"] = kt.End_Timestamp - kt.Start_Timestamp
    gcol = "Grid_Size_X" if "Grid_Size_X" in kt.columns else ("Grid_Size" if "Grid_Size" in kt.columns else None)
    def sel(pat): return kt[kt.Kernel_Name.str.contains(pat, regex=False)]
    sol, pos, task, order = sel("rp_stage_kernel<double, 1"), sel("rp_stage_kernel<double, 0"), sel("rp_task_"), sel("rp_order_kernel")
    other = kt[~kt.Kernel_Name.str.contains("rp_stage_kernel|rp_task_|rp_reset|rp_order|rp_lead|rp_mark", regex=True)]
    nstep = max(1, len(task))
    cfg = lambda df: {k: str(df.iloc[0][k]) for k in ["LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"] if k in df.columns}
    def by_grid(df):
        if gcol is None: return None
        res = {}
        for g, x in df.groupby(gcol):
            envs = int(g) // 64 if int(g) >= 64 * 8 else int(g)
            res["%d envs per launch" % envs] = {"launches": int(len(x)), "avg_us": float(x.dur.mean() / 1e3), "share_of_gpu_time": float(x.dur.sum() / kt.dur.sum())}
        return res
    posfull = pos[pos.dur > 30000]
    out.update({
        "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --steps 158 --warmup 20  (fp64, config 2, 4096 envs, staggered episode phases, full env.step, stream slices chosen by the engine; the trace also holds the untimed prologue and the lockstep aux leg)",
        "kernels": {
            "rp_stage_kernel<double, 1, 4, 9> (solver stage, dominant)": {"launches": int(len(sol)), "avg_us": float(sol.dur.mean() / 1e3), "min_us": float(sol.dur.min() / 1e3), "max_us": float(sol.dur.max() / 1e3), "share_of_gpu_time": float(sol.dur.sum() / kt.dur.sum()), "by_launch_size": by_grid(sol), "launch_config": cfg(sol)},
            "rp_stage_kernel<double, 0, 0, 9> (position/velocity stage)": {"launches": int(len(posfull)), "avg_us": float(posfull.dur.mean() / 1e3), "masked_forward_launches": int(len(pos) - len(posfull)), "share_of_gpu_time": float(pos.dur.sum() / kt.dur.sum()), "by_launch_size": by_grid(posfull), "launch_config": cfg(posfull)},
            "rp_order_kernel (cost-ordered launch)": {"launches": int(len(order)), "avg_us": float(order.dur.mean() / 1e3) if len(order) else None, "share_of_gpu_time": float(order.dur.sum() / kt.dur.sum())},
            "rp_task_advance_kernel<double> (fused task layer)": {"launches": int(len(task)), "avg_us": float(task.dur.mean() / 1e3) if len(task) else None, "share_of_gpu_time": float(task.dur.sum() / kt.dur.sum())},
            "torch kernels (action gather / scaling, ctrl scatter, masks, output copies)": {"launches_per_step": float(len(other) / nstep), "share_of_gpu_time": float(other.dur.sum() / kt.dur.sum())}},
        "note": "kernel durations overlap when the engine steps two slices on two streams: shares are of the summed kernel time, not of wall time"})
except Exception as e:  # keep going: the PMC summaries do not depend on the trace
    out["kernel_trace_error"] = repr(e)
pm = {}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = sorted(glob.glob(f"{R}/{d}/*/*counter_collection.csv"))
    if not f: continue
    df = pd.read_csv(f[-1]); df = df[df.Counter_Name == name]
    for tag, pat in (("solver", "<double, 1"), ("position", "<double, 0")):
        x = df[df.Kernel_Name.str.contains(pat, regex=False)]
        if tag == "position": x = x[x.Counter_Value > x.Counter_Value.max() * 0.05]
        pm[f"{name}_KB_per_launch_{tag}"] = float(x.Counter_Value.mean()); pm[f"n_{name}_{tag}"] = int(len(x))
if "FETCH_SIZE_KB_per_launch_solver" in pm and "WRITE_SIZE_KB_per_launch_solver" in pm:
    sol_b = pm["FETCH_SIZE_KB_per_launch_solver"] * 1024 * 2 + pm["WRITE_SIZE_KB_per_launch_solver"] * 1024
    pos_b = pm["FETCH_SIZE_KB_per_launch_position"] * 1024 * 2 + pm["WRITE_SIZE_KB_per_launch_position"] * 1024
    pm["note"] = ("separate --pmc passes (bench.py --stagger 0 --steps 4 --warmup 1, RP_STREAM_SLICES=1: one launch = one substep of all 4096 envs, "
                  "first control steps of the lockstep replay).  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 "
                  "counts 64 B per 128-B request, so the read side is doubled; WRITE_SIZE is used as reported (uncalibrated).  The counters "
                  "sit on the L2's fabric side: Infinity-Cache hits are included, i.e. an upper bound on HBM traffic.")
    pm["solver_kernel_bytes_per_launch_corrected"] = sol_b; pm["position_kernel_bytes_per_launch_corrected"] = pos_b
    out["pmc"] = pm
    json.dump({"envs": 4096, "envs_per_launch": 4096, "precision": 64, "solver_kernel_bytes_per_launch": sol_b, "position_kernel_bytes_per_launch": pos_b},
              open(os.path.join(OUT, "traffic_r02.json"), "w"))
elif pm:
    out["pmc_partial"] = pm
sq = {}
for d in ("sq1", "sq2"):
    f = sorted(glob.glob(f"{R}/{d}/*/*counter_collection.csv"))
    if not f: continue
    df = pd.read_csv(f[-1])
    for tag, pat in (("solver rp_stage_kernel<double, 1, 4, 9>", "<double, 1"), ("position rp_stage_kernel<double, 0, 0, 9>", "<double, 0")):
        x = df[df.Kernel_Name.str.contains(pat, regex=False)]
        if "<double, 0" in pat:
            big = x.groupby("Dispatch_Id").Counter_Value.sum(); x = x[x.Dispatch_Id.isin(big[big > big.max() * 0.05].index)]
        per = x.groupby("Counter_Name").Counter_Value.mean()
        waves = 4096.0
        sq.setdefault(tag, {"per_launch": {}, "per_wave": {}})
        for k, v in per.items():
            sq[tag]["per_launch"][k] = float(v); sq[tag]["per_wave"][k] = float(v) / waves
if sq:
    json.dump({"round": 2, "note": "rocprofv3 --pmc, separate passes, bench.py --stagger 0 --steps 4 --warmup 1 with one stream slice (config 2, lockstep, fp64, first five control steps); per wave = per launch / 4096 envs; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY count quad-cycles",
               "kernels": sq}, open(os.path.join(OUT, "r02_sq_instruction_mix.json"), "w"), indent=1)
json.dump(out, open(os.path.join(OUT, "r02_step_kernel_summary.json"), "w"), indent=1)
for f in glob.glob(R + "/bench_*.json") + glob.glob(R + "/bench_*.err") + glob.glob(R + "/*.log"):
    shutil.copy(f, OUT)
print(json.dumps(out, indent=1)[:4000])
