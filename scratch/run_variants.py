"""Times bench.py (config 2, fp64, no CPU leg) for every librp_engine build under
robopianist_amd/csrc/variants/ (scratch/build_variants.sh).  GPU box only."""
import glob, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
libs = sorted(glob.glob(os.path.join(root, "robopianist_amd/csrc/variants/librp_engine_*.so")))
only = sys.argv[1:]
for lib in libs:
    name = os.path.basename(lib)[len("librp_engine_"):-3]
    if only and name not in only:
        continue
    env = dict(os.environ, RP_ENGINE_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "158", "--warmup", "5", "--aux-fp32", "0",
                        "--host-io", "0", "--no-cpu-baseline", "--stagger", "0"], capture_output=True, text=True, env=env)
    try:
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        out[name] = dict(value=j["value"], solver_ms=j["roofline"]["kernel_avg_ms"], seq_ms=j["roofline"]["step_sequence_avg_ms"],
                         warn=j["sanity"]["warn_flags_or"])
    except Exception as e:
        out[name] = dict(error=r.stderr[-500:])
    print(name, out[name], flush=True)
json.dump(out, open(os.path.join(root, "gpurun_out", "variants.json"), "w"), indent=1)
