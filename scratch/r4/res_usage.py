"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: python scratch/r4/res_usage.py build.log"""
import re, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
KS = {"v": r"VGPRs", "a": r"AGPRs", "s": r"ScratchSize \[bytes/lane\]", "l": r"LDS Size \[bytes/block\]", "o": r"Occupancy \[waves/SIMD\]", "sp": r"VGPRs Spill", "ss": r"SGPRs Spill"}
ALL = []
for b in blocks:
    name = b.split('\n')[0].strip().split(' ')[0]
    if 'rp_' not in name: continue
    r = {}
    for k, pat in KS.items():
        m = re.search(pat + r': (\d+)', b); r[k] = int(m.group(1)) if m else -1
    import subprocess
    try: dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception: dn = name
    dn = re.sub(r'\(.*', '', dn)
    if len(sys.argv) > 2:
        ALL.append((dn, r))
    print(f"{dn[:90]:90s} VGPR {r['v']:4d} AGPR {r['a']:4d} spill {r['sp']:4d} scratch {r['s']:5d} LDS {r['l']:6d} occ {r['o']}")

if len(sys.argv) > 2:   # json dump of the fp64 kernels: python res_usage.py build.log out.json
    import json
    doc = {"source": "hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (the shipped build of robopianist_amd/csrc)", "kernels": {}}
    for dn, r in ALL:
        if "<double" in dn or "double>" in dn:
            doc["kernels"][dn] = {"vgprs": r["v"], "agprs": r["a"], "vgprs_spilled": r["sp"], "sgprs_spilled_to_vgpr_lanes": r["ss"],
                                  "scratch_bytes_per_lane": r["s"], "lds_bytes": r["l"], "waves_per_simd": r["o"]}
    json.dump(doc, open(sys.argv[2], "w"), indent=1)
