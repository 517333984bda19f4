"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: python scratch/r4/res_usage.py build.log"""
import re, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
KS = {"v": r"VGPRs", "a": r"AGPRs", "s": r"ScratchSize \[bytes/lane\]", "l": r"LDS Size \[bytes/block\]", "o": r"Occupancy \[waves/SIMD\]", "sp": r"VGPR Spill"}
for b in blocks:
    name = b.split('\n')[0].strip()
    if 'rp_' not in name: continue
    r = {}
    for k, pat in KS.items():
        m = re.search(pat + r': (\d+)', b); r[k] = int(m.group(1)) if m else -1
    import subprocess
    try: dn = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
    except Exception: dn = name
    dn = re.sub(r'\(.*', '', dn)
    print(f"{dn[:90]:90s} VGPR {r['v']:4d} AGPR {r['a']:4d} spill {r['sp']:4d} scratch {r['s']:5d} LDS {r['l']:6d} occ {r['o']}")
