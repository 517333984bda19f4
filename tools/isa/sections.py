#!/usr/bin/env python3
"""Static instruction mix of a kernel's ISA listing between the `; MARK n` comments PROF() leaves in -DRPK_MARK builds.
usage: sections.py listing.s [kernel-substring]"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
order = []; mix = collections.defaultdict(collections.Counter); cur = collections.Counter()
def kind(op):
    if op.startswith("v_"):
        if re.match(r"v_(fma|add|mul|fmac|min|max|rcp|rsq|sqrt|div|ldexp|frexp|trig|cvt)_?.*f64", op) or op.endswith("_f64"): return "valu_f64"
        if "cndmask" in op: return "valu_sel"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "valu_mov"
        if "readlane" in op or "readfirstlane" in op or "writelane" in op or "permlane" in op: return "valu_lane"
        if op.startswith("v_cmp"): return "valu_cmp"
        return "valu_int"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"
for ln in lines:
    t = ln.strip()
    m = re.match(r";\s*MARK (\d+)", t)
    if m:   # PROF(n) closes phase n: the code since the previous marker belongs to it
        sec = "%s#%d" % (m.group(1), len(order)); order.append(sec); mix[sec] = cur; cur = collections.Counter(); continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    op = t.split()[0]
    if not re.match(r"^[a-z_0-9]+$", op): continue
    cur[kind(op)] += 1
    if "dpp" in t: cur["(dpp)"] += 1
cols = ["valu_f64", "valu_sel", "valu_mov", "valu_lane", "valu_cmp", "valu_int", "lds", "vmem", "smem", "salu", "branch", "wait", "nop", "(dpp)"]
print("%-12s %6s " % ("section", "total") + " ".join("%8s" % c for c in cols))
order.append("tail"); mix["tail"] = cur
for s in order:
    c = mix[s]; tot = sum(v for k, v in c.items() if k != "(dpp)")
    print("%-12s %6d " % (s, tot) + " ".join("%8d" % c[k] for k in cols))
