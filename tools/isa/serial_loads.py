#!/usr/bin/env python3
"""Memory instructions whose result is waited for almost immediately (a dependent round trip each), per PROF section of an
ISA listing built with -DRPK_MARK.  usage: serial_loads.py listing.s [max-distance]"""
import re, sys, collections
lines = [l.strip() for l in open(sys.argv[1]).read().splitlines()]
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sec = "start"; out = collections.OrderedDict()
ins = []
for t in lines:
    m = re.match(r";\s*MARK (\d+)", t)
    if m: ins.append(("MARK", m.group(1))); continue
    if not t or t[0] in ";." or t.endswith(":"):
        if t.endswith(":") and t.startswith(".LBB"): ins.append(("LABEL", t))
        continue
    ins.append(("I", t))
cur = "start"; stats = collections.OrderedDict()
def st(k): return stats.setdefault(k, dict(rounds=0, lds_rounds=0, vm_rounds=0, waits=0))
pending = 0; last_load = None
for i, (k, t) in enumerate(ins):
    if k == "MARK": cur = "after " + t; continue
    if k != "I": continue
    op = t.split()[0]
    if op == "s_waitcnt":
        st(cur)["waits"] += 1
        # count a round trip when this wait drains to 0 loads of a kind that were issued since the previous drain
        if "lgkmcnt(0)" in t: st(cur)["lds_rounds"] += 1
        if "vmcnt(0)" in t: st(cur)["vm_rounds"] += 1
for k, v in stats.items():
    print("%-10s waits %4d  full lgkm drains %3d  full vm drains %3d" % (k, v["waits"], v["lds_rounds"], v["vm_rounds"]))
