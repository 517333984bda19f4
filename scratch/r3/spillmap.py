"""Where a kernel's scratch traffic comes from: scratch_load / scratch_store instructions per source line
(hipcc -S --cuda-device-only -gline-tables-only)."""
import re, collections, sys
cnt = collections.Counter(); cur = None; n_inst = 0
files = {}
for l in open(sys.argv[1]):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if re.match(r'\s+scratch_', l): cnt[cur] += 1
    if re.match(r'\s+(v_|s_|ds_|global_|scratch_|buffer_|flat_)', l): n_inst += 1
print('instructions', n_inst, 'scratch ops', sum(cnt.values()))
for (f, ln), c in sorted(cnt.items()):
    if c >= int(sys.argv[2]) if len(sys.argv) > 2 else 3: print(files.get(f, f), ln, c)
