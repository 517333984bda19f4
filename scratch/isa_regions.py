"""Static instruction mix between '; MARK n' comments of one kernel in an ISA listing."""
import re, sys, collections
f, kern = sys.argv[1], sys.argv[2]
lines = open(f).read().split('\n')
start = next(i for i,l in enumerate(lines) if l.startswith(kern) and ":" in l and not l.startswith("\t"))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i] and i > start + 1000)
cur = 'start'; stats = collections.OrderedDict()
def cls(op):
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith(('v_readlane','v_readfirstlane','v_writelane')): return 'lane'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_mov'): return 'vmov'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith(('s_cbranch','s_branch')): return 'br'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_','flat_','buffer_')): return 'vmem'
    if op.startswith('scratch_'): return 'scr'
    return 'other'
for l in lines[start:end]:
    m = re.match(r'\s*; MARK (\d+)', l)
    if m:
        cur = 'after %s' % m.group(1); continue
    t = l.strip()
    if not t or t.startswith((';','.')) or t.endswith(':'): continue
    op = t.split()[0]
    stats.setdefault(cur, collections.Counter())[cls(op)] += 1
keys = ['valu','vmov','cnd','acc','lane','salu','wait','br','nop','lds','vmem','scr','other']
print('%-10s' % 'region', ' '.join('%6s' % k for k in keys), ' total')
for r, c in stats.items():
    print('%-10s' % r, ' '.join('%6d' % c[k] for k in keys), '%6d' % sum(c.values()))
