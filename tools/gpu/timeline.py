"""Timeline of a rocprofv3 --kernel-trace csv: GPU busy share, overlap histogram, per-kernel totals (last 60 % of the run)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
t0, t1 = rows[0][1], rows[-1][2]
lo = t0 + 0.4 * (t1 - t0)
rows = [r for r in rows if r[1] >= lo]
span = rows[-1][2] - rows[0][1]
ev = []
for n, s, e in rows: ev += [(s, 1), (e, -1)]
ev.sort()
hist = collections.Counter(); cur = 0; last = ev[0][0]
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print('window %.1f ms, kernels %d' % (span / 1e6, len(rows)))
for k in sorted(hist): print('  %d kernels active: %5.1f %%' % (k, 100 * hist[k] / span))
tot = collections.Counter(); cnt = collections.Counter()
for n, s, e in rows: tot[n[:70]] += e - s; cnt[n[:70]] += 1
for n, v in tot.most_common(10): print('  %-70s sum %6.1f %% of window, avg %.1f us, n %d' % (n, 100 * v / span, v / cnt[n] / 1e3, cnt[n]))
# a segment of the raw timeline (mid-window)
r = csv.DictReader(open(f)); cols = r.fieldnames
qcol = next((c for c in cols if c.lower() in ('queue_id', 'stream_id')), None)
raw = sorted(((int(x['Start_Timestamp']), int(x['End_Timestamp']), x['Kernel_Name'], x.get('Queue_Id', ''), x.get('Stream_Id', '')) for x in csv.DictReader(open(f))))
mid = len(raw) // 2
base = raw[mid][0]
short = lambda n: ('LEAN' if 'lean' in n else 'HEAVY' if 'rp_stage_kernel<double, 1' in n else 'POS' if 'rp_stage_kernel<double, 0' in n else 'SENS' if 'rp_stage_kernel<double, 2' in n else 'order' if 'order' in n else n[:24])
for s, e, n, q, st in raw[mid:mid + int(sys.argv[2]) if len(sys.argv) > 2 else mid]:
    print('%9.1f us  +%7.1f  q%s s%s  %s' % ((s - base) / 1e3, (e - s) / 1e3, q, st, short(n)))
