#!/usr/bin/env python3
"""What sits between consecutive launches of the scoring kernel on the GPU's timeline?  Reads a rocprofv3 --kernel-trace csv
(*_kernel_trace.csv under the given directory) and prints, for the gaps between the end of one gmm_score_split16_kernel dispatch and
the start of the next: their size, and which kernels (name, queue) ran inside them."""
import collections, csv, glob, sys
files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
score = [r for r in rows if 'gmm_score_split16' in r[2]]
print('%d dispatches, %d scoring launches; queues of the scoring kernel: %s' % (len(rows), len(score), sorted({r[3] for r in score})))
gaps = []
for a, b in zip(score[:-1], score[1:]):
    g = (b[0] - a[1]) / 1e3
    if g < 5000:                                     # (us; the long ones are the pauses between the probe's loops)
        inside = [(r[2][:60], r[3], (r[0] - a[1]) / 1e3, (r[1] - r[0]) / 1e3) for r in rows if r[0] >= a[1] - 1 and r[0] < b[0] and 'gmm_score_split16' not in r[2]]
        gaps.append((g, inside))
gs = sorted(g for g, _ in gaps)
if gs:
    print('gaps between scoring launches (us): n=%d median %.1f mean %.1f p90 %.1f max %.1f' % (len(gs), gs[len(gs) // 2], sum(gs) / len(gs), gs[len(gs) * 9 // 10], gs[-1]))
hist = collections.Counter(int(g // 50) * 50 for g in gs)
print('histogram (us bucket: count):', sorted(hist.items()))
for lo in sorted(hist):
    ex = [x for x in gaps if lo <= x[0] < lo + 50][:2]
    for g, inside in ex:
        print('  gap %.1f us:' % g, [(n, q, round(t0, 1), round(d, 1)) for n, q, t0, d in inside][:8])
# per queue: which kernels run where
byq = collections.defaultdict(collections.Counter)
for r in rows:
    byq[r[3]][r[2][:50]] += 1
for q in sorted(byq):
    print('queue', q, dict(byq[q].most_common(6)))
