"""Round 6: from a rocprofv3 --kernel-trace database: for the LAST gradient evaluation in the trace (the dispatches after the last k_stem_fwd_pair /
k_patchify), the span, the sum of kernel durations, and the idle time between consecutive dispatches (by size of gap).
    python scratch/r6/trace_gaps.py <db> [first-kernel-substring]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else 'stem_fwd_pair'
rows = c.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if key in r[0]]
a = idx[-1]
ev = rows[a:]
span = ev[-1][2] - ev[0][1]
busy = sum(r[2] - r[1] for r in ev)
gaps = [ev[i + 1][1] - ev[i][2] for i in range(len(ev) - 1)]
print('%d dispatches, span %.3f ms, kernel time %.3f ms, idle between dispatches %.3f ms (%.1f %%)' % (len(ev), span / 1e6, busy / 1e6, sum(g for g in gaps if g > 0) / 1e6, 100.0 * sum(g for g in gaps if g > 0) / span))
import collections
h = collections.Counter()
for g in gaps:
    h[min(int(max(g, 0) / 1000), 20)] += 1
print('gap histogram (us: count):', dict(sorted(h.items())))
big = sorted(((g, ev[i][0][:60], ev[i + 1][0][:60]) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, a_, b_ in big: print('  %.1f us between %s -> %s' % (g / 1e3, a_, b_))
