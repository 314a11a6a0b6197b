"""Per-dispatch counter values of a rocprofv3 --pmc pass (rocpd sqlite): kernel, grid, duration, counters.
    python scratch/r6/pmc_per_dispatch.py <db> [kernel-substring]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ''
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
pmc_info = [x for x in t if x.startswith('rocpd_info_pmc')][0]; ev = [x for x in t if x.startswith('rocpd_pmc_event')][0]
disp = [x for x in t if x.startswith('rocpd_kernel_dispatch')][0]; sym = [x for x in t if x.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute(f"pragma table_info({disp})")]
gx = 'grid_size_x' if 'grid_size_x' in cols else ('grid_x' if 'grid_x' in cols else None)
wx = 'workgroup_size_x' if 'workgroup_size_x' in cols else None
q = (f"select d.id, s.kernel_name, {('d.' + gx) if gx else '0'}, {('d.' + wx) if wx else '1'}, d.end - d.start, i.name, sum(p.value) from {ev} p join {disp} d on p.event_id = d.event_id "
     f"join {sym} s on d.kernel_id = s.id join {pmc_info} i on p.pmc_id = i.id group by d.id, i.name order by d.start")
rows = {}
for did, k, g, w, ns, name, v in c.execute(q):
    if sub not in k: continue
    r = rows.setdefault(did, {'k': k[:48], 'wgs': (g // w) if w else g, 'us': ns / 1e3})
    r[name] = v
for did, r in rows.items():
    print('%-48s wgs %5d %8.1f us  ' % (r['k'], r['wgs'], r['us']) + '  '.join('%s %.4g' % (n, v) for n, v in r.items() if n not in ('k', 'wgs', 'us')))
