"""Any set of rocprofv3 --pmc counters (one pass, with --kernel-trace) -> per-kernel sums per launch.

    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/p -o c -- <cmd>
    python profiles/summarize_counters.py /tmp/p/.../c_results.db out.json

Derived: lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles / all LDS-array cycles, MI355X_MICROARCH.md LDS section);
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs) -- the counter sums the busy cycles of every matrix pipe."""
import json
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    try:
        # a dispatch has one row per counter INSTANCE (XCD / SE): sum them, count the dispatches themselves
        q = ("select s.kernel_name, i.name, count(distinct d.id), sum(p.value), avg(d.end - d.start) from rocpd_pmc_event p "
             "join rocpd_info_pmc i on p.pmc_id = i.id join rocpd_kernel_dispatch d on p.event_id = d.event_id "
             "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, i.name")
        rows = list(c.execute(q))
    except sqlite3.Error as e:          # another rocpd schema: show it
        print('query failed:', e)
        for name, sql in c.execute("select name, sql from sqlite_master where name like 'rocpd_%pmc%' or name like 'rocpd_info_pmc%'"):
            print(name, sql)
        raise
    res = {}
    for k, ctr, n, tot, avg_ns in rows:
        e = res.setdefault(k, {'calls': n, 'avg_ns_under_pmc': avg_ns})
        e[ctr] = tot / n
    for k, e in res.items():
        if e.get('SQ_LDS_IDX_ACTIVE'):
            e['lds_conflict_ratio'] = e.get('SQ_LDS_BANK_CONFLICT', 0.0) / e['SQ_LDS_IDX_ACTIVE']
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and e['avg_ns_under_pmc']:
            e['mfma_busy_frac'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['avg_ns_under_pmc'] * 2.4 * 1024)
    order = sorted(res, key=lambda k: -res[k]['calls'] * res[k]['avg_ns_under_pmc'])
    json.dump({k: res[k] for k in order}, open(out, 'w'), indent=1)
    for k in order[:14]:
        e = res[k]
        print('%-70s x%4d %8.1f us  lds_conflict %.3f  mfma_busy %.3f' % (k[:70], e['calls'], e['avg_ns_under_pmc'] / 1e3,
                                                                         e.get('lds_conflict_ratio', float('nan')), e.get('mfma_busy_frac', float('nan'))))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
