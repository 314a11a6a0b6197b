"""Per-kernel averages of arbitrary PMC counters / derived metrics from one or more rocprofv3 --pmc passes (rocpd sqlite output).

    python profiles/summarize_pmc_generic.py out.csv pass1_results.db pass2_results.db ...
Each row: kernel, calls, avg_ns, then one column per counter (value summed over the dispatch's dimensions, averaged over launches)."""
import csv
import sqlite3
import sys


def tables(c):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table'")]


def main(out, dbs):
    rows, counters = {}, []
    for db in dbs:
        c = sqlite3.connect(db)
        t = tables(c)
        pmc_info = [x for x in t if x.startswith('rocpd_info_pmc')][0]
        ev = [x for x in t if x.startswith('rocpd_pmc_event')][0]
        disp = [x for x in t if x.startswith('rocpd_kernel_dispatch')][0]
        sym = [x for x in t if x.startswith('rocpd_info_kernel_symbol')][0]
        q = (f"select s.kernel_name, i.name, count(distinct d.id), sum(p.value), avg(d.end - d.start) from {ev} p "
             f"join {disp} d on p.event_id = d.event_id join {sym} s on d.kernel_id = s.id join {pmc_info} i on p.pmc_id = i.id "
             f"group by s.kernel_name, i.name")
        for k, name, calls, total, avg_ns in c.execute(q):
            r = rows.setdefault(k, {'calls': calls, 'avg_ns': avg_ns})
            r[name] = total / calls
            if name not in counters:
                counters.append(name)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'avg_ns'] + counters)
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1]['avg_ns'] * kv[1]['calls']):
            w.writerow([k, r['calls'], '%.0f' % r['avg_ns']] + ['%.4g' % r[c] if c in r else '' for c in counters])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
