"""Per-kernel table of the corruption sweep (profiles/corruption_sweep.py at B = 256, severity 3, rotating buffers):
time from the rocprofv3 --kernel-trace pass, HBM bytes from the two --pmc passes (FETCH_SIZE x2 per the gfx950 correction
of MI355X_MICROARCH.md section HBM, WRITE_SIZE; KiB units), and the per-corruption launch-sequence times measured with
events by the sweep itself (algorithmic bytes = 2 x 150 528 per image).

    python profiles/summarize_corruptions.py gpurun_out/corr_kt/sweep_results.db gpurun_out/corr_pmc_fetch/sweep_results.db \
        gpurun_out/corr_pmc_write/sweep_results.db gpurun_out/corruption_events.json profiles/r02_corruption_kernels.csv
"""
import csv
import json
import sqlite3
import sys


def kernel_times(db):
    c = sqlite3.connect(db)
    q = ("select s.kernel_name, count(*), avg(d.end - d.start), sum(d.end - d.start) from rocpd_kernel_dispatch d "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name")
    return {r[0]: (r[1], r[2], r[3]) for r in c.execute(q)}


def pmc(db):
    c = sqlite3.connect(db)
    q = ("select s.kernel_name, count(*), sum(p.value) from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name")
    return {r[0]: (r[1], r[2]) for r in c.execute(q)}


def short(name):
    n = name.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')
    i = 0
    while i < len(n) and n[i].isdigit():
        i += 1
    return n[i:][:60] if i else name[:60]


def main(kt, fdb, wdb, events, out):
    t, f, w = kernel_times(kt), pmc(fdb), pmc(wdb)
    rows = []
    for k, (calls, avg_ns, tot_ns) in sorted(t.items(), key=lambda kv: -kv[1][2]):
        if 'at6native' in k or 'rocclr' in k:
            continue
        fb = f[k][1] * 1024.0 * 2.0 / f[k][0] if k in f else None
        wb = w[k][1] * 1024.0 / w[k][0] if k in w else None
        hb = (fb or 0) + (wb or 0) if (fb is not None or wb is not None) else None
        rows.append([short(k), calls, round(avg_ns / 1e3, 2), round(fb) if fb is not None else '', round(wb) if wb is not None else '',
                     round(hb) if hb is not None else '', round(hb / avg_ns, 1) if hb else ''])
    ev = json.load(open(events))
    with open(out, 'w', newline='') as fh:
        cw = csv.writer(fh)
        cw.writerow(['# per kernel: rocprofv3 kernel trace + PMC passes of profiles/corruption_sweep.py (B=256, severity 3)'])
        cw.writerow(['kernel', 'launches', 'avg_us', 'pmc_fetch_bytes_x2', 'pmc_write_bytes', 'pmc_hbm_bytes', 'pmc_GB_per_s'])
        cw.writerows(rows)
        cw.writerow([])
        cw.writerow(['# per corruption: whole launch sequence of one 256-image batch (events), algorithmic bytes = 2 x 38 535 168'])
        cw.writerow(['corruption', 'severity', 'us_per_batch', 'algorithmic_GB_per_s', 'fraction_of_8TB_per_s'])
        for r in ev:
            cw.writerow([r['corruption'], r['severity'], round(r['us_per_batch'], 1), round(r['algorithmic_GBps'], 1),
                         round(r['frac_hbm_8TBps'], 4)])
    print(open(out).read())


if __name__ == '__main__':
    main(*sys.argv[1:6])
