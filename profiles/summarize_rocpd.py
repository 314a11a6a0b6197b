"""Turn a rocprofv3 rocpd sqlite database (rocprofv3 --kernel-trace --stats -d DIR -o NAME) into the
per-kernel summary committed under profiles/.

    python profiles/summarize_rocpd.py gpurun_out/prof_r1/bench_results.db profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'percent', 'arch_vgpr', 'accum_vgpr',
                    'sgpr', 'lds_bytes'])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / total, 3), r[6], r[7], r[8], r[9]])
    for r in rows[:14]:
        print('%6.2f%% %7d calls avg %9.1f us  %s' % (100.0 * r[2] / total, r[1], r[3] / 1e3, r[0][:90]))
    print('total kernel time %.3f ms over %d kernels' % (total / 1e6, len(rows)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
