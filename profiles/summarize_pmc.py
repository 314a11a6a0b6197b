"""FETCH_SIZE / WRITE_SIZE (two separate rocprofv3 --pmc passes of the same bench.py command) -> per-kernel HBM
traffic per launch.  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md section HBM: the counters
are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of a wide coalesced read stream (calibrated below on
k_normal_noise_mfma, whose read bytes are known: 256 x 150528), WRITE_SIZE is exact.

    python profiles/summarize_pmc.py gpurun_out/pmc_fetch/b_results.db gpurun_out/pmc_write/b_results.db \
        profiles/r01_pmc_traffic.json
"""
import json
import sqlite3
import sys


def per_kernel(db):
    c = sqlite3.connect(db)
    q = ("select s.kernel_name, count(*), sum(p.value), avg(d.end - d.start) from rocpd_pmc_event p "
         "join rocpd_kernel_dispatch d on p.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name")
    return {r[0]: (r[1], r[2], r[3]) for r in c.execute(q)}


def main(fetch_db, write_db, out):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    res = {'units': 'bytes per launch', 'fetch_correction': 2.0, 'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE '
           '-- python bench.py --steps 1 --warmup 1 --no-cpu-baseline (two passes)', 'kernels': {}}
    for k in sorted(f, key=lambda k: -f[k][1]):
        if k not in w:
            continue
        calls = f[k][0]
        fetch = f[k][1] * 1024.0 * 2.0 / calls
        write = w[k][1] * 1024.0 / w[k][0]
        res['kernels'][k] = {'calls': calls, 'fetch_bytes': fetch, 'write_bytes': write, 'hbm_bytes': fetch + write,
                             'avg_ns_under_pmc': f[k][2]}
    ig = [v for k, v in res['kernels'].items() if 'k_conv_igemm_bf16' in k]
    n = sum(v['calls'] for v in ig)
    res['k_conv_igemm_bf16_all'] = {'calls': n, 'hbm_bytes': sum(v['hbm_bytes'] * v['calls'] for v in ig) / n,
                                    'fetch_bytes': sum(v['fetch_bytes'] * v['calls'] for v in ig) / n,
                                    'write_bytes': sum(v['write_bytes'] * v['calls'] for v in ig) / n}
    gn = [v for k, v in res['kernels'].items() if 'k_normal_noise_mfmaI' in k]      # the single-severity kernel (not ..._multi)
    if gn:
        res['calibration'] = {'kernel': 'k_normal_noise_mfma<0>', 'known_read_bytes': 256 * 150528,
                              'fetch_bytes_after_x2': gn[0]['fetch_bytes'], 'known_write_bytes': 256 * 150528,
                              'write_bytes': gn[0]['write_bytes']}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: res[k] for k in ('k_conv_igemm_bf16_all', 'calibration') if k in res}, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:4])
