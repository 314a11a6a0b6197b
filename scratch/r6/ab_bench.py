"""Round 6: A/B of two builds of the library on a bench.py workload (alternating child processes on one box).
    gpurun -- python scratch/r6/ab_bench.py scratch/r6/ab/lib_head.so product adv_train [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    from robustart_amd import _lib
    if sys.argv[2].split(':')[0] != 'product':
        _lib.LIB_PATH = os.path.abspath(sys.argv[2].split(':')[0])
    sys.argv = ['bench.py'] + (['--workload', sys.argv[3]] if sys.argv[3] != 'headline' else []) + ['--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-secondary', '--no-fast-mode']
    import runpy
    runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
    sys.exit(0)
libs, wl = sys.argv[1:3], sys.argv[3]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', l, wl], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, RART_BENCH_NO_4X='1', **{kv.split('=', 1)[0]: kv.split('=', 1)[1] for kv in l.split(':')[1:]}))
        try:
            d = json.loads([ln for ln in o.stdout.splitlines() if ln.startswith('{')][-1])
            res[l].append(d['value'])
            print(l, round(d['value'], 1), round(d['ms_per_step'], 2), flush=True)
        except Exception:
            print(l, 'FAILED', o.stderr[-600:], flush=True)
for l in libs:
    print(l, 'values', [round(v, 1) for v in res[l]])
