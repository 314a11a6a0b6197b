"""Round 6: lab builds of the ping-pong pair GEMM (scratch/r6/build_pp_variants.sh) on the K-deep launch shapes of the reference-precision
ResNet-50 at B = 256: time per launch for the product, round 4's loop (schedule 0) and every knock-out; phase stamps of the STAMPS build.
    gpurun -- python scratch/r6/pp_lab.py  ->  gpurun_out/r06_pp_lab.json"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [(50176, 2304, 256, 9), (50176, 1024, 256, 1), (12544, 4608, 512, 9), (12544, 2048, 512, 1), (50176, 256, 1024, 1)]
NAMES = ['M0 issue+reads', 'M0 vmcnt', 'M0 barrier', 'C0 mfma issue', 'C0 barrier', 'M1 issue+reads', 'M1 vmcnt', 'M1 barrier',
         'C1 mfma issue', 'C1 vmcnt', 'C1 barrier', 'K step']
if len(sys.argv) > 1:
    from robustart_amd import _lib
    libname, sched = sys.argv[1], int(sys.argv[2])
    if libname != 'product':
        _lib.LIB_PATH = libname
    import torch
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import ResNet50Engine
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    torch.manual_seed(0)
    eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'fp32x')
    eng.fused_tail_pair = False
    lib = _lib.load()
    lib.rart_gemm_pair_set_schedule(sched)
    x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
    calls = {}
    orig = eng._gemm_pair

    def rec(*a):
        src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols = a[:10]
        calls.setdefault((batch * grid[0] * grid[1], k_per_tap * len(taps), n_cols, len(taps)), a)
        return orig(*a)
    eng._gemm_pair = rec
    eng.forward_backward(x, MEAN, STD, y, 0)
    torch.cuda.synchronize()
    eng._gemm_pair = orig
    stamps_lib = 'STAMPS' in libname
    if stamps_lib:
        raw = ctypes.CDLL(_lib.LIB_PATH)
        buf = (ctypes.c_ulonglong * 32)()
    out = {}
    for key in SHAPES:
        a = calls[key]
        stamps = stamps_lib and key[1] >= 1024
        lib.rart_gemm_pair_set_schedule(0)
        orig(*a); torch.cuda.synchronize()
        want = a[2].clone()
        lib.rart_gemm_pair_set_schedule(sched)
        a[2].zero_()
        for _ in range(2):
            orig(*a)
        torch.cuda.synchronize()
        same = bool(torch.equal(a[2].view(torch.int16), want.view(torch.int16)))
        if stamps:
            raw.rart_debug_pp_stamps(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            orig(*a)
        e1.record(); torch.cuda.synchronize()
        r = {'us': round(e0.elapsed_time(e1) / 6 * 1e3, 1), 'same': same}
        if stamps:
            raw.rart_debug_pp_stamps(buf)
            for g in range(2):
                n = max(1, buf[g * 16 + 12])
                r['g%d' % g] = {NAMES[i]: round(buf[g * 16 + i] / n) for i in range(12)}
                wg = max(1, buf[g * 16 + 15])
                r['g%d' % g].update({'prologue': round(buf[g * 16 + 13] / wg), 'epilogue': round(buf[g * 16 + 14] / wg),
                                    'loop': round(buf[g * 16 + 11] / wg), 'launches_x_wgs': wg})
        out['%d_%d_%d_%d' % key] = r
    print(json.dumps(out))
    sys.exit(0)
res = {}
pp = os.path.join(ROOT, 'scratch', 'r6', 'pp')
runs = [('two_stage', 'product', 0, {}), ('pingpong', 'product', 1, {})]
for f in (sorted(os.listdir(pp)) if os.path.isdir(pp) else []):
    if f == 'lib_LAB.so':
        runs += [('opt%d' % o, os.path.join(pp, f), 1, {'RART_PP_OPT': str(o)}) for o in [int(v) for v in os.environ.get('PP_OPTS', '0,8,9').split(',')]]
    elif f.endswith('.so'):
        runs.append((f[4:-3], os.path.join(pp, f), 1, {}))
only = os.environ.get('PP_ONLY')
if only:
    runs = [r for r in runs if any(r[0].startswith(o) for o in only.split(','))]
for name, lib, sched, env in runs:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), lib, str(sched)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env))
    try:
        res[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    except Exception:
        res[name] = {'error': r.stderr[-800:]}
    print(name, json.dumps(res[name]), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'r06_pp_lab.json'), 'w'), indent=1)
names = [n for n in res if 'error' not in res[n]]
print('%-22s' % 'M_K_N_taps' + ''.join('%10s' % n for n in names))
for key in SHAPES:
    k = '%d_%d_%d_%d' % key
    print('%-22s' % k + ''.join('%9.1f%s' % (res[n][k]['us'], ' ' if res[n][k]['same'] else '!') for n in names))
for sn in [n for n in names if n.startswith('STAMPS')]:
    for key in SHAPES[:2]:
        k = '%d_%d_%d_%d' % key
        print(sn, k)
        for nm in NAMES + ['prologue', 'loop', 'epilogue', 'launches_x_wgs']:
            print('   %-16s g0 %6d   g1 %6d' % (nm, res[sn][k]['g0'][nm], res[sn][k]['g1'][nm]))
