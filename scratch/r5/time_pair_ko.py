"""Round 5: where the time of k_gemm_pair goes, per launch shape of a reference-precision ResNet-50 gradient evaluation at B = 256: the product
library beside the knock-out builds of scratch/r5/build_ko.sh (NOLOAD / NOMFMA / NOEPI / NORES / NOSTORE).
    gpurun -- python scratch/r5/time_pair_ko.py   ->  gpurun_out/r05_pair_knockouts.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [(50176, 256, 1024, 1), (50176, 1024, 256, 1), (50176, 2304, 256, 9), (802816, 256, 64, 1), (802816, 64, 256, 1),
          (12544, 4608, 512, 9), (200704, 128, 512, 1), (200704, 512, 128, 1), (12544, 512, 2048, 1), (12544, 2048, 512, 1)]
if len(sys.argv) > 1:          # child: time one library
    from robustart_amd import _lib
    if sys.argv[1] != 'product':
        _lib.LIB_PATH = sys.argv[1]
    import torch
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import ResNet50Engine
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    torch.manual_seed(0)
    eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'fp32x')
    eng.fused_tail_pair = False          # every convolution on k_gemm_pair: the shapes a fused kernel would replace are measured too
    x = torch.rand(256, 3, 224, 224, device='cuda')
    y = torch.randint(0, 1000, (256,), device='cuda')
    calls = {}
    orig = eng._gemm_pair

    def rec(*a):
        src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols = a[:10]
        key = (batch * grid[0] * grid[1], k_per_tap * len(taps), n_cols, len(taps))
        calls.setdefault(key, [0, a, a[13] is not None])[0] += 1       # a[13] = res
        return orig(*a)
    eng._gemm_pair = rec
    eng.forward_backward(x, MEAN, STD, y, 0)
    torch.cuda.synchronize()
    eng._gemm_pair = orig
    out = {}
    for key in SHAPES:
        if key not in calls:
            continue
        cnt, a, has_res = calls[key]
        for _ in range(2):
            orig(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            orig(*a)
        e1.record()
        torch.cuda.synchronize()
        out['%d_%d_%d_%d' % key] = dict(us=round(e0.elapsed_time(e1) / 6 * 1e3, 1), count=cnt, residual=has_res)
    print(json.dumps(out))
    sys.exit(0)
res = {}
ko = os.path.join(ROOT, 'scratch', 'r5', 'ko')
libs = ['product'] + (sorted(os.path.join(ko, f) for f in os.listdir(ko) if f.endswith('.so')) if os.path.isdir(ko) else [])
for lib in libs:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), lib], capture_output=True, text=True, timeout=600)
    name = os.path.basename(lib).replace('lib_', '').replace('.so', '')
    try:
        res[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    except Exception:
        res[name] = {'error': r.stderr[-500:]}
    print(name, res[name], flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'r05_pair_knockouts.json'), 'w'), indent=1)
names = [n for n in res if 'error' not in res[n]]
print('%-22s' % 'M_K_N_taps' + ''.join('%10s' % n for n in names))
for key in SHAPES:
    k = '%d_%d_%d_%d' % key
    if all(k in res[n] for n in names):
        print('%-22s' % k + ''.join('%10.1f' % res[n][k]['us'] for n in names))
