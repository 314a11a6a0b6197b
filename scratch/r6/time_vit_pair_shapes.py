"""Round 6: every rart_gemm_pair_bf16 launch shape of a reference-precision ViT-B/16 gradient evaluation at B = 256, replayed alone:
us per launch, issued TFLOP/s (3 products), fraction of the dense bf16 peak, and bytes moved per launch (A + W + C + residual).
    gpurun -- python scratch/r6/time_vit_pair_shapes.py  ->  gpurun_out/r06_vit_pair_shapes.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robustart_amd import _lib
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
lib = _lib.load()
torch.manual_seed(0)
B = int(os.environ.get('B', 256))
eng = ViTEngine(get_model({'type': 'vit_base_patch16_224'}).eval(), 'cuda', 'fp32x')
x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
calls = {}
orig = eng._gemm_pair
def rec(*a, **kw):
    M, N, K = a[3], a[4], a[5]
    key = (M, N, K, kw.get('flags', 0), kw.get('res') is not None, kw.get('aux') is not None)
    calls.setdefault(key, [0, a, kw])[0] += 1
    return orig(*a, **kw)
eng._gemm_pair = rec
eng.forward_backward(x, MEAN, STD, y, 0); torch.cuda.synchronize()
eng._gemm_pair = orig
def t_us(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lines = []
tot = {}
for key, (cnt, a, kw) in sorted(calls.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1][0]):
    M, N, K, flags, res, aux = key
    row = '%6d x %4d -> %4d flags %3d res %d aux %d x%3d:' % (M, K, N, flags, res, aux, cnt)
    for s in (1, 11, 21):          # 1 = ping-pong alone; 11 = + remainder split (RART_PAIR_SPLIT); 21 = + 224-row tiles where they save a pass
        os.environ['RART_PAIR_SPLIT'] = '1' if s == 11 else '0'
        os.environ['RART_PAIR_ROWS224'] = '1' if s == 21 else '0'
        lib.rart_gemm_pair_set_schedule(s % 10)
        us = min(t_us(lambda: orig(*a, **kw)) for _ in range(3))
        tf = 6.0 * M * N * K / us / 1e6
        row += '  sched %2d %7.1f us (%.2f)' % (s, us, tf / 2500)
        tot[s] = tot.get(s, 0) + us * cnt
    by = (M * K + N * K + M * N * (1 + res + aux)) * 4
    row += '  %6.0f MB -> %4.0f us at 5 TB/s' % (by / 1e6, by / 5e6)
    lines.append(row); print(row, flush=True)
lines.append('sum over a gradient evaluation: ping-pong %.1f ms, + remainder split %.1f ms, + 224-row tiles %.1f ms' % (tot[1] / 1e3, tot[11] / 1e3, tot[21] / 1e3)); print(lines[-1])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
open(os.path.join(ROOT, 'gpurun_out', 'r06_vit_pair_shapes.txt'), 'w').write('\n'.join(lines) + '\n')
