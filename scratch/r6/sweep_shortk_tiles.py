"""Round 6: the short-K 1x1 launches of a reference-precision ResNet-50 gradient evaluation (K <= 512, B = 256) under every tile of
rart_gemm_pair_bf16: us per launch.   gpurun -- python scratch/r6/sweep_shortk_tiles.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robustart_amd import _lib
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
lib = _lib.load()
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'fp32x')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
calls = {}
orig = eng._gemm_pair
def rec(*a):
    src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols = a[:10]
    key = (batch * grid[0] * grid[1], k_per_tap * len(taps), n_cols, len(taps), a[13] is not None)
    calls.setdefault(key, [0, a])[0] += 1
    return orig(*a)
eng._gemm_pair = rec
eng.forward_backward(x, MEAN, STD, y, 0); torch.cuda.synchronize()
eng._gemm_pair = orig
def t_us(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tiles = [(0, 0), (256, 64), (128, 64), (128, 128), (256, 128), (256, 256), (224, 256), (224, 128)]
print('%-42s' % 'M x K -> N (taps, res) x count' + ''.join('%10s' % ('auto' if t == (0, 0) else '%dx%d' % t) for t in tiles))
for key, (cnt, a) in sorted(calls.items(), key=lambda kv: -kv[0][0] * kv[0][2] * kv[1][0]):
    if key[1] > 512 or key[3] != 1 or key[0] < 10000: continue
    row = '%-42s' % ('%d x %d -> %d (res %d) x%d' % (key[0], key[1], key[2], key[4], cnt))
    for tile in tiles:
        eng.pair_tile = tile
        try:
            row += '%10.1f' % min(t_us(lambda: orig(*a)) for _ in range(2))
        except Exception as e:
            row += '%10s' % 'n/a'
    eng.pair_tile = (0, 0)
    print(row, flush=True)
