"""Round 4: every rart_gemm_pair_bf16 launch shape of one reference-precision ResNet-50 gradient evaluation at B = 256, replayed with
each (tile_m, tile_n) of the kernel -> which tile each layer wants (the policy in csrc/gemm_pair.hip is chosen from this table).
    gpurun -- python scratch/r4/sweep_pair_tiles.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robustart_amd.model import get_model                     # noqa: E402
from robustart_amd.model.engine import ResNet50Engine         # noqa: E402

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
B = int(os.environ.get('B', '256'))
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'fp32x')
x = torch.rand(B, 3, 224, 224, device='cuda')
y = torch.randint(0, 1000, (B,), device='cuda')
calls = {}
orig = eng._gemm_pair


def rec(*a):
    src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols = a[:10]
    key = (batch * grid[0] * grid[1], k_per_tap * len(taps), n_cols, len(taps), a[16][0])      # M, K, N, taps, stride
    calls.setdefault(key, [0, a])[0] += 1
    return orig(*a)


eng._gemm_pair = rec
eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize()
eng._gemm_pair = orig
res = {}
tot_auto = tot_best = 0.0
print('%9s %6s %5s %4s %3s | %s' % ('M', 'K', 'N', 'taps', 'x', '  '.join('%dx%d' % t for t in [(0, 0), (256, 256), (256, 128), (256, 64), (128, 256), (128, 128), (128, 64)])))
for key, (cnt, a) in sorted(calls.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1][0]):
    M, K, N, nt, st = key
    row = {}
    for tile in [(0, 0), (256, 256), (256, 128), (256, 64), (128, 256), (128, 128), (128, 64)]:
        if tile[1] and tile[1] > max(64, (N + 63) // 64 * 64) and tile != (0, 0):
            if tile[1] // 2 >= N:
                continue
        eng.pair_tile = tile
        for _ in range(2):
            orig(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            orig(*a)
        e1.record()
        torch.cuda.synchronize()
        row['%dx%d' % tile] = e0.elapsed_time(e1) / 5 * 1e3
    eng.pair_tile = (0, 0)
    best = min(row, key=row.get)
    tot_auto += cnt * row['0x0']
    tot_best += cnt * row[best]
    res['%d_%d_%d_%d_%d' % key] = dict(count=cnt, us=row, best=best)
    print('%9d %6d %5d %4d %3d | %s  best %s' % (M, K, N, nt, cnt, '  '.join('%7.1f' % row.get('%dx%d' % t, float('nan'))
          for t in [(0, 0), (256, 256), (256, 128), (256, 64), (128, 256), (128, 128), (128, 64)]), best))
print('sum over a gradient evaluation: automatic %.1f us, best per shape %.1f us' % (tot_auto, tot_best))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/r04_pair_tile_sweep.json', 'w'), indent=1)
