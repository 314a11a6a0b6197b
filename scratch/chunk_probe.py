"""Does running layer1/layer2 of ResNet-50 in sub-batches (whose activations fit the 256 MiB Infinity Cache) beat one
B=256 pass?  Times eng.logits / eng.forward_backward per image at several batch sizes and prints the per-shape igemm
table for B=64 vs B=256 (per image)."""
import sys; sys.path.insert(0, '/root/repo')
import time
import torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda')


def t(fn, n=6):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for B in (32, 64, 128, 256):
    x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
    a = t(lambda: eng.logits(x, MEAN, STD)); b = t(lambda: eng.forward_backward(x, MEAN, STD, y, 0))
    print('B=%3d fwd %.3f ms (%.2f us/img)   fwd+bwd %.3f ms (%.2f us/img)' % (B, a, a / B * 1e3, b, b / B * 1e3), flush=True)

import robustart_amd.model.engine as E
tabs = {}
for B in (64, 256):
    x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
    eng.forward_backward(x, MEAN, STD, y, 0)
    rec = []
    orig = eng._gemm
    def wrapped(src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, *a, **k); e1.record()
        rec.append(((grid[0] * grid[1], k_per_tap * len(taps), n_cols, len(taps)), e0, e1))
    eng._gemm = wrapped
    eng.forward_backward(x, MEAN, STD, y, 0)
    torch.cuda.synchronize()
    eng._gemm = orig
    d = {}
    for key, e0, e1 in rec:
        d.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
    tabs[B] = d
print('%8s %6s %6s %4s  %10s %10s  ratio' % ('pix/img', 'K', 'N', 'taps', 'us/img@64', 'us/img@256'))
for key in sorted(tabs[256], key=lambda k: -sum(tabs[256][k])):
    a = sum(tabs[64].get(key, [0])) / 64; b = sum(tabs[256][key]) / 256
    print('%8d %6d %6d %4d  %10.3f %10.3f  %.2f  x%d' % (key + (a, b, a / b if b else 0, len(tabs[256][key]))))
print('sum us/img: B=64 %.2f  B=256 %.2f' % (sum(sum(v) for v in tabs[64].values()) / 64, sum(sum(v) for v in tabs[256].values()) / 256))
