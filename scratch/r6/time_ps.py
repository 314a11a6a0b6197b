"""Round 6: the persistent pair GEMM on the wide 1x1 launch shapes of the reference-precision ResNet-50 at B = 256: schedule 0 / 1 / 2."""
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
    calls.setdefault((batch * grid[0] * grid[1], k_per_tap * len(taps), n_cols, len(taps), a[13] is not None), a)
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
for key in [(50176, 256, 1024, 1, True), (12544, 512, 2048, 1, True), (200704, 256, 512, 1, False), (50176, 512, 1024, 1, False), (50176, 1024, 512, 1, False)]:
    a = calls[key]
    r = []
    for s in (0, 1, 2):
        lib.rart_gemm_pair_set_schedule(s)
        r.append(min(t_us(lambda: orig(*a)) for _ in range(3)))
    print(key, ' two-stage %.1f  ping-pong %.1f  persistent %.1f us' % tuple(r), flush=True)
for s in (1, 2, 1, 2):
    lib.rart_gemm_pair_set_schedule(s)
    print('schedule', s, 'fwd+bwd %.3f ms' % (t_us(lambda: eng.forward_backward(x, MEAN, STD, y, 0), 5) / 1e3), flush=True)
