"""Round 6: B = 256 sits just above a pass boundary of the layer1 / layer2 kernels (B = 248: 18.7 ms, B = 252: 19.4 ms).  Does a gradient evaluation of
256 images run faster as a (256 - r)-image chain and an r-image chain on two streams?   gpurun -- python scratch/r6/split_batch_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
model = get_model({'type': 'resnet50_official'}).eval()
e1, e2 = ResNet50Engine(model, 'cuda', 'fp32x'), ResNet50Engine(model, 'cuda', 'fp32x')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
s2 = torch.cuda.Stream()
def t(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def whole(): e1.forward_backward(x, MEAN, STD, y, 0)
def split(r):
    def f():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            e2.forward_backward(x[256 - r:], MEAN, STD, y[256 - r:], 0)
            done = torch.cuda.Event(); done.record()
        e1.forward_backward(x[:256 - r], MEAN, STD, y[:256 - r], 0)
        torch.cuda.current_stream().wait_event(done)
    return f
for rnd in range(2):
    print('whole batch of 256: %.3f ms' % t(whole), flush=True)
    for r in (4, 8, 12, 16, 24, 32):
        print('  %3d + %2d on two streams: %.3f ms   (the %d-image chain alone: %.3f ms, the %d-image chain alone %.3f ms)' % (
            256 - r, r, t(split(r)), 256 - r, t(lambda: e1.forward_backward(x[:256 - r], MEAN, STD, y[:256 - r], 0)), r,
            t(lambda: e2.forward_backward(x[256 - r:], MEAN, STD, y[256 - r:], 0))), flush=True)
