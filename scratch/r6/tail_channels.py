"""Round 6: the fused 3x3 + 1x1 tail kernels per layer: reference-precision ResNet-50 gradient evaluation at B = 256 with the tails on layer1 and
layer2 (default), layer1 only, layer2 only, nowhere.   gpurun -- python scratch/r6/tail_channels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'fp32x')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
def t(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ref = None
for rnd in range(2):
    for ch in ((64, 128), (64,), (128,), ()):
        eng.fused_tail_channels = ch
        fb = t(lambda: eng.forward_backward(x, MEAN, STD, y, 0)); f = t(lambda: eng.logits(x, MEAN, STD))
        l, _, g, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        if ref is None: ref = (l.clone(), g.clone())
        print('tails on %-10s: forward + backward %.3f ms, forward %.3f ms; bit-identical to the default: %s %s' % (ch, fb, f, torch.equal(l, ref[0]), torch.equal(g, ref[1])), flush=True)
