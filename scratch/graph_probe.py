"""Does a captured hipGraph of one ResNet-50 gradient evaluation run faster than the eager launch sequence?"""
import sys, time; sys.path.insert(0, '/root/repo')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda')
B = 256
x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager fwd+bwd %.3f ms' % t(lambda: eng.forward_backward(x, MEAN, STD, y, 0)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = eng.forward_backward(x, MEAN, STD, y, 0)
    print('graph fwd+bwd %.3f ms' % t(lambda: g.replay()))
except Exception as e:
    print('capture failed:', repr(e)[:300])
