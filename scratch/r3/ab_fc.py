"""A/B of the classifier-head kernel inside the engine: forward and forward + backward at B = 256 with small_m_fc on / off."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for on in (True, False):
        eng.small_m_fc = on
        print('small_m_fc', on, 'fwd %.3f ms  fwd+bwd %.3f ms' % (t(lambda: eng.logits(x, MEAN, STD)), t(lambda: eng.forward_backward(x, MEAN, STD, y, 0))), flush=True)
