"""A/B of engine switches at B=256: python scratch/ab_engine.py halo_conv3x3 [sign_bit_masks ...]"""
import sys; sys.path.insert(0, '/root/repo')
import time, torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda')
B = 256
x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
def t(fn, n=8):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name in sys.argv[1:]:
    for rep in range(2):
        for v in (False, True):
            setattr(eng, name, v)
            print('%s=%s  fwd %.3f ms  fwd+bwd %.3f ms' % (name, v, t(lambda: eng.logits(x, MEAN, STD)),
                                                          t(lambda: eng.forward_backward(x, MEAN, STD, y, 0))), flush=True)
