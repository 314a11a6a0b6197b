"""Round 6: ViT-B/16 gradient evaluation (PREC), ms per image against the batch size -- does the per-XCD pass quantisation of the
launches show at the whole-engine level, and is B = 256 a lucky or an unlucky size?   gpurun -- python scratch/r6/batch_sweep_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ViTEngine(get_model({'type': 'vit_base_patch16_224'}).eval(), 'cuda', os.environ.get('PREC', 'fp32x'))
def t(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 96, 128, 192, 256, 320, 384]:
    x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
    fb = min(t(lambda: eng.forward_backward(x, MEAN, STD, y, 0)) for _ in range(2))
    print('B %4d: %7.3f ms per gradient evaluation, %6.2f us per image, %7.0f images/s' % (B, fb, fb / B * 1e3, B / fb * 1e3), flush=True)
