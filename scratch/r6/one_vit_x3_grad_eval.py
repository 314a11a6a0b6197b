"""Target of the round-6 ViT passes: three reference-precision gradient evaluations of ViT-B/16 at B = 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
eng = ViTEngine(get_model({'type': 'vit_base_patch16_224'}).eval(), 'cuda', os.environ.get('PREC', 'fp32x'))
for _ in range(3): eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize()
