"""One ResNet-50 gradient evaluation (forward + backward-to-input) at B = 256 after a warm one: the target of per-kernel PMC passes."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
for _ in range(3): eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize()
