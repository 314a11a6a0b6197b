"""Target of a kernel trace: two bf16 gradient evaluations of ResNet-50 at B = 256 (the second one is the one to read)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'bf16')
for _ in range(2): eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize()
