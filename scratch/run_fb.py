import sys; sys.path.insert(0, '/root/repo')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda')
B = 256
x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
for _ in range(4):
    eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize()
