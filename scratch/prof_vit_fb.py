import sys; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ViTEngine(get_model({'type': 'vit_base'}).eval(), 'cuda')
x = torch.rand(256,3,224,224,device='cuda'); y = torch.randint(0,1000,(256,),device='cuda')
for _ in range(4): eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize()
