import sys; sys.path.insert(0,'/root/repo')
import torch, time
from robustart_amd import _lib
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
lib=_lib.load()
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda')
veng = ViTEngine(get_model({'type': 'vit_base'}).eval(), 'cuda')
B=256
x = torch.rand(B,3,224,224,device='cuda'); y = torch.randint(0,1000,(B,),device='cuda')
u8 = torch.randint(0,256,(B,224,224,3),dtype=torch.uint8,device='cuda')
def t(fn,n=8):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for rep in range(2):
    a=t(lambda: eng.forward_backward(x, MEAN, STD, y, 0)); b=t(lambda: eng.logits(x, MEAN, STD)); c=t(lambda: veng.logits_from_u8(u8, MEAN, STD))
    print('resnet fwd+bwd %.2f ms  fwd %.2f ms   vit fwd %.2f ms' % (a,b,c))
