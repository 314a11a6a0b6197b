import sys, time; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
m = get_model({'type': 'vit_base'}).eval()
eng = ViTEngine(m, 'cuda')
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
x = torch.rand(B,3,224,224,device='cuda'); y = torch.randint(0,1000,(B,),device='cuda')
def t(fn,n=4):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print('ViT-B/16 B=%d  fwd %.2f ms   fwd+bwd-to-input %.2f ms' % (B, t(lambda: eng.logits(x, MEAN, STD)), t(lambda: eng.forward_backward(x, MEAN, STD, y, 0))))
mc = m.cuda()
mean = torch.tensor(MEAN, device='cuda').view(1,3,1,1); std = torch.tensor(STD, device='cuda').view(1,3,1,1)
def tfb():
    xr = x.clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = mc((xr-mean)/std)
    loss = torch.nn.functional.cross_entropy(out.float(), y, reduction='sum')
    g, = torch.autograd.grad(loss, xr)
for p in mc.parameters(): p.requires_grad_(False)
print('torch autocast-bf16 fwd+bwd-to-input %.2f ms' % t(tfb))
