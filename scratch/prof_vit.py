import sys; sys.path.insert(0,'/root/repo')
import torch, time
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
m = get_model({'type': 'vit_base'}).eval()
eng = ViTEngine(m, 'cuda')
B=256
u8 = torch.randint(0,256,(B,224,224,3),dtype=torch.uint8,device='cuda')
for _ in range(2): eng.logits_from_u8(u8, MEAN, STD)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5): eng.logits_from_u8(u8, MEAN, STD)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print('vit-b/16 fwd B=256: %.2f ms  %.0f img/s  %.1f TFLOP/s' % (dt*1e3, B/dt, 35.1e9*B/dt/1e12))
x = torch.rand(4,3,224,224,device='cuda')
mg = m.cuda()
mean=torch.tensor(MEAN,device='cuda').view(1,3,1,1); std=torch.tensor(STD,device='cuda').view(1,3,1,1)
ref = mg((x-mean)/std); got = eng.logits(x, MEAN, STD)
print('err', (got-ref).abs().max().item(), 'scale', ref.abs().max().item())
mb = mg.to(torch.bfloat16)
xb = ((torch.rand(B,3,224,224,device='cuda')-mean)/std).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(2): mb(xb)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): mb(xb)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print('torch-rocm bf16 eager fwd B=256: %.2f ms %.0f img/s' % (dt*1e3, B/dt))
