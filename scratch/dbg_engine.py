import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, torch.nn.functional as F
from test_engine_gpu import _setup, rb, MEAN, STD
from robustart_amd.noise.adv import logit_loss
m, eng = _setup()
B,HW=2,96
g=torch.Generator().manual_seed(5)
x=torch.rand(B,3,HW,HW,generator=g).cuda(); y=torch.randint(0,1000,(B,),generator=g).cuda()
logits, loss, grad, pred = eng.forward_backward(x, MEAN, STD, y, 0)
# reference with retained intermediates (fp64 cpu)
dt=torch.float64
xr=x.cpu().double().requires_grad_(True)
mean=torch.tensor(MEAN,dtype=dt).view(1,3,1,1); std=torch.tensor(STD,dtype=dt).view(1,3,1,1)
v=(xr-mean)*(1.0/std); hi=rb(v); v=hi+rb(v-hi)
def conv(c,t,relu,res=None):
    w=c.w_folded.to(torch.bfloat16).to(dt)
    o=F.conv2d(t,w,c.b_folded.to(dt),stride=c.stride,padding=c.pad)
    if res is not None: o=o+res
    if relu: o=torch.relu(o)
    return rb(o)
keep={}
t=conv(eng.stem,v,True); t.retain_grad(); keep['y1']=t
t=F.max_pool2d(t,3,2,1); t.retain_grad(); keep[-1]=t
for bi,(ca,cb,cc,ds) in enumerate(eng.blocks):
    a=conv(ca,t,True); b=conv(cb,a,True)
    sk=conv(ds,t,False) if ds is not None else t
    t=conv(cc,b,True,res=sk); t.retain_grad(); keep[bi]=t
p=rb(t.mean((2,3))); p.retain_grad(); keep['pool']=p
out=p@eng.fc_w[:1000].cpu().to(dt).t()+eng.fc_b.cpu().to(dt)
_,dl,_=logit_loss(out.detach().float().cuda(),y,0)
out.backward(dl.cpu().double())
def cmp(name,a,b):
    a=a.flatten().double().cpu(); b=b.flatten().double().cpu()
    cos=(a@b/(a.norm()*b.norm()+1e-30)).item(); print(f'{name:10s} cos {cos:.5f} |a| {a.norm():.4e} |b| {b.norm():.4e}')
print('logit err', (logits.cpu().double()-out.detach()).abs().max().item(), out.abs().max().item())
cmp('dpool', eng._buf['dpool'].float(), keep['pool'].grad)
for k in range(15,-2,-1):
    ref=keep[k].grad*(keep[k].detach()>0)
    cmp('g_out_%d'%k, eng._buf['g_out_%d'%k].float().permute(0,3,1,2), ref)
ref=keep['y1'].grad*(keep['y1'].detach()>0)
cmp('g_y1', eng._buf['g_y1'].float().permute(0,3,1,2), ref)
cmp('grad', grad, xr.grad)
