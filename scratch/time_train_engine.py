import sys, time; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.train_engine import ResNet50TrainEngine
from robustart_amd.train.arena import ParamArena, HipOptimizer, label_smooth_ce
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
torch.manual_seed(0)
model=get_model({'type':'resnet50_official'}).cuda().train()
arena=ParamArena(model)
opt=HipOptimizer(arena,'SGD',lr=0.01,momentum=0.9,nesterov=True,weight_decay=1e-4,ema_decay=0.9999)
eng=ResNet50TrainEngine(model,on_grad_ready=arena.grad_ready)
x=torch.rand(B,3,224,224,device='cuda'); y=torch.randint(0,1000,(B,),device='cuda')
mean,std=(0.485,0.456,0.406),(0.229,0.224,0.225)
def t(fn,n=5):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
state={}
def fwd(): state['l']=eng.forward(x,False,mean,std)
def fb():
    l=eng.forward(x,False,mean,std); lr,dl=label_smooth_ce(l,y,0.1,1.0/B); eng.backward(dl)
def step():
    fb(); arena.finish_grad_exchange(); opt.step(1.0); eng.repack()
print('B=%d  fwd %.2f ms   fwd+bwd %.2f ms   full step %.2f ms   repack %.2f ms   opt %.2f ms' % (B, t(fwd), t(fb), t(step), t(eng.repack), t(lambda: opt.step(1.0))))
# torch scaffold: autocast bf16 channels_last
m2=get_model({'type':'resnet50_official'}).cuda().train().to(memory_format=torch.channels_last)
o2=torch.optim.SGD(m2.parameters(),lr=0.01,momentum=0.9,nesterov=True,weight_decay=1e-4)
xn=((x-torch.tensor(mean,device='cuda').view(1,3,1,1))/torch.tensor(std,device='cuda').view(1,3,1,1)).contiguous(memory_format=torch.channels_last)
def tstep():
    with torch.autocast('cuda',dtype=torch.bfloat16):
        out=m2(xn)
    loss=torch.nn.functional.cross_entropy(out.float(),y,label_smoothing=0.1)
    o2.zero_grad(set_to_none=True); loss.backward(); o2.step()
print('torch autocast-bf16 channels_last full step %.2f ms' % t(tstep))
