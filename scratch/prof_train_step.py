import sys, time; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.train_engine import ResNet50TrainEngine
from robustart_amd.train.arena import ParamArena, HipOptimizer, label_smooth_ce
B=256
torch.manual_seed(0)
model=get_model({'type':'resnet50_official'}).cuda().train()
arena=ParamArena(model)
opt=HipOptimizer(arena,'SGD',lr=0.01,momentum=0.9,nesterov=True,weight_decay=1e-4,ema_decay=0.9999)
eng=ResNet50TrainEngine(model,on_grad_ready=arena.grad_ready)
x=torch.rand(B,3,224,224,device='cuda'); y=torch.randint(0,1000,(B,),device='cuda')
mean,std=(0.485,0.456,0.406),(0.229,0.224,0.225)
for _ in range(6):
    l=eng.forward(x,False,mean,std); lr,dl=label_smooth_ce(l,y,0.1,1.0/B); eng.backward(dl)
    arena.finish_grad_exchange(); opt.step(1.0); eng.repack()
torch.cuda.synchronize()
