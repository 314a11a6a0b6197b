import sys, time; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.vit_train_engine import ViTTrainEngine
from robustart_amd.train.arena import ParamArena, HipOptimizer, label_smooth_ce
B=int(sys.argv[1]) if len(sys.argv)>1 else 128
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
model=get_model({'type':'vit_base'}).cuda().train()
arena=ParamArena(model)
opt=HipOptimizer(arena,'AdamW',lr=5e-4,weight_decay=0.05,ema_decay=0.9999)
eng=ViTTrainEngine(model,on_grad_ready=arena.grad_ready)
x=torch.rand(B,3,224,224,device='cuda'); y=torch.randint(0,1000,(B,),device='cuda')
def t(fn,n=3):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
losses=[]
def step():
    l=eng.forward(x,False,MEAN,STD); lr,dl=label_smooth_ce(l,y,0.1,1.0/B); eng.backward(dl)
    arena.finish_grad_exchange(); opt.step(1.0); eng.repack(); losses.append(float(lr.mean()))
print('ViT-B/16 B=%d HIP train step %.1f ms   (repack %.1f ms)  loss %.3f -> %.3f' % (B, t(step), t(eng.repack), losses[0], losses[-1]))
m2=get_model({'type':'vit_base'}).cuda().train()
o2=torch.optim.AdamW(m2.parameters(),lr=5e-4,weight_decay=0.05)
mean=torch.tensor(MEAN,device='cuda').view(1,3,1,1); std=torch.tensor(STD,device='cuda').view(1,3,1,1)
xn=(x-mean)/std
def tstep():
    with torch.autocast('cuda',dtype=torch.bfloat16):
        out=m2(xn)
    loss=torch.nn.functional.cross_entropy(out.float(),y,label_smoothing=0.1)
    o2.zero_grad(set_to_none=True); loss.backward(); o2.step()
print('torch autocast-bf16 train step %.1f ms' % t(tstep))
# solver integration (3 iterations)
from robustart_amd.train import cls_solver as S
class A: pass
a=A(); a.engine='hip'; a.train_engine='hip'; a.max_iter=3
cfg={'model':{'type':'vit_base'},'data':{'fake_size':64,'batch_size':32,'input_size':224},'label_smooth':0.1,'max_iter':3,
     'ema':{'enable':True,'kwargs':{'decay':0.9999}},'optimizer':{'type':'AdamW','no_wd':{'norm':True,'fc':True},'kwargs':{'weight_decay':0.05}},
     'lr_scheduler':{'kwargs':{'base_lr':1e-5,'warmup_lr':5e-4}},'adv_train':{'eps':'4/255','steps':2,'rel_stepsize':0.4}}
loss,_=S.train(cfg,a,0,1,torch.device('cuda'))
print('cls_solver adversarial training on ViT-B/16 (HIP attack + train engines): final loss %.3f' % loss)
