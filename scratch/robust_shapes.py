"""Edge-shape robustness of the engines: batch 1 / odd batches / non-square inputs; logits vs torch fp32, finite gradients."""
import sys; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.resnet_torch import randomize_bn_stats
from robustart_amd.model.engine import ResNet50Engine
from robustart_amd.model.vit_engine import ViTEngine
from robustart_amd.model.train_engine import ResNet50TrainEngine
from robustart_amd.train.arena import label_smooth_ce
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
mean=torch.tensor(MEAN,device='cuda').view(1,3,1,1); std=torch.tensor(STD,device='cuda').view(1,3,1,1)
torch.manual_seed(0)
m=randomize_bn_stats(get_model({'type':'resnet50_official'})).eval().cuda()
eng=ResNet50Engine(m,'cuda')
ok=True
for shape in [(1,3,224,224),(3,3,96,160),(5,3,256,256),(7,3,32,64),(257,3,64,64)]:
    x=torch.rand(*shape,device='cuda'); y=torch.randint(0,1000,(shape[0],),device='cuda')
    lg,loss,grad,pred=eng.forward_backward(x,MEAN,STD,y,0)
    ref=m((x-mean)/std)
    err=float((lg-ref).abs().max()/ref.abs().max())
    fin=bool(torch.isfinite(grad).all()) and float(grad.abs().sum())>0
    xr=x.clone().requires_grad_(True); g,=torch.autograd.grad(torch.nn.functional.cross_entropy(m((xr-mean)/std),y,reduction='sum'),xr)
    a,b=grad.double().flatten(),g.double().flatten(); cos=float(a@b/(a.norm()*b.norm()))
    print('resnet50 eval', shape, 'logit err %.4f  grad finite %s  cos vs torch %.3f' % (err, fin, cos)); ok &= err<0.05 and fin and cos>0.5
vm=get_model({'type':'vit_base'}).eval().cuda(); ve=ViTEngine(vm,'cuda')
for B in (1,3):
    x=torch.rand(B,3,224,224,device='cuda'); y=torch.randint(0,1000,(B,),device='cuda')
    lg,loss,grad,pred=ve.forward_backward(x,MEAN,STD,y,0)
    ref=vm((x-mean)/std); err=float((lg-ref).abs().max()/ref.abs().max())
    print('vit eval B=%d logit err %.4f grad finite %s' % (B, err, bool(torch.isfinite(grad).all()))); ok &= err<0.05
mt=get_model({'type':'resnet50_official'}).cuda().train()
for p in mt.parameters(): p.grad=torch.zeros_like(p)
te=ResNet50TrainEngine(mt)
for shape in [(2,3,64,64),(3,3,96,160),(9,3,224,224)]:
    x=torch.rand(*shape,device='cuda'); y=torch.randint(0,1000,(shape[0],),device='cuda')
    lg=te.forward(x,False,MEAN,STD); lr,dl=label_smooth_ce(lg,y,0.1,1.0/shape[0]); te.backward(dl)
    fin=all(bool(torch.isfinite(p.grad).all()) for p in mt.parameters())
    print('resnet50 train', shape, 'loss %.3f grads finite %s' % (float(lr.mean()), fin)); ok &= fin
print('ALL OK' if ok else 'PROBLEM')
