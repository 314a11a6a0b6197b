import sys; sys.path.insert(0,'/root/repo')
import torch, copy
import torch.nn.functional as F
from robustart_amd.model import get_model
from robustart_amd.model.train_engine import ResNet50TrainEngine
def cos(a,b):
    a,b=a.detach().double().flatten(),b.detach().double().flatten(); return float((a@b)/(a.norm()*b.norm()+1e-30))
def _rb(t): return t + (t.to(torch.bfloat16).float() - t).detach()
torch.manual_seed(0)
B,S=16,64
model=get_model({'type':'resnet50_official'}).cuda().train()
x01=torch.rand(B,3,S,S,device='cuda')
mean,std=(0.485,0.456,0.406),(0.229,0.224,0.225)
ref=copy.deepcopy(model)
for p in model.parameters(): p.grad=torch.zeros_like(p)
eng=ResNet50TrainEngine(model)
logits=eng.forward(x01,False,mean,std)
A=eng.acts
mt=torch.tensor(mean,device='cuda').view(1,3,1,1); st=torch.tensor(std,device='cuda').view(1,3,1,1)
xn=(x01-mt)/st
def nhwc(t): return t.float().permute(0,3,1,2)
def conv(x,m): return _rb(F.conv2d(x,_rb(m.weight),stride=m.stride,padding=m.padding))
def bn(z,m,relu,res=None):
    y=F.batch_norm(z,None,None,m.weight,m.bias,training=True,eps=m.eps)
    if res is not None: y=y+res
    return _rb(y.relu() if relu else y)
z1=conv(xn,ref.conv1); print('z1',cos(nhwc(A['z1']),z1), float((nhwc(A['z1'])-z1).abs().max()))
y1=bn(z1,ref.bn1,True); print('y1',cos(nhwc(A['y1']),y1), float((nhwc(A['y1'])-y1).abs().max()))
# BN of engine's own z1 through torch: isolates BN
y1b=bn(nhwc(A['z1']),ref.bn1,True); print('y1 from engine z1', cos(nhwc(A['y1']),y1b), float((nhwc(A['y1'])-y1b).abs().max()))
x=F.max_pool2d(y1,3,2,1); print('p1',cos(nhwc(A['p1']),x))
k=0
for layer in (ref.layer1,ref.layer2,ref.layer3,ref.layer4):
    for blk in layer:
        xe,xhw,za,ya,zb,yb,zc,zd,out,ohw=A['b%d'%k]
        # teacher-forced: feed the ENGINE's block input to the emulated block, compare each stage
        xin=nhwc(xe)
        za_t=conv(xin,blk.conv1); ya_t=bn(za_t,blk.bn1,True)
        zb_t=conv(nhwc(ya),blk.conv2); yb_t=bn(nhwc(zb),blk.bn2,True)
        zc_t=conv(nhwc(yb),blk.conv3)
        if blk.downsample is not None:
            zd_t=conv(xin,blk.downsample[0]); sk=bn(nhwc(zd),blk.downsample[1],False); c_zd=cos(nhwc(zd),zd_t)
        else: sk=xin; c_zd=1.0
        out_t=bn(nhwc(zc),blk.bn3,True,res=sk)
        print('block %2d  za %.6f ya %.6f zb %.6f yb %.6f zc %.6f zd %.6f out %.6f' % (k,cos(nhwc(za),za_t),cos(nhwc(ya),bn(nhwc(za),blk.bn1,True)),cos(nhwc(zb),zb_t),cos(nhwc(yb),yb_t),cos(nhwc(zc),zc_t),c_zd,cos(nhwc(out),out_t)))
        k+=1
