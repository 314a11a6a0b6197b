import sys; sys.path.insert(0,'/root/repo')
import torch, copy
import torch.nn.functional as F
from robustart_amd.model import get_model
from robustart_amd.model.train_engine import ResNet50TrainEngine
from robustart_amd.train.arena import label_smooth_ce
def cos(a,b):
    a,b=a.double().flatten(),b.double().flatten(); return float((a@b)/(a.norm()*b.norm()+1e-30))
torch.manual_seed(0)
B,S=int(sys.argv[1]) if len(sys.argv)>1 else 16, int(sys.argv[2]) if len(sys.argv)>2 else 64
model=get_model({'type':'resnet50_official'}).cuda().train()
x01=torch.rand(B,3,S,S,device='cuda'); y=torch.randint(0,1000,(B,),device='cuda')
mean,std=(0.485,0.456,0.406),(0.229,0.224,0.225)
ref=copy.deepcopy(model)
for p in model.parameters(): p.grad=torch.zeros_like(p)
eng=ResNet50TrainEngine(model)
logits=eng.forward(x01,False,mean,std)
feats={}
def hook(name):
    def f(m,i,o): feats[name]=o.detach()
    return f
ref.relu.register_forward_hook(hook('y1')); ref.maxpool.register_forward_hook(hook('p1'))
k=0
for li,layer in enumerate((ref.layer1,ref.layer2,ref.layer3,ref.layer4)):
    for blk in layer:
        blk.register_forward_hook(hook('b%d'%k)); k+=1
mt=torch.tensor(mean,device='cuda').view(1,3,1,1); st=torch.tensor(std,device='cuda').view(1,3,1,1)
def run(m, amp):
    with torch.autocast('cuda',dtype=torch.bfloat16,enabled=amp):
        return m((x01-mt)/st)
out=run(ref,False)
f32={k_:v.clone() for k_,v in feats.items()}
A=eng.acts
print('y1', cos(A['y1'].float().permute(0,3,1,2), f32['y1']))  # note: ref.relu hook fires multiple times? top-level relu only once
print('p1', cos(A['p1'].float().permute(0,3,1,2), f32['p1']))
for bi in range(16):
    o=A['b%d'%bi][8]
    print('block',bi, cos(o.float().permute(0,3,1,2), f32['b%d'%bi]))
print('logits', cos(logits,out))
ref2=copy.deepcopy(ref); out2=run(ref2,True)
print('torch autocast-bf16 vs fp32 logits cos', cos(out2.float(),out))
