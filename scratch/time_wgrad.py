import sys, time; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd import _lib
from robustart_amd.model import get_model
from robustart_amd.model.train_engine import ResNet50TrainEngine, _TConv
torch.manual_seed(0)
model=get_model({'type':'resnet50_official'}).cuda().train()
for p in model.parameters(): p.grad=torch.zeros_like(p)
eng=ResNet50TrainEngine(model)
B=256
lib=eng.lib
orig_check=_lib.check
ev=[]
import robustart_amd.model.train_engine as TE
def timed(name, fn):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); r=fn(); e1.record(); ev.append((name,e0,e1)); return r
# wrap lib functions
class LibWrap:
    def __init__(self, lib): self._l=lib
    def __getattr__(self, n):
        f=getattr(self._l,n)
        if n in ('rart_transpose_gather_bf16','rart_conv_igemm_bf16','rart_wgrad_reduce_f32'):
            return lambda *a: timed(n, lambda: f(*a))
        return f
eng.lib=LibWrap(lib)
for cin,cout,r,stride,H in [(64,64,3,1,56),(64,256,1,1,56),(256,64,1,1,56),(128,128,3,1,28),(256,256,3,1,14),(512,512,3,1,7),(1024,256,1,1,14),(512,2048,1,1,7),(256,512,1,2,56)]:
    conv=torch.nn.Conv2d(cin,cout,r,stride=stride,padding=r//2,bias=False).cuda(); conv.weight.grad=torch.zeros_like(conv.weight)
    tc=_TConv(conv,None,torch.device('cuda'),torch)
    x=torch.randn(B,H,H,cin,device='cuda').to(torch.bfloat16); oh=H//stride
    dz=torch.randn(B,oh,oh,cout,device='cuda').to(torch.bfloat16)
    for _ in range(2): eng._conv_wgrad(tc,dz,(oh,oh),x,(H,H))
    ev.clear()
    for _ in range(3): eng._conv_wgrad(tc,dz,(oh,oh),x,(H,H))
    torch.cuda.synchronize()
    agg={}
    for n,e0,e1 in ev: agg.setdefault(n,[]).append(e0.elapsed_time(e1)*1e3)
    fl=2.0*B*oh*oh*cin*r*r*cout
    tt=[sum(v)/3 for v in agg.values()]
    print('cin %4d cout %4d %dx%d/%d H=%2d : transposes %7.1f us  gemm %7.1f us (%.0f TF)  reduce %6.1f us   total %.1f us' % (cin,cout,r,r,stride,H, sum(agg['rart_transpose_gather_bf16'])/3, sum(agg['rart_conv_igemm_bf16'])/3, fl/(sum(agg['rart_conv_igemm_bf16'])/3)/1e6, sum(agg['rart_wgrad_reduce_f32'])/3, sum(tt)))
