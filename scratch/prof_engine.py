import sys; sys.path.insert(0,'/root/repo')
import torch, time
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
import robustart_amd.model.engine as E
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
m = get_model({'type': 'resnet50_official'}).eval()
eng = ResNet50Engine(m, 'cuda')
B = 256
x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
# annotate launches: wrap _gemm to record shapes
rows = []
orig = eng._gemm
def wrapped(src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, dst_hw, dst_pix, **kw):
    Mrows = batch*grid[0]*grid[1]; K = k_per_tap*len(taps)
    bytes_ = Mrows*K*2/ max(1,(len(taps) if len(taps)>1 and k_per_tap!=32 else 1)) + wgt.numel()*2 + Mrows*n_cols*(4 if kw.get('flags',0)&2 else 2)
    if kw.get('res') is not None: bytes_ += Mrows*n_cols*2
    if kw.get('mask') is not None: bytes_ += Mrows*n_cols*2
    rows.append((Mrows, K, n_cols, len(taps), bytes_))
    return orig(src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, dst_hw, dst_pix, **kw)
eng._gemm = wrapped
for _ in range(2): eng.forward_backward(x, MEAN, STD, y, 0)
rows.clear(); eng.profile = []
torch.cuda.synchronize(); t0=time.perf_counter()
eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize(); wall=time.perf_counter()-t0
prof = eng.profile; eng.profile=None
tot=0
print('%9s %6s %6s %4s %9s %8s %8s' % ('M','K','N','taps','us','TF/s','GB/s'))
agg={}
for (Mr,K,N,T,by),(fl,a,b) in zip(rows,prof):
    us=a.elapsed_time(b)*1e3; tot+=us
    key=(Mr,K,N,T); agg.setdefault(key,[0,0,fl,by]); agg[key][0]+=us; agg[key][1]+=1
for key,(us,cnt,fl,by) in sorted(agg.items(), key=lambda kv:-kv[1][0]):
    print('%9d %6d %6d %4d %9.1f %8.1f %8.1f  x%d' % (*key, us, fl*cnt/us/1e6, by*cnt/us/1e3, cnt))
print('total gemm us', tot, 'wall ms (profiled, serialised by events)', wall*1e3)
# un-profiled wall for fwd+bwd and fwd only
for name,fn in (('fwd+bwd', lambda: eng.forward_backward(x, MEAN, STD, y, 0)), ('fwd', lambda: eng.logits(x, MEAN, STD))):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(name, 'ms', (time.perf_counter()-t0)/5*1e3)
