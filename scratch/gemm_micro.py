import sys; sys.path.insert(0,'/root/repo')
import torch, time, ctypes
from robustart_amd import _lib
from robustart_amd.model.vit_engine import ViTEngine
lib=_lib.load()
eng = ViTEngine.__new__(ViTEngine); eng.lib=lib; eng.device=torch.device('cuda'); eng._buf={}
def run(M,K,N,iters=20):
    a=(torch.randn(M,K,device='cuda')*0.5).to(torch.bfloat16)
    w=(torch.randn((N+127)//128*128,K,device='cuda')*0.05).to(torch.bfloat16)
    out=torch.empty(M,N,device='cuda',dtype=torch.bfloat16)
    for _ in range(3): eng._gemm(a,w,out,M,K,N,K,N)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(iters): eng._gemm(a,w,out,M,K,N,K,N)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/iters
    ref=(a[:256].float()@w[:N].float().t())
    err=(out[:256].float()-ref).abs().max().item()
    print('M=%d K=%d N=%d: %.1f us  %.1f TF/s  err %.3g' % (M,K,N,dt*1e6, 2*M*K*N/dt/1e12, err))
shapes = [(50432,768,2304),(50432,768,3072),(50432,3072,768),(50432,768,768),(8192,8192,8192)] if len(sys.argv)<2 else [tuple(int(v) for v in sys.argv[1].split(','))]
for s in shapes: run(*s)
if len(sys.argv) < 2:
    print('--- BK=64 disabled ---')
    lib.rart_igemm_set_bk64_min_k(1 << 40)
    for s in shapes: run(*s)
