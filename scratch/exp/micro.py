import sys, os; sys.path.insert(0,'/root/repo')
import torch, time, ctypes
from robustart_amd import _lib
_lib.LIB_PATH = os.environ['RART_LIB']
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
    return dt*1e6, 2*M*K*N/dt/1e12
shapes=[(50176,2304,256),(50176,1024,256),(50176,256,1024),(200704,128,512),(200704,512,128),(12544,512,2048),(8192,8192,8192),(802816,64,256)]
print(os.path.basename(os.path.dirname(_lib.LIB_PATH)).ljust(9), '  '.join('%7.1fus/%4.0fTF' % run(*s) for s in shapes))
