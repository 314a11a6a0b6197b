import sys, time, ctypes; sys.path.insert(0,'/root/repo')
import torch
from robustart_amd import _lib
lib=_lib.load()
B,H=256,224
patches=torch.randn(B,112,112,152,device='cuda').to(torch.bfloat16)
grad=torch.empty(B,3,H,H,device='cuda')
std=(ctypes.c_float*3)(0.229,0.224,0.225)
def run(): _lib.check(lib.rart_engine_stem_col2im(_lib.ptr(patches),_lib.ptr(grad),B,H,H,152,std,_lib.stream_ptr()))
for _ in range(3): run()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
print('stem col2im %.1f us (%.2f TB/s of patch reads)' % (dt*1e6, patches.numel()*2/dt/1e12))
