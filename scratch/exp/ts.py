import sys, os; sys.path.insert(0,'/root/repo')
import torch, time, ctypes, numpy as np
from robustart_amd import _lib
_lib.LIB_PATH = '/root/repo/scratch/exp/ts/librobustart_hip.so'
from robustart_amd.model.vit_engine import ViTEngine
lib=_lib.load()
raw=ctypes.CDLL(_lib.LIB_PATH)
eng = ViTEngine.__new__(ViTEngine); eng.lib=lib; eng.device=torch.device('cuda'); eng._buf={}
def run(M,K,N):
    a=(torch.randn(M,K,device='cuda')*0.5).to(torch.bfloat16)
    w=(torch.randn((N+127)//128*128,K,device='cuda')*0.05).to(torch.bfloat16)
    out=torch.empty(M,N,device='cuda',dtype=torch.bfloat16)
    for _ in range(3): eng._gemm(a,w,out,M,K,N,K,N)
    torch.cuda.synchronize()
    buf=np.zeros(8192*16,dtype=np.uint64)
    raw.rart_dbg_read(buf.ctypes.data_as(ctypes.c_void_p))
    t=buf.reshape(8192,16).astype(np.int64)
    nb=min(8192, ((M+127)//128+7)//8*8*((N+127)//128))
    t=t[:nb]; t=t[t[:,0]>0]
    t0=t[:,0].min()
    e=t[:,[3,6,7,8,4]]; de=np.diff(e,axis=1)
    print('   epilogue median: pass0 lds-write %d | pass0 read+store %d | pass1 lds-write %d | pass1 read+store %d' % tuple(np.median(de,axis=0).astype(int)))
    d=np.diff(t[:,:6],axis=1)
    print('M=%d K=%d N=%d blocks=%d  span %.1f us (100MHz ticks?)' % (M,K,N,len(t),(t[:,5].max()-t0)))
    print('   median ticks: issue-loads %d | loads-arrive+lds+barrier %d | main loop %d | epilogue issue %d | store drain %d ; total %d' % tuple(list(np.median(d,axis=0).astype(int))+[int(np.median(t[:,5]-t[:,0]))]))
    print('   p90    ticks: %s' % np.percentile(d,90,axis=0).astype(int))
    # start-time distribution: how many blocks start in first 10% of span
    st=np.sort(t[:,0]-t0); print('   block start quantiles', st[[0,len(st)//4,len(st)//2,3*len(st)//4,-1]])
for s in [(200704,128,512),(50176,256,1024),(50176,1024,256),(50176,2304,256),(802816,64,256)]: run(*s)
