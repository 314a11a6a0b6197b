"""Round 6: phase stamps of the forward pair attention kernel (lab build -DRART_ATT_STAMPS): cycles per workgroup in K / V staging, and per wave
tile in Q load + S = K Q^T, soft-max + P V, the stores.   gpurun -- python scratch/r6/att_stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robustart_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'r6', 'pp', 'lib_ATT_STAMPS.so')
import torch
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
B, T, H, hd = 256, 197, 12, 64
qkv = torch.randn(2, B * T, 3 * H * hd, device='cuda').bfloat16()
att = torch.empty(2, B * T, H * hd, device='cuda', dtype=torch.bfloat16)
import numpy as np
buf = (ctypes.c_ulonglong * (4096 * 8 * 4))()
sp = _lib.stream_ptr()
for _ in range(2):
    _lib.check(lib.rart_vit_attention_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(att[0]), _lib.ptr(att[1]), B, T, H, hd, sp))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    _lib.check(lib.rart_vit_attention_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(att[0]), _lib.ptr(att[1]), B, T, H, hd, sp))
e1.record(); torch.cuda.synchronize()
raw.rart_debug_att_stamps(buf)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8, 4)[:3072, :7].astype(np.float64)
print('%.1f us per launch (3072 workgroups, 12 per CU)' % (e0.elapsed_time(e1) / 4 * 1e3))
for i, nm in enumerate(['K / V staging up to the barrier', 'Q load + S = K Q^T', 'soft-max + P V', 'normalise + stores (drained)']):
    print('%-34s median %7d  p10 %7d  p90 %7d ticks' % (nm, np.median(t[:, :, i]), np.percentile(t[:, :, i], 10), np.percentile(t[:, :, i], 90)))
print('sum of the medians %d ticks per workgroup' % sum(np.median(t[:, :, i]) for i in range(4)))
