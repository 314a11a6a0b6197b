"""Round 5: cycle stamps (s_memtime) per phase of a K step of k_gemm_pair<256, TN, conv> -- load issue / fragment reads + MFMA issue / vmcnt(0) wait /
barrier -- summed over every wave and K step of one launch (scratch/r5/ko/lib_TS.so, built from the product source with stamps added).
    gpurun -- python scratch/r5/time_pair_ts.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robustart_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'r5', 'ko', 'lib_TS.so')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'fp32x')
eng.fused_tail_pair = False
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
calls = {}
orig = eng._gemm_pair
def rec(*a):
    src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols = a[:10]
    calls.setdefault((batch * grid[0] * grid[1], k_per_tap * len(taps), n_cols, len(taps)), a)
    return orig(*a)
eng._gemm_pair = rec
eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize()
eng._gemm_pair = orig
lib = _lib.load()
lib.rart_debug_ts_read.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
for key in [(50176, 2304, 256, 9), (50176, 1024, 256, 1), (12544, 4608, 512, 9), (50176, 256, 1024, 1), (802816, 256, 64, 1)]:
    if key not in calls:
        continue
    a = calls[key]
    orig(*a); torch.cuda.synchronize(); lib.rart_debug_ts_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(*a); e1.record(); torch.cuda.synchronize()
    lib.rart_debug_ts_read(buf)
    n = buf[4]           # sum over lane-0 waves of KT = waves x KT
    print(key, 'us %.1f' % (e0.elapsed_time(e1) * 1e3), 'per wave and K step (100 MHz ticks x cycles?):',
          {k: round(buf[i] / max(n, 1), 1) for i, k in enumerate(['issue', 'reads+mfma', 'vmcnt_wait', 'barrier'])}, 'wave-steps', n)
