"""Round 6: phase stamps (lab build -DRART_PP_STAMPS) of the ping-pong pair GEMM on ViT-B/16's launch shapes at B = 256: prologue / K loop /
epilogue cycles per workgroup.   gpurun -- python scratch/r6/vit_pp_stamps.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robustart_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'r6', 'pp', 'lib_STAMPS.so')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
lib = _lib.load()
os.environ['RART_PAIR_SPLIT'] = '0'
lib.rart_gemm_pair_set_schedule(int(os.environ.get('SCHED', 1)))
torch.manual_seed(0)
eng = ViTEngine(get_model({'type': 'vit_base_patch16_224'}).eval(), 'cuda', 'fp32x')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
calls = {}
orig = eng._gemm_pair
def rec(*a, **kw):
    key = (a[3], a[4], a[5], kw.get('flags', 0), kw.get('res') is not None)
    calls.setdefault(key, (a, kw))
    return orig(*a, **kw)
eng._gemm_pair = rec
eng.forward_backward(x, MEAN, STD, y, 0); torch.cuda.synchronize()
eng._gemm_pair = orig
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
for key, (a, kw) in sorted(calls.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2]):
    if key[0] < 1000: continue
    orig(*a, **kw); torch.cuda.synchronize()
    raw.rart_debug_pp_stamps(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): orig(*a, **kw)
    e1.record(); torch.cuda.synchronize()
    raw.rart_debug_pp_stamps(buf)
    us = e0.elapsed_time(e1) / 4 * 1e3
    r = {}
    for g in range(2):
        wg = max(1, buf[g * 16 + 15]); n = max(1, buf[g * 16 + 12])
        r['g%d' % g] = dict(prologue=round(buf[g * 16 + 13] / wg), loop=round(buf[g * 16 + 11] / wg), epilogue=round(buf[g * 16 + 14] / wg), kstep=round(buf[g * 16 + 11] / n), wgs=wg // 4)
    tiles = ((key[0] + 255) // 256) * ((key[1] + 255) // 256)
    print(key, '%.1f us, %d tiles (%.2f rounds): %.0f us per round' % (us, tiles, tiles / 256, us / -(-tiles // 256)), r, flush=True)
