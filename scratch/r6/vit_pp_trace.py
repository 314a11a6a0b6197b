"""Round 6: per-workgroup trace (lab build -DRART_PP_STAMPS: s_memtime at entry / K loop end / exit + HW_ID) of ONE ping-pong launch per ViT
shape: for every CU the gap between one workgroup's exit and the next one's entry.   gpurun -- python scratch/r6/vit_pp_trace.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robustart_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'r6', 'pp', 'lib_STAMPS.so')
import torch
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
lib = _lib.load()
torch.manual_seed(0)
eng = ViTEngine(get_model({'type': 'vit_base_patch16_224'}).eval(), 'cuda', 'fp32x')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
calls = {}
orig = eng._gemm_pair
def rec(*a, **kw):
    calls.setdefault((a[3], a[4], a[5], kw.get('flags', 0), kw.get('res') is not None), (a, kw))
    return orig(*a, **kw)
eng._gemm_pair = rec
eng.forward_backward(x, MEAN, STD, y, 0); torch.cuda.synchronize()
eng._gemm_pair = orig
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (4096 * 4))()
for key in [(50432, 2304, 768, 0, False), (50432, 3072, 768, 64, False), (50432, 768, 3072, 0, False)]:
    a, kw = calls[key]
    for _ in range(3): orig(*a, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(*a, **kw); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    raw.rart_debug_pp_trace(buf)
    tiles = ((key[0] + 255) // 256) * ((key[1] + 255) // 256)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4)[:min(tiles, 4096)].astype(np.int64)
    t = t[t[:, 0] > 0]
    hw = t[:, 3] & 0xFFFFFFFF; xcc = (t[:, 3] >> 32) & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    span = t[:, 2].max() - t[:, 0].min()
    print(key, '%.1f us, %d workgroups traced, %d distinct CUs; span %d ticks -> %.3f ticks per ns' % (us, len(t), len(np.unique(cu)), span, span / us / 1e3))
    gaps, durs, kl, ep = [], [], [], []
    for c in np.unique(cu):
        q = t[cu == c]; q = q[np.argsort(q[:, 0])]
        gaps += list(q[1:, 0] - q[:-1, 2]); durs += list(q[:, 2] - q[:, 0]); kl += list(q[:, 1] - q[:, 0]); ep += list(q[:, 2] - q[:, 1])
    gaps = np.array(gaps)
    print('   per workgroup: entry->loop end %d, loop end->exit %d, total %d ticks; gap exit->next entry on the same CU: median %d, mean %d, p10 %d, p90 %d (n = %d)'
          % (np.median(kl), np.median(ep), np.median(durs), np.median(gaps), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 90), len(gaps)))
    x8 = cu >> 8
    for xc in np.unique(x8)[:3]:
        q = t[x8 == xc]; c8 = cu[x8 == xc]
        span = q[:, 2].max() - q[:, 0].min()
        busy = (q[:, 2] - q[:, 0]).sum(); inloop = (q[:, 1] - q[:, 0]).sum()
        d = q[:, 2] - q[:, 0]
        cnt = np.array([np.sum(c8 == c) for c in np.unique(c8)])
        start = np.array([q[c8 == c][:, 0].min() for c in np.unique(c8)]) - q[:, 0].min()
        end = q[:, 2].max() - np.array([q[c8 == c][:, 2].max() for c in np.unique(c8)])
        print('   XCD %d: %d CUs, %d workgroups, span %d ticks (%.2f GHz if the span is the launch); in a workgroup %.3f of CU time, in its K loop %.3f; workgroup ticks p10 %d p50 %d p90 %d; per CU count %s; start skew p50 %d max %d; idle tail p50 %d max %d'
              % (xc, len(cnt), len(q), span, span / us / 1e3, busy / (len(cnt) * span), inloop / (len(cnt) * span), np.percentile(d, 10), np.median(d), np.percentile(d, 90), dict(zip(*np.unique(cnt, return_counts=True))), np.median(start), start.max(), np.median(end), end.max()))
