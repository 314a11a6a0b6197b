"""Round 6: us per launch of the three pair attention kernels of ViT-B/16 (B = 256, 12 heads, 197 tokens) for a library build.
    python scratch/r6/time_att_pair.py [lib.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robustart_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != 'product': _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
lib = _lib.load()
B, T, H, hd = 256, 197, 12, 64
torch.manual_seed(0)
qkv = (torch.randn(2, B * T, 3 * H * hd, device='cuda') * torch.tensor([1.0, 2 ** -9], device='cuda').view(2, 1, 1)).bfloat16()
att = torch.empty(2, B * T, H * hd, device='cuda', dtype=torch.bfloat16)
sp = _lib.stream_ptr()
def f(): _lib.check(lib.rart_vit_attention_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(att[0]), _lib.ptr(att[1]), B, T, H, hd, sp))
for _ in range(3): f()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
print('%s: attention forward %.1f us per launch' % (sys.argv[1] if len(sys.argv) > 1 else 'product', best))
