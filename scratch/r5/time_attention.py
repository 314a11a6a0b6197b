"""Round 5: ViT-B/16's fused attention (k_vit_attention<7>, B = 256, 12 heads, 197 tokens) in isolation on rotating buffers: us per launch
in the engine's interleaved [token][3 x 768] layout, and the same arithmetic with every head as its own 'image' (H = 1: rows of 384 contiguous
bytes instead of 128-byte pieces 4 608 bytes apart) -- is the strided gather what the launch waits for?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd import _lib
lib = _lib.load()
T, hd = 197, 64
def run(B, H, reps=12, nbuf=4):
    qkv = [torch.randn(B * T, 3 * H * hd, device='cuda').bfloat16() for _ in range(nbuf)]
    att = [torch.empty(B * T, H * hd, device='cuda', dtype=torch.bfloat16) for _ in range(nbuf)]
    sp = _lib.stream_ptr()
    _lib.check(lib.rart_vit_attention(_lib.ptr(qkv[0]), _lib.ptr(att[0]), B, T, H, hd, sp)); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        _lib.check(lib.rart_vit_attention(_lib.ptr(qkv[r % nbuf]), _lib.ptr(att[r % nbuf]), B, T, H, hd, sp))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    nbytes = B * T * H * hd * 2 * 4
    print('B %5d  heads %2d: %7.1f us per launch  (%5.0f GB/s on q, k, v, out; %5.1f TFLOP/s)' % (B, H, us, nbytes / us / 1e3, 4 * B * H * T * T * hd / us / 1e6), flush=True)
run(256, 12)
run(256 * 12, 1)
run(64, 12)
run(64 * 12, 1)
