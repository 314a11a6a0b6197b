"""Kernel-trace target for ViT-B/16 at B = 256: 4 evaluation forwards (logits) and 4 gradient evaluations (forward + backward-to-input).
rocprofv3 --kernel-trace --stats -- python scratch/prof_vit_fb.py; profiles/summarize_rocpd.py -> profiles/r0N_vit_fwd_bwd_kernel_stats.csv"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ViTEngine(get_model({'type': 'vit_base'}).eval(), 'cuda')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
for name, fn in (('fwd', lambda: eng.logits(x, MEAN, STD)), ('fwd+bwd', lambda: eng.forward_backward(x, MEAN, STD, y, 0))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): fn()
    torch.cuda.synchronize(); print(name, 'ms %.3f' % ((time.perf_counter() - t0) / 4 * 1e3), flush=True)
