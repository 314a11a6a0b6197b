"""Round 4: per-layer time of the weight gradient at B = 256 (every distinct conv shape of ResNet-50): rart_wgrad_direct_bf16 + fold / reduce
against the transpose_gather + implicit-GEMM path.  gpurun -- python scratch/r4/time_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.model import get_model
from robustart_amd.model.train_engine import ResNet50TrainEngine, _TConv
B = 256
model = get_model({'type': 'resnet50_official'}).cuda().train()
for p in model.parameters():
    p.grad = torch.zeros_like(p)
eng = ResNet50TrainEngine(model)
shapes = [(64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2), (256, 128, 1, 1, 56, 1), (128, 128, 3, 2, 56, 1),
          (256, 512, 1, 2, 56, 1), (128, 512, 1, 1, 28, 4), (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3), (512, 256, 1, 1, 28, 1),
          (256, 256, 3, 2, 28, 1), (512, 1024, 1, 2, 28, 1), (256, 1024, 1, 1, 14, 6), (1024, 256, 1, 1, 14, 5), (256, 256, 3, 1, 14, 5),
          (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 1), (1024, 2048, 1, 2, 14, 1), (512, 2048, 1, 1, 7, 3), (2048, 512, 1, 1, 7, 2),
          (512, 512, 3, 1, 7, 2)]
tot = {True: 0.0, False: 0.0}
print('%5s %5s r s %3s  x   direct_us  (TF/s)   gather_us')
for cin, cout, r, stride, H, cnt in shapes:
    conv = torch.nn.Conv2d(cin, cout, r, stride=stride, padding=r // 2, bias=False).cuda()
    conv.weight.grad = torch.zeros_like(conv.weight)
    tc = _TConv(conv, None, torch.device('cuda'), torch)
    x = torch.randn(B, H, H, cin, device='cuda').to(torch.bfloat16)
    oh = H // stride
    dz = torch.randn(B, oh, oh, cout, device='cuda').to(torch.bfloat16)
    t = {}
    for direct in (True, False):
        eng.direct_wgrad = direct
        for _ in range(2):
            eng._conv_wgrad(tc, dz, (oh, oh), x, (H, H))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            eng._conv_wgrad(tc, dz, (oh, oh), x, (H, H))
        e1.record(); torch.cuda.synchronize()
        t[direct] = e0.elapsed_time(e1) * 200
        tot[direct] += t[direct] * cnt
    fl = 2.0 * B * oh * oh * cin * cout * r * r
    print('%5d %5d %d %d %3d x%d  %9.1f  (%5.0f)  %9.1f' % (cin, cout, r, stride, H, cnt, t[True], fl / t[True] / 1e6, t[False]))
print('per step: direct %.2f ms, gather %.2f ms' % (tot[True] / 1e3, tot[False] / 1e3))
