"""Round 4: ViT-B/16 reference-precision ('fp32x') engine at B = 256 -- forward and forward + backward-to-input times beside the
bf16 engine, and the per-kind breakdown of the pair GEMM launches (gpurun -- python scratch/r4/time_vit_x3.py)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robustart_amd.model import get_model                     # noqa: E402
from robustart_amd.model.vit_engine import ViTEngine          # noqa: E402

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
B = int(os.environ.get('B', '256'))
torch.manual_seed(0)
m = get_model({'type': 'vit_base'}).eval()
g = torch.Generator().manual_seed(1)
x = torch.rand(B, 3, 224, 224, generator=g).cuda()
y = torch.randint(0, 1000, (B,), generator=g).cuda()
out = {}
for prec in ('bf16', 'fp32x'):
    eng = ViTEngine(m, 'cuda', precision=prec)

    def t(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    out[prec] = {'fwd_ms': t(lambda: eng.logits(x, MEAN, STD)), 'fwd_bwd_ms': t(lambda: eng.forward_backward(x, MEAN, STD, y, 0))}
    if prec == 'fp32x':
        eng.profile = []
        eng.forward_backward(x, MEAN, STD, y, 0)
        torch.cuda.synchronize()
        prof, eng.profile = eng.profile, None
        secs = sum(a.elapsed_time(b) for _, a, b, _ in prof) * 1e-3
        fl = sum(f for f, _, _, _ in prof)
        big = [(f, a.elapsed_time(b)) for f, a, b, _ in prof if f > 1e11]
        out[prec]['pair_gemm'] = {'launches': len(prof), 'ms': secs * 1e3, 'issued_tflops': fl / secs / 1e12,
                                  'big_launches': len(big), 'big_ms': sum(t_ for _, t_ in big),
                                  'big_issued_tflops': sum(f for f, _ in big) / (sum(t_ for _, t_ in big) * 1e-3) / 1e12}
    del eng
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/r04_vit_x3_times.json', 'w'), indent=1)
