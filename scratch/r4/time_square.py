"""Round 4: Square-L2 query loop against the same number of bare forwards (the per-query sign rows now come from the
device generator, no numpy draw + H2D copy per query).  B = 64, ResNet-50 bf16 engine, 300 queries.
    gpurun -- python scratch/r4/time_square.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robustart_amd.model import get_model                     # noqa: E402
from robustart_amd.model.engine import EngineModel            # noqa: E402
from robustart_amd.noise import adv                           # noqa: E402

torch.manual_seed(0)
f = EngineModel(get_model({'type': 'resnet50_official'}).eval(), takes_normalized=False)
g = torch.Generator().manual_seed(1)
B, Q = 64, 300
x = torch.rand(B, 3, 224, 224, generator=g).cuda()
y = f(x).argmax(1)
out = {}
for norm, eps in (('L2', 0.05), ('L1', 2.0), ('Linf', 0.05 / 255)):        # tiny radii: nobody gets fooled, every query runs
    def run():
        if norm == 'Linf':
            return adv.square_perturb(f, x, y, eps, Q, seed=1, sample_offset=0)
        return adv.square_lp_perturb(f, x, y, norm, eps, Q, seed=1, sample_offset=0)
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    out[norm] = (time.perf_counter() - t0) / Q * 1e3
for _ in range(20):
    f(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(Q):
    f(x)
torch.cuda.synchronize()
fw = (time.perf_counter() - t0) / Q * 1e3
res = {'batch': B, 'queries': Q, 'forward_ms': fw, 'square_ms_per_query': out, 'ratio_to_forward': {k: v / fw for k, v in out.items()}}
print(json.dumps(res, indent=1))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/r04_square_vs_forward.json', 'w'), indent=1)
