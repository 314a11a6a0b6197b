"""Round 5: every ImageNet-C corruption at every severity on B = 256 (rotating buffers), us per batch by events -> the long tail of the generator."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robustart_amd.noise import imagenet_c as C
B, NP = 256, 4
g = torch.Generator().manual_seed(11)
src = [torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).cuda() for _ in range(NP)]
dst = [torch.empty_like(s) for s in src]
C.set_frost_textures(list(np.random.RandomState(0).randint(0, 256, (6, 300, 300, 3)).astype(np.uint8)))
res = {}
for cid, nm in enumerate(C.CORRUPTION_NAMES):
    row = []
    for sev in range(1, 6):
        C.corrupt_batch_(src[0], cid, sev, seed=0, sample_offset=0, out=dst[0]); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(3):
            C.corrupt_batch_(src[(r + 1) % NP], cid, sev, seed=0, sample_offset=r * B, out=dst[(r + 1) % NP])
        e1.record(); torch.cuda.synchronize()
        row.append(round(e0.elapsed_time(e1) / 3 * 1e3, 1))
    res[nm] = row
    print('%-18s' % nm + ''.join('%10.1f' % v for v in row), flush=True)
tot = sum(sum(v) for v in res.values())
print('all 19 x 5: %.1f ms per 256 source images = %.0f corrupted images/s' % (tot / 1e3, 95 * B / (tot * 1e-6)))
json.dump(res, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'gpurun_out', 'r05_all_severities.json'), 'w'), indent=1)
