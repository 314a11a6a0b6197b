"""Round 5: the long tail of the all-severity sweep under rocprofv3 --kernel-trace --stats (per-kernel time of spatter 1-3, glass, elastic 1-2, frost, fog, motion, snow)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robustart_amd.noise import imagenet_c as C
B = 256
g = torch.Generator().manual_seed(11)
src = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).cuda()
dst = torch.empty_like(src)
C.set_frost_textures(list(np.random.RandomState(0).randint(0, 256, (6, 300, 300, 3)).astype(np.uint8)))
cases = [(n, s) for n, ss in (('spatter', (1, 3)), ('glass_blur', (3,)), ('elastic_transform', (1, 2, 3)), ('frost', (1,)), ('fog', (3,)),
                              ('motion_blur', (3, 5)), ('snow', (3,)), ('zoom_blur', (2,))) for s in ss]
only = sys.argv[1:] 
for nm, sev in cases:
    if only and nm not in only:
        continue
    cid = C.CORRUPTION_NAMES.index(nm)
    for r in range(3):
        C.corrupt_batch_(src, cid, sev, seed=0, sample_offset=r * B, out=dst)
    torch.cuda.synchronize()
