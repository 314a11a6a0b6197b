"""Target of a kernel trace: the heaviest ImageNet-C generators (elastic, spatter, zoom, glass, fog, gaussian_blur, motion, snow, defocus) at
severities 1, 3, 5 on B = 256, two calls each (the second one is the one to read)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robustart_amd.noise import imagenet_c as C
B = 256
g = torch.Generator().manual_seed(11)
src = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).cuda()
dst = torch.empty_like(src)
names = sys.argv[1:] or ['elastic_transform', 'spatter', 'zoom_blur', 'glass_blur', 'fog', 'gaussian_blur', 'motion_blur', 'snow', 'defocus_blur']
for nm in names:
    cid = C.CORRUPTION_NAMES.index(nm)
    for sev in (1, 3, 5):
        for r in range(2):
            C.corrupt_batch_(src, cid, sev, seed=0, sample_offset=r * B, out=dst)
        torch.cuda.synchronize()
