"""Round 5: the eleven ImageNet-S resize operators on B = 256 images of 375 x 500 (the common ImageNet size), 'val' transform: short side -> 256
(rh, rw = 256, 341), centre crop 224.  us per batch by events; algorithmic bytes = source + output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.noise import imagenet_s as S
B, H, W = 256, 375, 500
g = torch.Generator().manual_seed(3)
src = [torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).cuda() for _ in range(3)]
rh, rw = 256, 341
crop = ((rh - 224) // 2, (rw - 224) // 2, 224, 224)
nbytes = B * (H * W * 3 + 224 * 224 * 3)
for name, fid in list(S.PIL_FILTERS.items()) + list(S.CV_MODES.items()):
    fn = S.pil_resize if name.startswith('pil') else S.cv_resize
    fn(src[0], (rh, rw), fid, crop); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(6):
        fn(src[r % 3], (rh, rw), fid, crop)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 6 * 1e3
    print('%-16s %8.1f us per 256 images   %7.0f k images/s   %6.0f GB/s algorithmic' % (name, us, B / us * 1e3, nbytes / us / 1e3), flush=True)
