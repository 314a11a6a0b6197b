"""gaussian_noise B = 256: per-launch events vs a captured graph of 40 back-to-back launches between two events."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd.noise import imagenet_c as C
B, H, W, npairs, launches = 256, 224, 224, 9, 40
g = torch.Generator().manual_seed(7)
src = [torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).cuda() for _ in range(npairs)]
dst = [torch.empty_like(s) for s in src]
for i in range(npairs): C.corrupt_batch_(src[i], 0, 3, seed=0, sample_offset=0, out=dst[i])
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
for i in range(launches):
    ev[i][0].record(); C.corrupt_batch_(src[i % npairs], 0, 3, seed=0, sample_offset=i * B, out=dst[i % npairs]); ev[i][1].record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
print('per-launch events avg us %.2f' % (sum(ms) / len(ms) * 1e3))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(launches): C.corrupt_batch_(src[i % npairs], 0, 3, seed=0, sample_offset=i * B, out=dst[i % npairs])
e1.record(); torch.cuda.synchronize()
print('two events around 40 eager launches: us per launch %.2f' % (e0.elapsed_time(e1) / launches * 1e3))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for i in range(launches): C.corrupt_batch_(src[i % npairs], 0, 3, seed=0, sample_offset=i * B, out=dst[i % npairs])
    gr.replay(); torch.cuda.synchronize()
    for _ in range(3):
        e0.record(st); gr.replay(); e1.record(st); torch.cuda.synchronize()
        print('graph of 40 launches: us per launch %.2f' % (e0.elapsed_time(e1) / launches * 1e3))
