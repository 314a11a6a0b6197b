"""Time rart_engine_stem_bwd_fused alone at B = 256 (random pooled gradient and argmax codes)."""
import sys; sys.path.insert(0, '/root/repo')
import ctypes, torch
from robustart_amd import _lib
from robustart_amd.model.engine import ResNet50Engine
lib = _lib.load()
B, H, W = 256, 224, 224
g = torch.Generator().manual_seed(5)
wb = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(torch.bfloat16).float()
dp = torch.randn(B, H // 4, W // 4, 64, generator=g).to(torch.bfloat16).cuda()
cd = torch.randint(0, 10, (B, H // 4, W // 4, 64), generator=g, dtype=torch.uint8)
cd[cd == 9] = 15
cd = cd.cuda()
wt = ResNet50Engine._stem_bwd_table(wb).cuda()
grad = torch.empty(B, 3, H, W, device='cuda')
stdf = (ctypes.c_float * 3)(1, 1, 1)
run = lambda: _lib.check(lib.rart_engine_stem_bwd_fused(_lib.ptr(dp), _lib.ptr(cd), _lib.ptr(wt), _lib.ptr(grad), B, H, W, stdf, _lib.stream_ptr()))
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print('stem backward fused: %.1f us per launch' % (e0.elapsed_time(e1) / 20 * 1e3))
