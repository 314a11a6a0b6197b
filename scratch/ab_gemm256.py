"""A/B of the 256 x 256 GEMM on the ViT-B/16 engine (forward, forward + backward-to-input) at B = 256."""
import sys, time; sys.path.insert(0, '/root/repo')
import torch
from robustart_amd import _lib
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
eng = ViTEngine(get_model({'type': 'vit_base'}).eval(), 'cuda')
lib = _lib.load()
B = 256
x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
def t(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = {}
for en in (0, 1, 0, 1):
    lib.rart_igemm_set_gemm256(en)
    lg, _, g, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    out[en] = (lg.clone(), g.clone())
    print('gemm256=%d  fwd %.2f ms   fwd+bwd-to-input %.2f ms' % (en, t(lambda: eng.logits(x, MEAN, STD)), t(lambda: eng.forward_backward(x, MEAN, STD, y, 0))), flush=True)
a, b = out[0], out[1]
print('logits max diff %.4f (scale %.2f); grad cos %.6f' % ((a[0] - b[0]).abs().max().item(), a[0].abs().max().item(),
      torch.nn.functional.cosine_similarity(a[1].flatten().double(), b[1].flatten().double(), dim=0).item()))
