"""Round 5: rart_layernorm_bf16 on ViT-B/16's [256 x 197][768] activations, rotating buffers: us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robustart_amd import _lib
lib = _lib.load()
rows, d = 256 * 197, 768
x = [torch.randn(rows, d, device='cuda').bfloat16() for _ in range(4)]
o = [torch.empty_like(x[0]) for _ in range(4)]
g = torch.randn(d, device='cuda'); b = torch.randn(d, device='cuda')
def go(i):
    _lib.check(lib.rart_layernorm_bf16(_lib.ptr(x[i]), _lib.ptr(g), _lib.ptr(b), _lib.ptr(o[i]), rows, d, d, d, 1e-6, _lib.stream_ptr()))
go(0); torch.cuda.synchronize()
ref = torch.nn.functional.layer_norm(x[0].float(), (d,), g, b, 1e-6)
print('max abs diff vs torch fp32 layer_norm of the same bf16 input: %.4f (bf16 output)' % (o[0].float() - ref).abs().max().item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(40): go(r % 4)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 40 * 1e3
print('%.1f us per launch, %.0f GB/s on input + output' % (us, rows * d * 4 / us / 1e3))
