"""Round 6: the bf16 256 x 256 x 64 GEMM of the transformer layers on round 2's two-stage loop (rart_igemm_set_gemm256(1)) and on the ping-pong
schedule (2), on the four ViT-B/16 shapes at B = 256 (M = 50 432) and on 8192^3, random operands; then ViT-B/16 forward and forward + backward."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robustart_amd import _lib
lib = _lib.load()
def gemm(M, K, N, a, w, out):
    d = _lib.ConvDesc()
    d.src, d.wgt, d.dst = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.batch, d.grid_h, d.grid_w = 1, M, 1
    d.src_h, d.src_w, d.src_pix_stride = M, 1, K
    d.k_per_tap, d.n_taps, d.sy, d.sx = K, 1, 1, 1
    d.n_cols, d.dst_h, d.dst_w = N, M, 1
    d.dst_sy, d.dst_sx, d.dst_oy, d.dst_ox, d.dst_pix_stride = 1, 1, 0, 0, N
    _lib.check(lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr()))
def t_us(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for M, K, N in [(50432, 768, 2304), (50432, 768, 768), (50432, 768, 3072), (50432, 3072, 768), (8192, 8192, 8192)]:
    a = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    w = (torch.rand(N, K, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    r = {}
    for rep in range(3):
        for mode in (1, 2):
            lib.rart_igemm_set_gemm256(mode)
            r.setdefault(mode, []).append(t_us(lambda: gemm(M, K, N, a, w, out)))
    fl = 2.0 * M * K * N
    print('%6d x %5d -> %5d   two-stage %7.1f us %6.0f TFLOP/s   ping-pong %7.1f us %6.0f TFLOP/s' % (M, K, N, min(r[1]), fl / min(r[1]) / 1e6, min(r[2]), fl / min(r[2]) / 1e6), flush=True)
from robustart_amd.model import get_model
from robustart_amd.model.vit_engine import ViTEngine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
torch.manual_seed(0)
v = ViTEngine(get_model({'type': 'vit_base'}).eval(), 'cuda')
x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
for rep in range(2):
    for mode in (1, 2):
        lib.rart_igemm_set_gemm256(mode)
        f = t_us(lambda: v.logits(x, MEAN, STD), 5) / 1e3
        fb = t_us(lambda: v.forward_backward(x, MEAN, STD, y, 0), 5) / 1e3
        print('gemm256 mode %d: ViT-B/16 bf16 forward %.2f ms, forward + backward %.2f ms' % (mode, f, fb), flush=True)
