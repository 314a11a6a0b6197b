import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robustart_amd import _lib
lib = _lib.load()
os.environ['RART_PAIR_SPLIT'] = '0'
K, N = 3072, 768
a = torch.randn(2, 256 * 256, K, device='cuda').bfloat16(); w = (torch.randn(2, N, K, device='cuda') * 0.05).bfloat16()
out = torch.empty(2, 256 * 256, N, device='cuda', dtype=torch.bfloat16)
for mt in (64, 85, 128, 170, 197, 256, 64, 85, 128, 170, 197, 256):
    d = _lib.GemmPairDesc()
    d.a_hi, d.a_lo, d.w_hi, d.w_lo = a[0].data_ptr(), a[1].data_ptr(), w[0].data_ptr(), w[1].data_ptr()
    d.dst_hi, d.dst_lo = out[0].data_ptr(), out[1].data_ptr()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc, d.w_rows = mt * 256, N, K, K, K, N, N
    d.tile_m, d.tile_n = 256, 256
    _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
torch.cuda.synchronize()
