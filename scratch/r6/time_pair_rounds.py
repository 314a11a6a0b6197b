"""Round 6: is a pair-GEMM launch quantised in rounds of 256 tiles?  Plain product, K = 3072 -> N = 768 (three 256-column tiles per row tile),
row tiles swept; and the same rows on 256 x 128 tiles.   gpurun -- python scratch/r6/time_pair_rounds.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from robustart_amd import _lib
lib = _lib.load()
os.environ['RART_PAIR_SPLIT'] = '0'
K, N = int(os.environ.get('K', 3072)), int(os.environ.get('N', 768))
MT = 260
a = torch.randn(2, MT * 256, K, device='cuda').bfloat16()
w = (torch.randn(2, N, K, device='cuda') * 0.05).bfloat16()
out = torch.empty(2, MT * 256, N, device='cuda', dtype=torch.bfloat16)
def run(mt, tn, sched=1):
    d = _lib.GemmPairDesc()
    d.a_hi, d.a_lo, d.w_hi, d.w_lo = a[0].data_ptr(), a[1].data_ptr(), w[0].data_ptr(), w[1].data_ptr()
    d.dst_hi, d.dst_lo = out[0].data_ptr(), out[1].data_ptr()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc, d.w_rows = mt * 256, N, K, K, K, N, N
    d.tile_m, d.tile_n = 256, tn
    lib.rart_gemm_pair_set_schedule(sched)
    def f(): _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    return best
for tn in (256, 128):
    for mt in (20, 43, 64, 85, 86, 100, 128, 150, 170, 171, 197, 220, 256):
        tiles = mt * (N // tn)
        us = run(mt, tn)
        print('tile 256 x %3d: %3d row tiles = %4d tiles (%.2f x 256): %7.1f us  -> %6.1f us per 256 tiles, %.2f of peak' % (tn, mt, tiles, tiles / 256, us, us / (tiles / 256), 6.0 * mt * 256 * N * K / us / 1e6 / 2500), flush=True)
