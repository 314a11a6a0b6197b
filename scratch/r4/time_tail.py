"""Round 4: where the time of rart_conv3x3_tail_pair goes at layer1's shape (B = 256, 56 x 56, C = 64): the product kernel beside knock-out
builds (scratch/r4/build_ko.sh: KO_MAIN = 2 of the 18 K steps, KO_TAIL = one of the four 64-channel chunks, KO_W3 = no table loads in the
1x1, KO_RES = no skip loads, KO_STORE = no stores).  gpurun -- python scratch/r4/time_tail.py"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:          # child: time one library
    from robustart_amd import _lib
    if sys.argv[1] != 'product':
        _lib.LIB_PATH = sys.argv[1]
    import torch
    lib = _lib.load()
    C = int(os.environ.get('C', '64'))
    B, H, W, N = 256, (56 if C == 64 else 28), (56 if C == 64 else 28), 4 * C
    P = B * H * W
    g = torch.Generator().manual_seed(0)

    def pair(*shape):
        t = torch.randn(*shape, generator=g).cuda()
        hi = t.to(torch.bfloat16)
        return torch.stack([hi, (t - hi.float()).to(torch.bfloat16)]).contiguous()
    x, res, dst, dn = pair(P, C), pair(P, N), pair(P, N), pair(P, C)
    tab = torch.randn(C, 3 * 9 * C, generator=g).to(torch.bfloat16).cuda()
    tail, ntab = pair(N * C), pair(N * C)
    b2, b3 = torch.randn(C).cuda(), torch.randn(N).cuda()
    sm, so, sn = (torch.empty(P, c // 8, dtype=torch.uint8, device='cuda') for c in (C, N, C))
    out = {}
    for nxt in (False, True):
        d = _lib.ConvTailDesc()
        d.a_hi, d.a_lo, d.w_hi, d.w_lo = x[0].data_ptr(), x[1].data_ptr(), tab.data_ptr(), tab.data_ptr() + 2 * 9 * C
        d.t_hi, d.t_lo = tail[0].data_ptr(), tail[1].data_ptr()
        d.res_hi, d.res_lo, d.dst_hi, d.dst_lo = res[0].data_ptr(), res[1].data_ptr(), dst[0].data_ptr(), dst[1].data_ptr()
        d.batch, d.h, d.w, d.c_mid, d.ldw = B, H, W, C, 3 * 9 * C
        for i in range(9):
            d.tap_dy[i], d.tap_dx[i] = i // 3 - 1, i % 3 - 1
        d.bias_mid, d.bias_out, d.sign_mid, d.sign_out = b2.data_ptr(), b3.data_ptr(), sm.data_ptr(), so.data_ptr()
        d.relu_mid = d.relu_out = 1
        if nxt:
            d.n_hi, d.n_lo, d.dstn_hi, d.dstn_lo = ntab[0].data_ptr(), ntab[1].data_ptr(), dn[0].data_ptr(), dn[1].data_ptr()
            d.bias_next, d.sign_next, d.relu_next = b2.data_ptr(), sn.data_ptr(), 1
        for _ in range(3):
            _lib.check(lib.rart_conv3x3_tail_pair(ctypes.byref(d), _lib.stream_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _lib.check(lib.rart_conv3x3_tail_pair(ctypes.byref(d), _lib.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        out['next' if nxt else 'plain'] = round(e0.elapsed_time(e1) * 100, 1)      # us per launch
    print(json.dumps(out))
    sys.exit(0)
res = {}
ko = os.path.join(ROOT, 'scratch', 'r4', 'ko')
libs = ['product'] + sorted(os.path.join(ko, f) for f in os.listdir(ko) if f.endswith('.so')) if os.path.isdir(ko) else ['product']
for lib in libs:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), lib], capture_output=True, text=True)
    res[os.path.basename(lib)] = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
    print(os.path.basename(lib), res[os.path.basename(lib)], flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'r04_tail_knockouts_C%s.json' % os.environ.get('C', '64')), 'w'), indent=1)
