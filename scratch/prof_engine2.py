"""Per-launch table of one ResNet-50 gradient evaluation (forward + backward-to-input) at B = 256: every igemm launch with
its algorithmic FLOPs and HBM bytes, the measured time (events on the launch stream) and the roofline floor
max(FLOPs / 2.5 PFLOP/s, bytes / 8 TB/s).  Output -> profiles/r0N_igemm_per_shape.txt."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, time, torch
if os.environ.get('RART_LIB'):          # A/B against another build of the library
    from robustart_amd import _lib as _l; _l.LIB_PATH = os.environ['RART_LIB']
from robustart_amd.model import get_model
from robustart_amd.model.engine import ResNet50Engine
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
PEAK_F, PEAK_B = 2.5e15, 8.0e12
torch.manual_seed(0)
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', os.environ.get('PREC', 'bf16'))   # PREC=fp32x: the reference-precision chain
for a in sys.argv[1:]:
    if '=' in a:
        k, v = a.split('='); setattr(eng, k, bool(int(v)))
import os
if os.environ.get('BK64'):
    eng.lib.rart_igemm_set_bk64_min_k(int(os.environ['BK64']))
B = 256
x = torch.rand(B, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (B,), device='cuda')
rows = []
orig = eng._gemm
def wrapped(src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, dst_hw, dst_pix, **kw):
    M = batch * grid[0] * grid[1]; K = k_per_tap * len(taps)
    stride = kw.get('stride', (1, 1))
    # unique source bytes: the pixels the row grid touches once (taps re-read them from cache), capped by the tensor
    src_bytes = min(src.numel() * src.element_size(), M * K * 2) if len(taps) > 1 else M * K * 2
    by = src_bytes + K * n_cols * 2 + M * n_cols * (4 if kw.get('flags', 0) & 2 else 2)
    if kw.get('res') is not None: by += M * n_cols * 2
    mk = kw.get('mask')
    if mk is not None: by += M * n_cols * (2 if mk.dtype != torch.uint8 else 0.125)
    if kw.get('sign_out') is not None: by += M * n_cols * 0.125
    fl = 2.0 * M * K * n_cols
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, dst_hw, dst_pix, **kw); e1.record()
    rows.append(((M, K, n_cols, len(taps)), fl, by, e0, e1))
    return r
orig_halo = eng._halo
def wrapped_halo(src, w, dst, B_, hw, ch, taps, **kw):
    M = B_ * hw[0] * hw[1]
    by = 2 * M * ch * 2 + 9 * ch * ch * 2 + (M * ch * 0.125 if (kw.get('mask') is not None or kw.get('sign') is not None) else 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_halo(src, w, dst, B_, hw, ch, taps, **kw); e1.record()
    rows.append(((M, 9 * ch, ch, 99), 2.0 * M * 9 * ch * ch, by, e0, e1))      # taps column 99 = k_conv3x3_halo
    return r
orig_bneck = eng._bneck
def wrapped_bneck(x_, w1, w2, w3, b1, b2, b3, m1, m2, m3, out, B_, hw, c_io, c_mid, taps, backward, w4=None, c_in=None):
    M = B_ * hw[0] * hw[1]
    cin = c_io if w4 is None else c_in
    by = M * (c_io + cin) * 2 + ((c_io + cin) * c_mid + 9 * c_mid * c_mid + (0 if w4 is None else c_in * c_io)) * 2 + sum(M * c * 0.125 for c, m_ in ((c_mid, m1), (c_mid, m2), (c_io, m3)) if m_ is not None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_bneck(x_, w1, w2, w3, b1, b2, b3, m1, m2, m3, out, B_, hw, c_io, c_mid, taps, backward, w4, c_in); e1.record()
    k_all = (2 * c_io + 9 * c_mid) if w4 is None else (c_in + 9 * c_mid + c_io + c_in * c_io // c_mid)
    rows.append(((M, k_all, c_mid, 98), 2.0 * M * c_mid * k_all, by, e0, e1))   # taps column 98 = fused block
    return r
orig_b14 = eng._bneck14
def wrapped_b14(x_, w1, w2, w3, b1, b2, b3, m1, m2, m3, out, B_, hw, c_io, c_mid, taps, backward, fn=None):
    M = B_ * hw[0] * hw[1]
    by = 2 * M * c_io * 2 + (2 * c_io * c_mid + 9 * c_mid * c_mid) * 2 + sum(M * c * 0.125 for c, m_ in ((c_mid, m1), (c_mid, m2), (c_io, m3)) if m_ is not None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_b14(x_, w1, w2, w3, b1, b2, b3, m1, m2, m3, out, B_, hw, c_io, c_mid, taps, backward, fn); e1.record()
    rows.append(((M, 2 * c_io + 9 * c_mid, c_mid, {14: 97, 28: 96, 7: 95}[hw[0]]), 2.0 * M * c_mid * (2 * c_io + 9 * c_mid), by, e0, e1))   # taps column 97 / 96 = fused layer3 / layer2 block
    return r
orig_s2 = eng._bneck_s2
def wrapped_s2(x_, ca, cb, cc, ds, m1, m2, m3, out, B_, xhw):
    pin, pout = B_ * xhw[0] * xhw[1], B_ * xhw[0] * xhw[1] // 4
    fl = 2.0 * (pin * ca.cin * ca.cout + pout * (9 * cb.cin * cb.cout + cc.cin * cc.cout + ds.cin * ds.cout))
    by = pin * ca.cin * 2 + pout * cc.cout * 2 + (ca.cin * ca.cout + 9 * cb.cin * cb.cout + cc.cin * cc.cout + ds.cin * ds.cout) * 2 + \
        sum(n_ * c * 0.125 for n_, c, m_ in ((pin, ca.cout, m1), (pout, cb.cout, m2), (pout, cc.cout, m3)) if m_ is not None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_s2(x_, ca, cb, cc, ds, m1, m2, m3, out, B_, xhw); e1.record()
    rows.append(((pout, ca.cin * 4 + 9 * cb.cin + cc.cin + ds.cin, cc.cout, 94), fl, by, e0, e1))   # taps column 94 = fused stride-2 block (forward)
    return r
orig_s2b = eng._bneck_s2_bwd
def wrapped_s2b(g_, ca, cb, cc, ds, m2, m1, m0, dx_, B_, xhw):
    pin, pout = B_ * xhw[0] * xhw[1], B_ * xhw[0] * xhw[1] // 4
    fl = 2.0 * (pin * ca.cin * ca.cout + pout * (9 * cb.cin * cb.cout + cc.cin * cc.cout + ds.cin * ds.cout))
    by = pin * ca.cin * 2 + pout * cc.cout * 2 + (ca.cin * ca.cout + 9 * cb.cin * cb.cout + cc.cin * cc.cout + ds.cin * ds.cout) * 2 + \
        (pin * ca.cout + pout * cb.cout + (pin * ca.cin if m0 is not None else 0)) * 0.125
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_s2b(g_, ca, cb, cc, ds, m2, m1, m0, dx_, B_, xhw); e1.record()
    rows.append(((pout, ca.cin * 4 + 9 * cb.cin + cc.cin + ds.cin, cc.cout, 93), fl, by, e0, e1))   # taps column 93 = fused stride-2 block (backward)
    return r
orig_tail = eng._tail
def wrapped_tail(src, wgt, taps, tail, dst, batch, hw, c_mid, **kw):
    M = batch * hw[0] * hw[1]
    pair_b = 4      # bytes per element of a pair tensor
    by = M * c_mid * pair_b + M * 4 * c_mid * pair_b * (2 if kw.get('res') is not None else 1) + (9 * c_mid * c_mid + 4 * c_mid * c_mid) * pair_b + \
        sum(M * c * 0.125 for c, m_ in ((c_mid, kw.get('mask_mid')), (c_mid, kw.get('sign_mid')), (4 * c_mid, kw.get('mask_out')), (4 * c_mid, kw.get('sign_out'))) if m_ is not None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_tail(src, wgt, taps, tail, dst, batch, hw, c_mid, **kw); e1.record()
    rows.append(((M, 9 * c_mid + 4 * c_mid, c_mid, 92), 2.0 * M * c_mid * (9 * c_mid + 4 * c_mid), by, e0, e1))   # taps column 92 = 3x3 + 1x1 expansion, one launch (fp32x)
    return r
for _ in range(2): eng.forward_backward(x, MEAN, STD, y, 0)
eng._tail = wrapped_tail
eng._bneck14 = wrapped_b14
eng._bneck_s2_bwd = wrapped_s2b
eng._bneck_s2 = wrapped_s2
eng._gemm = wrapped
eng._halo = wrapped_halo
eng._bneck = wrapped_bneck
eng.forward_backward(x, MEAN, STD, y, 0)
rows.clear()
eng.forward_backward(x, MEAN, STD, y, 0)
torch.cuda.synchronize()
eng._gemm = orig
eng._tail = orig_tail
eng._halo = orig_halo
eng._bneck = orig_bneck
eng._bneck14 = orig_b14
eng._bneck_s2 = orig_s2
eng._bneck_s2_bwd = orig_s2b
agg = {}
for key, fl, by, a, b in rows:
    us = a.elapsed_time(b) * 1e3
    e = agg.setdefault(key, [0.0, 0, 0.0, 0.0]); e[0] += us; e[1] += 1; e[2] += fl; e[3] += by
print('%9s %6s %6s %4s %4s %9s %8s %8s %9s %6s' % ('M', 'K', 'N', 'taps', 'x', 'us', 'TF/s', 'GB/s', 'floor_us', 'frac'))
tot = totf = 0.0
for key, (us, cnt, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    floor = max(fl / PEAK_F, by / PEAK_B) * 1e6
    tot += us; totf += floor
    print('%9d %6d %6d %4d %4d %9.1f %8.1f %8.1f %9.1f %6.3f' % (*key, cnt, us, fl / us / 1e6, by / us / 1e3, floor, floor / us))
print('(taps 99 = the LDS-resident 3x3 kernels, 98 / 97 / 96 / 95 = the fused Bottleneck kernels of layer1 / layer3 / layer2 / layer4: one row = 1x1 + 3x3 + 1x1; 94 / 93 = the fused stride-2 first blocks of layer2 / layer3, forward / backward; 92 = 3x3 + 1x1 expansion of the fp32x engine, one launch)')
print('conv launches %d: measured %.1f us, roofline floor %.1f us, fraction of the per-layer floor %.3f' %
      (len(rows), tot, totf, totf / tot))
for name, fn in (('fwd+bwd', lambda: eng.forward_backward(x, MEAN, STD, y, 0)), ('fwd', lambda: eng.logits(x, MEAN, STD))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): fn()
    torch.cuda.synchronize(); print(name, 'ms %.3f' % ((time.perf_counter() - t0) / 6 * 1e3))
