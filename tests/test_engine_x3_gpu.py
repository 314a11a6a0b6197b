"""Reference-precision mode of the ResNet-50 engine (ResNet50Engine(precision='bf16x3'), alias 'fp32x').

The reference runs the model in fp32 (RobustART/noise/utils/adv/attack.py:20-23 `f_model`; Attacks/autoattack/
autopgd_base.py:271-289 fp32 logits and gradients) and the north star asks for attack logits within 1e-4 of it.
The bf16 engine is ~3e-3 away; this mode stores every activation / gradient / weight as a hi + lo pair of bf16 values and
forms every contraction as hi.hi + hi.lo + lo.hi on the bf16 MFMA with fp32 accumulation.  Tolerances stated here:
  * logits vs the fp32 torch module AND vs an fp64 evaluation: <= 1e-4 of the logit scale (measured ~1e-5);
  * gradient w.r.t. the input vs an fp64 backward with the engine's ReLU / max-pool decisions: relative L2 error <= 5e-4;
  * kernel-level: the pair GEMM vs fp64 of the same pair operands <= 2e-6 of the output scale (fp32 accumulation only).
"""
import copy
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _split(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()


def _setup(seed=0):
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import ResNet50Engine
    from robustart_amd.model.resnet_torch import randomize_bn_stats
    torch.manual_seed(seed)
    m = randomize_bn_stats(get_model({'type': 'resnet50_official'}), seed).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m.cuda(), ResNet50Engine(m, 'cuda', precision='fp32x')


@pytest.fixture(scope='module')
def setup():
    return _setup()


def test_pair_gemm_kernel_vs_fp64():
    """rart_conv_igemm_bf16 with flag 32 on a 3x3 stride-1 conv: pair in, pair out, bias + residual pair + ReLU + sign bits,
    against fp64 of the same hi/lo operands (only the dropped lo.lo term and fp32 accumulation differ)."""
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    B, Hh, Ww, C, N = 3, 10, 12, 64, 128
    x = torch.randn(B, Hh, Ww, C, generator=g).cuda()
    w = (torch.randn(N, 3, 3, C, generator=g) * 0.05).cuda()             # [n][r][s][c]
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(B, Hh, Ww, N, generator=g).cuda()
    xp, rp = _split(x), _split(res)
    wh = w.to(torch.bfloat16)
    wl = (w - wh.float()).to(torch.bfloat16)
    rows = lambda t: t.reshape(N, 9 * C)                                # noqa: E731
    w3 = torch.cat([rows(wh), rows(wl), rows(wh)], 1).contiguous()
    out = torch.zeros(2, B, Hh, Ww, N, dtype=torch.bfloat16, device='cuda')
    sign = torch.zeros(B, Hh, Ww, N // 8, dtype=torch.uint8, device='cuda')
    d = _lib.ConvDesc()
    d.src, d.wgt, d.dst, d.bias, d.res = xp.data_ptr(), w3.data_ptr(), out.data_ptr(), bias.data_ptr(), rp.data_ptr()
    d.sign_out = sign.data_ptr()
    d.batch, d.grid_h, d.grid_w, d.src_h, d.src_w, d.src_pix_stride = B, Hh, Ww, Hh, Ww, C
    taps = [(r - 1, s - 1) for r in range(3) for s in range(3)]
    lo = xp[1].data_ptr() - xp[0].data_ptr()
    d.k_per_tap, d.n_taps, d.sy, d.sx = C, 27, 1, 1
    for i, (dy, dx) in enumerate(taps * 3):
        d.tap_dy[i], d.tap_dx[i] = dy, dx
        d.tap_src_off[i] = 0 if i < 18 else lo // 2
    d.n_cols, d.dst_h, d.dst_w, d.dst_sy, d.dst_sx, d.dst_pix_stride = N, Hh, Ww, 1, 1, N
    d.flags = 1 | 32
    d.dst_pair_off = (out[1].data_ptr() - out[0].data_ptr()) // 2
    d.res_pair_off = (rp[1].data_ptr() - rp[0].data_ptr()) // 2
    _lib.check(lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr()))
    torch.cuda.synchronize()
    f64 = lambda p: (p[0].double() + p[1].double())                     # noqa: E731
    xv, wv = f64(xp).permute(0, 3, 1, 2), (wh.double() + wl.double()).permute(0, 3, 1, 2)
    want = torch.relu(torch.nn.functional.conv2d(xv, wv, bias.double(), padding=1).permute(0, 2, 3, 1) + f64(rp))
    got = f64(out)
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    print('pair GEMM: scale %.3f, max |err| %.3g (%.2g of scale)' % (scale, err, err / scale))
    # output pair keeps 16 significand bits: |err| <= 2^-17 |v| from the split + the dropped lo.lo products + fp32 sums
    assert err <= 2.5e-5 * scale
    unpacked = torch.stack([(sign >> j) & 1 for j in range(8)], -1).reshape(B, Hh, Ww, N).bool()
    assert torch.equal(unpacked, out[0].float() > 0)
    assert ((got > 0) == unpacked).all()
    # more than 16 taps is refused without the... (27 <= 32 accepted); 33 is an argument error
    d.n_taps = 33
    assert lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr()) != 0


@pytest.mark.parametrize('B,HW', [(3, 96), (4, 224)])
def test_x3_logits_within_1e4_of_fp32_and_fp64(setup, B, HW):
    m, eng = setup
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 3, HW, HW, generator=g).cuda()
    got = eng.logits(x, MEAN, STD)
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    pure = m((x - mean) / std)
    m64 = copy.deepcopy(m).cpu().double()                     # fp64 on the host: independent of MIOpen / rocBLAS
    ref = m64((x.cpu().double() - mean.cpu().double()) / std.cpu().double()).cuda()
    scale = ref.abs().max().item()
    e32 = (got.double() - pure.double()).abs().max().item() / scale
    e64 = (got.double() - ref).abs().max().item() / scale
    t32 = (pure.double() - ref).abs().max().item() / scale
    print('x3 logits B=%d HW=%d: scale %.2f; |x3 - fp32 module| %.2e, |x3 - fp64| %.2e, |fp32 module - fp64| %.2e (of scale)'
          % (B, HW, scale, e32, e64, t32))
    assert e32 <= 1e-4 and e64 <= 1e-4                       # the north star's tolerance
    assert (got.argmax(1) == ref.argmax(1)).all()


def test_x3_u8_entry_and_batch_invariance(setup):
    m, eng = setup
    u8 = torch.randint(0, 256, (5, 64, 96, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
    a = eng.logits_from_u8(u8, MEAN, STD).clone()
    b = eng.logits(u8.permute(0, 3, 1, 2).float() / 255.0, MEAN, STD).clone()
    scale = b.abs().max().item()
    assert (a - b).abs().max().item() <= 2e-5 * scale      # u8/255 by multiply vs divide: one fp32 ulp on the pixel
    one = eng.logits_from_u8(u8[2:3].contiguous(), MEAN, STD)
    assert torch.equal(one[0], a[2])                          # per-element arithmetic does not depend on the batch


def _x3_reference_backward_with_engine_masks(m, eng, acts, dl, std):
    """fp64 backward-to-input of the fp32 network `m` (BatchNorm folded in fp64) using the ENGINE's forward decisions -- its
    1-bit ReLU sign tensors and the max pool's argmax codes -- so that only the arithmetic of the backward chain is
    compared: a ReLU that flips between two forwards that differ by 1e-5 changes the gradient discontinuously, which is a
    property of the network, not of the kernels."""
    import numpy as np
    dt = torch.float64
    F = torch.nn.functional

    def dgrad(c, dz, in_hw):
        return torch.nn.grad.conv2d_input((dz.shape[0], c.cin, in_hw[0], in_hw[1]), c.w_folded.cpu().to(dt), dz, stride=c.stride,
                                          padding=c.pad)

    def sign(t):
        return torch.from_numpy(np.unpackbits(t.cpu().numpy(), axis=-1, bitorder='little')).permute(0, 3, 1, 2).to(dt)
    B = dl.shape[0]
    dpool = dl.detach().cpu().to(dt) @ m.fc.weight.detach().cpu().to(dt)
    xl, xlhw = acts['last']
    dz = sign(acts['last_sign']) * dpool.view(B, -1, 1, 1) / (xlhw[0] * xlhw[1])
    for bi in range(len(eng.blocks) - 1, -1, -1):
        ca, cb, cc, ds = eng.blocks[bi]
        x, xhw, ya, yb, yc, ohw = acts['b%d' % bi]
        mx, ma, mb = acts['b%d_masks' % bi]
        dzb = dgrad(cc, dz, ohw) * sign(mb)
        dza = dgrad(cb, dzb, xhw) * sign(ma)
        if ds is None:
            dz = (dgrad(ca, dza, xhw) + dz) * sign(mx)
        else:
            dz = (dgrad(ca, dza, xhw) + dgrad(ds, dz, xhw)) * sign(mx)
    # max-pool backward + stem ReLU from the engine's argmax codes (code k = window element (k / 3, k % 3); 15 = window maximum <= 0)
    codes = acts['p1_argmax'].cpu().long().permute(0, 3, 1, 2)
    h2, w2 = codes.shape[2], codes.shape[3]
    dz1 = torch.zeros(B, codes.shape[1], 2 * h2, 2 * w2, dtype=dt)
    for k in range(9):
        ky, kx = divmod(k, 3)
        ys, xs = 2 * torch.arange(h2) - 1 + ky, 2 * torch.arange(w2) - 1 + kx
        oky, okx = ys >= 0, xs >= 0
        sel = ((codes == k).to(dt) * dz)[:, :, oky][:, :, :, okx]
        dz1[:, :, ys[oky][:, None], xs[okx][None, :]] += sel
    g = dgrad(eng.stem, dz1, (dz1.shape[2] * 2, dz1.shape[3] * 2))
    return g / torch.tensor(std, dtype=dt).view(1, 3, 1, 1)


@pytest.mark.parametrize('kind', [0, 1, 4])
def test_x3_gradient_vs_fp64(setup, kind):
    """forward_backward (CE and DLR) on the pair engine: (1) against an fp64 backward of the fp32 weights with the engine's own
    ReLU / max-pool decisions (pins the backward arithmetic: rel L2 <= 5e-4), (2) end to end against fp64 autograd of the
    module, with torch's own fp32 autograd measured by the same yardstick beside it."""
    from robustart_amd.noise.adv import logit_loss
    m, eng = setup
    g = torch.Generator().manual_seed(11)
    B = 4
    x = torch.rand(B, 3, 128, 128, generator=g).cuda()
    y = torch.randint(0, 1000, (B,), generator=g).cuda()
    yt = ((y + 1 + torch.randint(0, 998, (B,), generator=g).cuda()) % 1000) if kind == 4 else None     # 4: FAB's z_t - z_y
    logits, loss, grad, pred = eng.forward_backward(x, MEAN, STD, y, kind, yt)
    dl = eng.last_dlogits
    ref = _x3_reference_backward_with_engine_masks(m, eng, eng.last_acts, dl, STD).cuda()
    a, b = grad.double().flatten(1), ref.flatten(1)
    rel = ((a - b).norm(dim=1) / b.norm(dim=1)).cpu()
    cos = ((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).cpu()
    print('x3 gradient (engine masks) kind=%d: rel L2 %s, 1 - cos %s' % (kind, rel.tolist(), (1 - cos).tolist()))
    assert (rel <= 5e-4).all() and (cos >= 1 - 2e-7).all()
    # (2) end to end: fp64 autograd of the module (its own forward, its own ReLU decisions), the engine's loss gradient pushed through
    mean = torch.tensor(MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float64).view(1, 3, 1, 1)
    m64 = copy.deepcopy(m).cpu().double()
    xr = x.cpu().double().requires_grad_(True)
    lg = m64((xr - mean) / std)
    _, dl_at_ref, _ = logit_loss(lg.detach().float().cuda(), y, kind, yt, 1.0)
    assert (dl - dl_at_ref).abs().max().item() <= 2e-2 * dl_at_ref.abs().max().item()
    want, = torch.autograd.grad((lg * dl.double().cpu()).sum(), xr)
    want, lg = want.cuda(), lg.detach().cuda()
    xt = x.clone().requires_grad_(True)
    lt = m((xt - mean.float().cuda()) / std.float().cuda())
    gt, = torch.autograd.grad((lt * dl).sum(), xt)
    b = want.flatten(1)
    rel = ((a - b).norm(dim=1) / b.norm(dim=1)).cpu()
    rel_t = ((gt.double().flatten(1) - b).norm(dim=1) / b.norm(dim=1)).cpu()
    lerr = (logits.double() - lg).abs().max().item() / lg.abs().max().item()
    print('x3 gradient (end to end vs fp64 autograd) kind=%d: rel L2 %s; torch fp32 autograd by the same yardstick %s; logits %.2e'
          % (kind, rel.tolist(), rel_t.tolist(), lerr))
    assert lerr <= 1e-4
    assert (rel <= 5e-2).all()            # ReLU decisions that flip between the two forwards: a sanity bound, (1) is the pin
    assert torch.equal(pred.long(), lg.argmax(1))


def test_x3_through_engine_model_and_pgd(setup):
    """EngineModel(precision=...) carries the mode through the f_model key; a PGD run stays in the eps ball and its final
    logits agree with the fp32 module on the same adversarial examples to 1e-4."""
    from robustart_amd.model.engine import EngineModel
    from robustart_amd.noise import adv
    m, eng = setup
    f = EngineModel(None, takes_normalized=False, engine=eng)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 96, 96, generator=g).cuda()
    y = torch.randint(0, 1000, (4,), generator=g).cuda()
    eps = 4 / 255
    xa = adv.pgd_linf(x, y, f, eps, 3 / 40, 3, seed=2)
    assert (xa - x).abs().max().item() <= eps + 1e-6 and xa.min().item() >= 0 and xa.max().item() <= 1
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    want = m((xa - mean) / std)
    got = f(xa)
    assert (got - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    with pytest.raises(ValueError):
        from robustart_amd.model.engine import ResNet50Engine
        ResNet50Engine(m, 'cuda', precision='fp16')


@pytest.mark.parametrize('N,tile_n,tile_m', [(128, 0, 0), (128, 64, 256), (64, 0, 256), (256, 256, 256), (152, 0, 0), (256, 256, 128), (128, 128, 128),
                                             (64, 64, 128)])
def test_pair_conv_kernel_vs_fp64(N, tile_n, tile_m):
    """rart_gemm_pair_bf16 in conv mode (round 4: the four operand planes of a K step staged once, three MFMAs per fragment pair) on a
    3x3 stride-1 conv: pair in, pair out, bias + residual pair + ReLU + sign bits, against fp64 of the same hi / lo operands; every
    column tile (64 / 128 / 256, forced and automatic) and an N that is not a multiple of the tile."""
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    B, Hh, Ww, C = 3, 10, 12, 64
    x = torch.randn(B, Hh, Ww, C, generator=g).cuda()
    w = (torch.randn(N, 3, 3, C, generator=g) * 0.05).cuda()             # [n][r][s][c]
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(B, Hh, Ww, N, generator=g).cuda()
    xp, rp = _split(x), _split(res)
    wh = w.to(torch.bfloat16)
    wl = (w - wh.float()).to(torch.bfloat16)
    rows = lambda t: t.reshape(N, 9 * C)                                # noqa: E731
    w3 = torch.cat([rows(wh), rows(wl), rows(wh)], 1).contiguous()
    out = torch.zeros(2, B, Hh, Ww, N, dtype=torch.bfloat16, device='cuda')
    sign = torch.zeros(B, Hh, Ww, N // 8, dtype=torch.uint8, device='cuda')
    d = _lib.GemmPairDesc()
    d.a_hi, d.a_lo, d.w_hi, d.w_lo = xp[0].data_ptr(), xp[1].data_ptr(), w3.data_ptr(), w3.data_ptr() + 2 * 9 * C
    d.bias, d.res_hi, d.res_lo, d.dst_hi, d.dst_lo = bias.data_ptr(), rp[0].data_ptr(), rp[1].data_ptr(), out[0].data_ptr(), out[1].data_ptr()
    d.sign_out = sign.data_ptr()
    d.N, d.lda, d.ldw, d.ldc, d.w_rows, d.flags, d.tile_n, d.tile_m = N, C, 27 * C, N, N, 1, tile_n, tile_m
    d.conv, d.batch, d.grid_h, d.grid_w, d.src_h, d.src_w, d.sy, d.sx, d.k_per_tap, d.n_taps = 1, B, Hh, Ww, Hh, Ww, 1, 1, C, 9
    for i, (dy, dx) in enumerate([(r - 1, s - 1) for r in range(3) for s in range(3)]):
        d.tap_dy[i], d.tap_dx[i] = dy, dx
    d.dst_h, d.dst_w, d.dst_sy, d.dst_sx = Hh, Ww, 1, 1
    _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
    torch.cuda.synchronize()
    f64 = lambda p: (p[0].double() + p[1].double())                     # noqa: E731
    xv, wv = f64(xp).permute(0, 3, 1, 2), (wh.double() + wl.double()).permute(0, 3, 1, 2)
    want = torch.relu(torch.nn.functional.conv2d(xv, wv, bias.double(), padding=1).permute(0, 2, 3, 1) + f64(rp))
    got = f64(out)
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    print('pair conv kernel N=%d tile %d x %d: scale %.3f, max |err| %.3g (%.2g of scale)' % (N, tile_m, tile_n, scale, err, err / scale))
    assert err <= 2.5e-5 * scale
    unpacked = torch.stack([(sign >> j) & 1 for j in range(8)], -1).reshape(B, Hh, Ww, N).bool()
    assert torch.equal(unpacked, out[0].float() > 0)
    # 17 taps / a k_per_tap that is not a power of two are argument errors
    d.n_taps = 17
    assert lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()) != 0
    d.n_taps, d.k_per_tap = 9, 96
    assert lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()) != 0


def test_pair_conv_kernel_stride2_backward_parity_classes_with_mask_bits():
    """The backward-to-input of a 3x3 / 2 convolution as four input-parity classes (sub-grid + strided destination), 1-bit ReLU mask of
    the destination, fp64 reference by conv_transpose -- the launch shape `_conv_bwd` uses for the stride-2 blocks."""
    from robustart_amd import _lib
    from robustart_amd.model.engine import _Conv
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    B, Ho, Wo, Cin, Cout = 2, 7, 9, 64, 128
    conv = torch.nn.Conv2d(Cin, Cout, 3, 2, 1, bias=False)
    conv.weight.data.copy_(torch.randn(conv.weight.shape, generator=g) * 0.05)
    c = _Conv(conv, None, 'cuda', split=True)
    dz = _split(torch.randn(B, Ho, Wo, Cout, generator=g).cuda())
    keep = torch.rand(B, 2 * Ho, 2 * Wo, Cin, generator=g) > 0.4
    bits = torch.zeros(B, 2 * Ho, 2 * Wo, Cin // 8, dtype=torch.uint8)
    for j in range(8):
        bits |= (keep.view(B, 2 * Ho, 2 * Wo, Cin // 8, 8)[..., j].to(torch.uint8) << j)
    bits = bits.cuda()
    dx = torch.full((2, B, 2 * Ho, 2 * Wo, Cin), float('nan'), dtype=torch.bfloat16, device='cuda')
    for (ph, pw), taps, w in c.bwd:
        d = _lib.GemmPairDesc()
        kt = Cout * len(taps)
        d.a_hi, d.a_lo, d.w_hi, d.w_lo = dz[0].data_ptr(), dz[1].data_ptr(), w.data_ptr(), w.data_ptr() + 2 * kt
        d.dst_hi, d.dst_lo, d.mask_bits = dx[0].data_ptr(), dx[1].data_ptr(), bits.data_ptr()
        d.N, d.lda, d.ldw, d.ldc, d.w_rows = Cin, Cout, 3 * kt, Cin, w.shape[0]
        d.conv, d.batch, d.grid_h, d.grid_w, d.src_h, d.src_w, d.sy, d.sx, d.k_per_tap, d.n_taps = 1, B, Ho, Wo, Ho, Wo, 1, 1, Cout, len(taps)
        for i, (dy, dx_) in enumerate(taps):
            d.tap_dy[i], d.tap_dx[i] = dy, dx_
        d.dst_h, d.dst_w, d.dst_sy, d.dst_sx, d.dst_oy, d.dst_ox = 2 * Ho, 2 * Wo, 2, 2, ph, pw
        _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
    torch.cuda.synchronize()
    f64 = lambda p: (p[0].double() + p[1].double())                     # noqa: E731
    wv = c.w_folded.double()
    wv = (wv.to(torch.bfloat16).double() + (wv - wv.to(torch.bfloat16).double()).to(torch.bfloat16).double()).cuda()    # hi + lo of the table
    want = torch.nn.grad.conv2d_input((B, Cin, 2 * Ho, 2 * Wo), wv, f64(dz).permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
    want = want * keep.cuda()
    got = f64(dx)
    assert torch.isfinite(got).all()                                     # every destination pixel belongs to exactly one parity class
    assert (got - want).abs().max().item() <= 2.5e-5 * want.abs().max().item()


def test_x3_engine_new_kernel_matches_the_round3_igemm_pair_path(setup):
    """The whole reference-precision forward + backward on rart_gemm_pair_bf16 against the same engine on round 3's path (the three
    products as 3 x the taps of rart_conv_igemm_bf16): same operands, fp32 accumulation in another order."""
    m, eng = setup
    g = torch.Generator().manual_seed(8)
    x = torch.rand(3, 3, 96, 128, generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    assert eng.pair_gemm_kernel
    la, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    la, ga = la.clone(), ga.clone()
    eng.pair_gemm_kernel = False
    try:
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    finally:
        eng.pair_gemm_kernel = True
    assert (la - lb).abs().max().item() <= 2e-5 * lb.abs().max().item()
    a, b = ga.double().flatten(1), gb.double().flatten(1)
    rel = ((a - b).norm(dim=1) / b.norm(dim=1)).max().item()
    print('x3 engine: new pair kernel vs igemm pair path: logits %.2e, gradient rel L2 %.2e'
          % ((la - lb).abs().max().item() / lb.abs().max().item(), rel))
    # ReLU decisions of near-zero pre-activations flip between two forwards that differ by 1e-5 (random-init network): the gradient
    # moves discontinuously with them -- the fp64 tests with the engine's own masks are the pin, this is a sanity bound
    cos = ((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).min().item()
    assert rel <= 0.15 and cos >= 0.98


@pytest.mark.parametrize('HW', [(96, 128), (224, 224), (32, 64)])
def test_x3_fused_stem_backward_matches_the_three_launch_chain(setup, HW):
    """rart_engine_stem_bwd_fused_pair (max-pool backward + ReLU mask + transposed 7x7/2 conv on pairs, one kernel) against the chain it
    replaces (rart_engine_maxpool_bwd_pair -> patches GEMM with fp32 output -> rart_engine_stem_col2im_f32) inside the same
    forward + backward: identical forward, identical decisions, so only the summation order of the last contraction differs.  Image
    sizes with full, partial and single 16 x 16 tiles of the stem grid."""
    m, eng = setup
    g = torch.Generator().manual_seed(21)
    x = torch.rand(3, 3, HW[0], HW[1], generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    assert eng.fused_stem_bwd
    la, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    la, ga = la.clone(), ga.clone()
    eng.fused_stem_bwd = False
    try:
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    finally:
        eng.fused_stem_bwd = True
    assert torch.equal(la, lb)
    assert torch.isfinite(ga).all()
    a, b = ga.double().flatten(1), gb.double().flatten(1)
    rel = ((a - b).norm(dim=1) / b.norm(dim=1)).max().item()
    mx = (a - b).abs().max().item() / b.abs().max().item()
    print('x3 fused stem backward vs chain %s: rel L2 %.2e, max %.2e of scale' % (HW, rel, mx))
    assert rel <= 2e-6 and mx <= 2e-6


def _bits(t, channels):
    """[P][channels / 8] uint8 -> [P][channels] of 0 / 1 (bit c % 8 of byte c / 8)"""
    sh = torch.arange(8, device=t.device, dtype=torch.uint8)
    return ((t.unsqueeze(-1) >> sh) & 1).reshape(t.shape[0], channels)


@pytest.mark.parametrize('C,backward,nxt', [(64, False, False), (64, True, False), (128, False, False), (128, True, False),
                                            (64, False, True), (64, True, True), (128, False, True), (128, True, True)])
def test_conv_tail_pair_kernel_vs_fp64(C, backward, nxt):
    """rart_conv3x3_tail_pair (3x3 + point-wise step + 1x1 expansion + skip + point-wise step on pairs, one launch; the intermediate
    stays in registers) against fp64 of the same pair operands with the intermediate rounded to a pair where the kernel rounds it.
    Forward mode: biases, ReLU, both sign tensors out.  Backward mode: both 1-bit masks in, no bias.  M = 2 x 9 x 11 positions is not a
    multiple of the 128-position tile; taps reach outside the image on every side.  nxt: the neighbouring block's 1x1 reduction
    (4C -> C) of the output tile in the same launch (forward: bias + ReLU + sign tensor; backward: mask)."""
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(C + backward)
    B, H, W, N = 2, 9, 11, 4 * C
    P = B * H * W
    x = _split(torch.randn(B, H, W, C, generator=g).cuda())
    w2 = (torch.randn(C, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()             # [n][r][s][c]
    w3 = (torch.randn(N, C, generator=g) * (1.0 / C) ** 0.5).cuda()
    res = _split(torch.randn(B, H, W, N, generator=g).cuda())
    b2, b3 = torch.randn(C, generator=g).cuda() * 0.3, torch.randn(N, generator=g).cuda() * 0.3
    w2p, w3p = _split(w2.reshape(C, 9 * C)), _split(w3)
    tab = torch.cat([w2p[0], w2p[1], w2p[0]], 1).contiguous()
    tail = torch.stack([w3p[pl].reshape(N // 32, 32, C // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1) for pl in range(2)]).contiguous()
    dst = torch.full((2, B, H, W, N), float('nan'), dtype=torch.bfloat16, device='cuda')
    d = _lib.ConvTailDesc()
    d.a_hi, d.a_lo, d.w_hi, d.w_lo = x[0].data_ptr(), x[1].data_ptr(), tab.data_ptr(), tab.data_ptr() + 2 * 9 * C
    d.t_hi, d.t_lo = tail[0].data_ptr(), tail[1].data_ptr()
    d.res_hi, d.res_lo, d.dst_hi, d.dst_lo = res[0].data_ptr(), res[1].data_ptr(), dst[0].data_ptr(), dst[1].data_ptr()
    d.batch, d.h, d.w, d.c_mid, d.ldw = B, H, W, C, 3 * 9 * C
    for i in range(9):
        d.tap_dy[i], d.tap_dx[i] = i // 3 - 1, i % 3 - 1
    if backward:
        mm = torch.randint(0, 256, (P, C // 8), generator=g, dtype=torch.uint8).cuda()
        mo = torch.randint(0, 256, (P, N // 8), generator=g, dtype=torch.uint8).cuda()
        d.mask_mid, d.mask_out = mm.data_ptr(), mo.data_ptr()
    else:
        sm = torch.full((P, C // 8), 0xAA, dtype=torch.uint8, device='cuda')
        so = torch.full((P, N // 8), 0xAA, dtype=torch.uint8, device='cuda')
        d.bias_mid, d.bias_out, d.sign_mid, d.sign_out = b2.data_ptr(), b3.data_ptr(), sm.data_ptr(), so.data_ptr()
        d.relu_mid = d.relu_out = 1
    if nxt:
        wn = (torch.randn(C, N, generator=g) * (1.0 / N) ** 0.5).cuda()
        wnp = _split(wn)
        ntab = torch.stack([wnp[pl].reshape(C // 32, 32, N // 64, 4, 2, 8).permute(2, 0, 3, 4, 1, 5).contiguous().view(-1)
                            for pl in range(2)]).contiguous()
        dn = torch.full((2, P, C), float('nan'), dtype=torch.bfloat16, device='cuda')
        d.n_hi, d.n_lo, d.dstn_hi, d.dstn_lo = ntab[0].data_ptr(), ntab[1].data_ptr(), dn[0].data_ptr(), dn[1].data_ptr()
        if backward:
            mn = torch.randint(0, 256, (P, C // 8), generator=g, dtype=torch.uint8).cuda()
            d.mask_next = mn.data_ptr()
        else:
            bn = torch.randn(C, generator=g).cuda() * 0.3
            sn = torch.full((P, C // 8), 0xAA, dtype=torch.uint8, device='cuda')
            d.bias_next, d.sign_next, d.relu_next = bn.data_ptr(), sn.data_ptr(), 1
    _lib.check(lib.rart_conv3x3_tail_pair(ctypes.byref(d), _lib.stream_ptr()))
    torch.cuda.synchronize()
    x64 = (x[0].double() + x[1].double()).permute(0, 3, 1, 2)
    w64 = (w2p[0].double() + w2p[1].double()).reshape(C, 3, 3, C).permute(0, 3, 1, 2)
    mid = torch.nn.functional.conv2d(x64, w64, padding=1).permute(0, 2, 3, 1).reshape(P, C)
    if backward:
        mid = mid * _bits(mm, C).double()
    else:
        mid = (mid + b2.double()).clamp_min(0)
    mp = _split(mid.float())
    out = (mp[0].double() + mp[1].double()) @ (w3p[0].double() + w3p[1].double()).t() + (res[0].double() + res[1].double()).reshape(P, N)
    if backward:
        out = out * _bits(mo, N).double()
    else:
        out = (out + b3.double()).clamp_min(0)
    got = (dst[0].double() + dst[1].double()).reshape(P, N)
    assert torch.isfinite(got).all()
    err = (got - out).abs().max().item() / out.abs().max().item()
    print('conv tail pair C=%d %s: max err %.2e of scale' % (C, 'backward' if backward else 'forward', err))
    assert err <= 2e-5
    if not backward:
        assert torch.equal(_bits(so, N), (dst[0].reshape(P, N) > 0).to(torch.uint8))       # the sign tensor of the kernel's own output
        clear = mid.abs() > 1e-3
        assert torch.equal(_bits(sm, C)[clear], (mid > 0).to(torch.uint8)[clear])
    if nxt:
        # the reduction multiplies the pair the kernel WROTE
        nref = got @ (wnp[0].double() + wnp[1].double()).t()
        nref = nref * _bits(mn, C).double() if backward else (nref + bn.double()).clamp_min(0)
        ngot = dn[0].double() + dn[1].double()
        assert torch.isfinite(ngot).all()
        nerr = (ngot - nref).abs().max().item() / nref.abs().max().item()
        print('   neighbour reduction: max err %.2e of scale' % nerr)
        assert nerr <= 1e-5
        if not backward:
            assert torch.equal(_bits(sn, C), (dn[0] > 0).to(torch.uint8))


@pytest.mark.parametrize('HW,u8', [((96, 128), False), ((224, 224), True), ((32, 64), False)])
def test_x3_fused_stem_forward_is_bit_identical_to_the_three_launch_chain(setup, HW, u8):
    """rart_engine_stem_fwd_fused_pair (normalise + split + 7x7/2 conv with three products per K step + bias + ReLU + split + max pool of
    the pair values, one persistent kernel) against rart_engine_prep_input -> rart_gemm_pair_bf16 -> rart_engine_maxpool_pair: the pooled
    pair, the argmax codes and the sign bits, bit for bit (same products in the same order, same rounding points); fp32 and uint8 input."""
    m, eng = setup
    g = torch.Generator().manual_seed(41)
    if u8:
        x = torch.randint(0, 256, (3, HW[0], HW[1], 3), generator=g, dtype=torch.uint8).cuda()
        run = lambda: eng._forward(x, True, MEAN, STD, True)          # noqa: E731
    else:
        x = torch.rand(3, 3, HW[0], HW[1], generator=g).cuda()
        run = lambda: eng._forward(x, False, MEAN, STD, True)         # noqa: E731
    assert eng.fused_stem_fwd
    la, _ = run()
    a = [eng._buf[k].clone() for k in ('x3_p1', 'p1_argmax', 'p1_sign')]
    la = la.clone()
    eng.fused_stem_fwd = False
    try:
        lb, _ = run()
    finally:
        eng.fused_stem_fwd = True
    b = [eng._buf[k] for k in ('x3_p1', 'p1_argmax', 'p1_sign')]
    for name, u, v in zip(('p1 pair', 'argmax codes', 'sign bits'), a, b):
        assert torch.equal(u, v), name
    assert torch.equal(la, lb)


def test_x3_fused_tail_matches_the_two_launch_path(setup):
    """The reference-precision forward + backward with conv2 + conv3 (forward) / conv2^T + conv1^T (backward) of layer1 / layer2 as one
    launch each against the same engine on two launches of rart_gemm_pair_bf16: the same products in the same order with the
    intermediate rounded to a pair at the same place."""
    m, eng = setup
    g = torch.Generator().manual_seed(31)
    x = torch.rand(3, 3, 96, 128, generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    assert eng.fused_tail_pair and getattr(eng.blocks[1][2], 'tail_fwd', None) is not None and getattr(eng.blocks[1][0], 'tail_bwd', None) is not None
    # (round 6: the engine's default keeps layer2 on two launches -- faster at B = 256 -- so the test switches its tail kernel on)
    default_channels = eng.fused_tail_channels
    eng.fused_tail_channels = (64, 128)
    request_restore = lambda: setattr(eng, 'fused_tail_channels', default_channels)
    try:
        _x3_fused_tail_body(eng, x, y)
    finally:
        request_restore()


def _x3_fused_tail_body(eng, x, y):
    la, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    la, ga = la.clone(), ga.clone()
    signs_a = {k: v.clone() for k, v in eng._buf.items() if k.endswith('_sign')}
    assert eng.fused_next_pair and getattr(eng.blocks[1][0], 'next_fwd', None) is not None and getattr(eng.blocks[0][2], 'next_bwd', None) is not None
    eng.fused_next_pair = False          # the tail launches without the neighbour's reduction
    try:
        lc, _, gc, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        lc, gc = lc.clone(), gc.clone()
    finally:
        eng.fused_next_pair = True
    print('x3 tail with vs without the neighbour reduction: logits bit-equal %s, gradient bit-equal %s' % (torch.equal(la, lc), torch.equal(ga, gc)))
    assert (la - lc).abs().max().item() <= 2e-5 * lc.abs().max().item()
    eng.fused_tail_pair = False
    try:
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    finally:
        eng.fused_tail_pair = True
    signs_b = {k: v for k, v in eng._buf.items() if k.endswith('_sign')}
    le = (la - lb).abs().max().item() / lb.abs().max().item()
    a, b = ga.double().flatten(1), gb.double().flatten(1)
    rel = ((a - b).norm(dim=1) / b.norm(dim=1)).max().item()
    same = sum(int(torch.equal(signs_a[k], signs_b[k])) for k in signs_a)
    print('x3 fused tail vs two launches: logits %.2e of scale (bit-equal: %s), gradient rel L2 %.2e, sign tensors equal %d / %d'
          % (le, torch.equal(la, lb), rel, same, len(signs_a)))
    assert le <= 2e-5
    cos = ((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).min().item()
    assert rel <= 0.15 and cos >= 0.98          # ReLU decisions of near-zero pre-activations may flip (see the igemm cross-check above)


def test_x3_engine_b256_matches_small_batches_bit_for_bit(setup):
    """The reference-precision ResNet-50 engine at the benchmark's B = 256: every 32nd image of a forward / forward + backward equals the
    same image in a batch of 2 bit for bit (every tile of k_gemm_pair sums a row's K slices in the same order whatever the batch), and
    the logits of those rows are within 1e-4 of the fp32 module -- the small-batch parity statements hold at the full size."""
    m, eng = setup
    g = torch.Generator().manual_seed(6)
    x = torch.rand(256, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (256,), generator=g).cuda()
    big = eng.logits(x, MEAN, STD).clone()
    lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    lb, gb = lb.clone(), gb.clone()
    assert torch.equal(big, lb)
    for i in range(0, 256, 32):
        xs, ys = x[i:i + 2].contiguous(), y[i:i + 2].contiguous()
        small = eng.logits(xs, MEAN, STD)
        assert torch.equal(small[0], big[i]) and torch.equal(small[1], big[i + 1]), i
        ls, _, gs, _ = eng.forward_backward(xs, MEAN, STD, ys, 0)
        assert torch.equal(ls[0], lb[i]) and torch.equal(gs[0], gb[i]) and torch.equal(gs[1], gb[i + 1]), i
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    rows = torch.arange(0, 256, 32, device='cuda')
    ref = m((x[rows] - mean) / std)
    assert (big[rows] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def _pair_schedule_case(lib, _lib, conv, M, K, N, tile_n, batched, seed):
    """One rart_gemm_pair_bf16 problem with every epilogue operand (bias, residual pair, ReLU; conv: sign bits), 256-row tiles forced."""
    g = torch.Generator().manual_seed(seed)
    nz = 3 if batched else 1
    d = _lib.GemmPairDesc()
    keep = []
    if conv:
        B, Hh, Ww, C = conv
        assert K == 9 * C and M == B * Hh * Ww
        xp = _split(torch.randn(B, Hh, Ww, C, generator=g).cuda())
        w = (torch.randn(N, 9 * C, generator=g) * 0.05).cuda()
    else:
        xp = _split(torch.randn(nz, M, K, generator=g).cuda())
        w = (torch.randn(nz, N, K, generator=g) * 0.05).cuda()
    wh = w.to(torch.bfloat16)
    wl = (w - wh.float()).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    rp = _split(torch.randn(nz, M, N, generator=g).cuda())
    out = torch.full((2, nz, M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
    sign = torch.zeros(M, N // 8, dtype=torch.uint8, device='cuda')
    keep += [xp, wh, wl, bias, rp, out, sign]
    d.a_hi, d.a_lo, d.w_hi, d.w_lo = xp[0].data_ptr(), xp[1].data_ptr(), wh.data_ptr(), wl.data_ptr()
    d.bias, d.res_hi, d.res_lo, d.dst_hi, d.dst_lo = bias.data_ptr(), rp[0].data_ptr(), rp[1].data_ptr(), out[0].data_ptr(), out[1].data_ptr()
    d.N, d.ldw, d.ldc, d.w_rows, d.flags, d.tile_n, d.tile_m = N, K, N, N, 1, tile_n, 256
    if conv:
        d.sign_out = sign.data_ptr()
        d.lda = C
        d.conv, d.batch, d.grid_h, d.grid_w, d.src_h, d.src_w, d.sy, d.sx, d.k_per_tap, d.n_taps = 1, B, Hh, Ww, Hh, Ww, 1, 1, C, 9
        for i, (dy, dx) in enumerate([(r - 1, s - 1) for r in range(3) for s in range(3)]):
            d.tap_dy[i], d.tap_dx[i] = dy, dx
        d.dst_h, d.dst_w, d.dst_sy, d.dst_sx = Hh, Ww, 1, 1
    else:
        d.M, d.K, d.lda = M, K, K
        if batched:
            d.n_batched, d.z_inner = nz, 1
            d.a_z_outer, d.w_z_outer, d.c_z_outer = M * K, N * K, M * N
    return d, out, sign, keep


@pytest.mark.parametrize('conv,M,K,N,tile_n,batched', [
    (None, 700, 32, 256, 256, False),            # one K step: prologue only
    (None, 513, 64, 128, 128, False),            # two K steps: the refill of group 0's first M0 and nothing else
    (None, 1000, 96, 384, 256, False),           # three K steps, a half-empty column tile
    (None, 4103, 1024, 256, 256, False),         # 32 K steps, >= 16 row tiles (the XCD remap), ragged last tile
    (None, 777, 160, 256, 128, True),            # batched over blockIdx.y, five K steps, 128-column tiles
    ((5, 20, 20, 64), 2000, 576, 256, 256, False),      # 3x3 gather, 18 K steps, zero padding through the zero page
    ((3, 12, 10, 32), 360, 288, 128, 128, False),       # 32-channel taps (one K step per tap)
    ((4, 33, 32, 128), 4224, 1152, 512, 256, False),    # 36 K steps, two column tiles, 17 row tiles
])
def test_pair_gemm_pingpong_schedule_is_bit_identical(conv, M, K, N, tile_n, batched):
    """Round 6: the ping-pong schedule of the 256-row tiles (csrc/gemm_pair_pp.hip: the two halves of the workgroup alternate memory and
    matrix phases, counted vmcnt, raw barriers) issues the same products in the same order as round 4's two-stage loop: the outputs must be
    equal BIT FOR BIT -- and stay so over repeated launches beside a bandwidth hog on another stream (a missing wait shows as a rare wrong
    tile, not as a wrong kernel)."""
    from robustart_amd import _lib
    lib = _lib.load()
    d, out, sign, keep = _pair_schedule_case(lib, _lib, conv, M, K, N, tile_n, batched, seed=11)
    old = lib.rart_gemm_pair_get_schedule()
    try:
        _lib.check(lib.rart_gemm_pair_set_schedule(0))
        _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
        torch.cuda.synchronize()
        want, want_sign = out.view(torch.int16).clone(), sign.clone()
        assert torch.isfinite(out.float()).all()
        _lib.check(lib.rart_gemm_pair_set_schedule(1))
        hog_stream = torch.cuda.Stream()
        hog = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
        for rep in range(6):
            out.fill_(float('nan'))
            sign.zero_()
            if rep >= 2:
                with torch.cuda.stream(hog_stream):
                    for _ in range(4):
                        hog.add_(1.0)
            _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int16), want), 'ping-pong schedule differs from the two-stage loop (repetition %d)' % rep
            assert torch.equal(sign, want_sign)
    finally:
        lib.rart_gemm_pair_set_schedule(old)
    assert lib.rart_gemm_pair_set_schedule(4) != 0


@pytest.mark.parametrize('conv,M,K,N,tile_n', [
    ((16, 64, 64, 64), 65536, 576, 512, 256),    # 3x3 gather, 18 K steps, 512 tiles: two per workgroup, zero padding in every tile
    ((9, 60, 60, 32), 32400, 288, 640, 128),     # one K step per tap, 127 x 5 tiles of 256 x 128, ragged last row tile and column tile
    (None, 70000, 96, 512, 128),                 # three K steps, 274 x 4 tiles of 256 x 128
    (None, 66000, 32, 256, 256),                 # ONE K step per tile: the walk is prologue after prologue
    (None, 66100, 64, 296, 256),                 # two K steps, a column tile with 40 valid columns, holes in the XCD enumeration
])
def test_pair_gemm_walked_tiles_are_bit_identical(conv, M, K, N, tile_n):
    """Schedule 3 on forced 256-row tiles with taps / short K loops (the shapes the persistent cases above do not reach): a workgroup's second and
    third tile start from the stage the PREVIOUS tile's epilogue ran beside."""
    from robustart_amd import _lib
    lib = _lib.load()
    d, out, sign, keep = _pair_schedule_case(lib, _lib, conv, M, K, N, tile_n, False, seed=31)
    old = lib.rart_gemm_pair_get_schedule()
    try:
        _lib.check(lib.rart_gemm_pair_set_schedule(0))
        _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
        torch.cuda.synchronize()
        want, want_sign = out.view(torch.int16).clone(), sign.clone()
        assert torch.isfinite(out.float()).all()
        _lib.check(lib.rart_gemm_pair_set_schedule(3))
        hog_stream = torch.cuda.Stream()
        hog = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
        for rep in range(5):
            out.fill_(float('nan'))
            sign.zero_()
            if rep >= 2:
                with torch.cuda.stream(hog_stream):
                    for _ in range(4):
                        hog.add_(1.0)
            _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
            torch.cuda.synchronize()
            bad = (out.view(torch.int16) != want).any(0).any(0)
            assert not bad.any(), 'walked tiles differ from the two-stage loop (repetition %d): %d elements, first rows %s' % (
                rep, int(bad.sum()), bad.any(1).nonzero().flatten()[:8].tolist())
            assert torch.equal(sign, want_sign)
    finally:
        lib.rart_gemm_pair_set_schedule(old)


@pytest.mark.parametrize('conv,M,K,N,tile_n', [
    ((4, 33, 32, 128), 4224, 1152, 512, 256),    # 3x3, 36 K steps, 19 tiles of 224 rows (the XCD enumeration), two column tiles
    ((3, 12, 10, 32), 360, 288, 128, 128),       # two tiles, the second with 136 rows: its blocks past M AND its block past 224
    (None, 224 * 9, 160, 256, 256),              # M a multiple of 224: the last tile is full
    (None, 4103, 1024, 384, 128),                # 32 K steps, ragged rows and columns, 256 x 128 tiles
    (None, 200, 64, 256, 256),                   # one tile shorter than 224 rows
])
def test_pair_gemm_224_row_tiles_are_bit_identical(conv, M, K, N, tile_n):
    """Round 6: the ping-pong kernel with tiles that step 224 rows (tile_m = 224: the last 32-row block of a tile is neither multiplied nor
    stored; chosen automatically where it saves a pass) against the two-stage loop on 256-row tiles: equal bit for bit, every row written
    exactly once."""
    from robustart_amd import _lib
    lib = _lib.load()
    d, out, sign, keep = _pair_schedule_case(lib, _lib, conv, M, K, N, tile_n, False, seed=51)
    old = lib.rart_gemm_pair_get_schedule()
    try:
        _lib.check(lib.rart_gemm_pair_set_schedule(0))
        _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
        torch.cuda.synchronize()
        want, want_sign = out.view(torch.int16).clone(), sign.clone()
        _lib.check(lib.rart_gemm_pair_set_schedule(1))
        d.tile_m = 224
        for rep in range(3):
            out.fill_(float('nan'))
            sign.zero_()
            _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
            torch.cuda.synchronize()
            bad = (out.view(torch.int16) != want).any(0).any(0)
            assert not bad.any(), '224-row tiles differ (repetition %d): %d elements, first rows %s' % (rep, int(bad.sum()), bad.any(1).nonzero().flatten()[:8].tolist())
            assert torch.equal(sign, want_sign)
    finally:
        lib.rart_gemm_pair_set_schedule(old)


@pytest.mark.parametrize('M,K,N', [(25 * 256 + 70, 64, 1024), (3300, 256, 64), (17 * 128 + 5, 96, 192)])
def test_pair_gemm_two_stage_short_row_tiles_are_bit_identical(M, K, N, monkeypatch):
    """The lab level RART_PAIR_ROWS224=2 (tiles that step TM - 32 rows in the two-stage k_gemm_pair where the pass arithmetic says so; measured
    slower on ResNet-50, off by default) must still produce the same bits: every row written exactly once, the block a tile leaves out
    computed by the next tile."""
    from robustart_amd import _lib
    lib = _lib.load()
    d, out, sign, keep = _pair_schedule_case(lib, _lib, None, M, K, N, 0, False, seed=61)
    d.tile_m = 0
    old = lib.rart_gemm_pair_get_schedule()
    try:
        res = []
        for sched, level in ((0, '0'), (1, '2'), (1, '2')):
            monkeypatch.setenv('RART_PAIR_ROWS224', level)
            _lib.check(lib.rart_gemm_pair_set_schedule(sched))
            out.fill_(float('nan'))
            _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all()
            res.append(out.view(torch.int16).clone())
        assert torch.equal(res[1], res[0]) and torch.equal(res[2], res[0])
    finally:
        lib.rart_gemm_pair_set_schedule(old)


@pytest.mark.parametrize('M,K,N', [
    (197 * 256 - 37, 96, 768),       # ViT-B/16's geometry: 197 x 3 tiles = 75 per XCD -> 168 row tiles on 256 x 256, 29 (ragged) on 256 x 128
    (100 * 256, 64, 1000),           # 100 x 4 tiles, a ragged column tile in both launches (232 / 104 valid columns)
])
def test_pair_gemm_remainder_split_is_bit_identical(M, K, N, monkeypatch):
    """Round 6: a plain product that fills the CUs a whole number of times and then less than half of them once more is issued as two launches
    (whole passes on 256 x 256 tiles, the remaining rows on 256 x 128 tiles: csrc/gemm_pair.hip).  Against the unsplit launch and against the
    two-stage loop: equal bit for bit."""
    from robustart_amd import _lib
    lib = _lib.load()
    d, out, sign, keep = _pair_schedule_case(lib, _lib, None, M, K, N, 0, False, seed=41)
    d.tile_m = 0
    old = lib.rart_gemm_pair_get_schedule()
    try:
        res = []
        for sched, split in ((0, '0'), (1, '0'), (1, '1'), (1, '1')):
            monkeypatch.setenv('RART_PAIR_SPLIT', split)
            _lib.check(lib.rart_gemm_pair_set_schedule(sched))
            out.fill_(float('nan'))
            _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all()
            res.append(out.view(torch.int16).clone())
        for r in res[1:]:
            assert torch.equal(r, res[0])
    finally:
        lib.rart_gemm_pair_set_schedule(old)


def _pair_ps_case(_lib, conv, M, K, N, flags_relu, with_res, with_mask, seed):
    """A one-tap problem large enough for the persistent kernel (>= 2 x 256 tiles of 256 x 128), automatic tiles."""
    g = torch.Generator().manual_seed(seed)
    d = _lib.GemmPairDesc()
    if conv:
        B, Hh, Ww, stride = conv
        C = K
        xp = _split(torch.randn(B, Hh * stride, Ww * stride, C, generator=g).cuda())
        assert M == B * Hh * Ww
    else:
        xp = _split(torch.randn(M, K, generator=g).cuda())
    w = (torch.randn(N, K, generator=g) * 0.05).cuda()
    wh = w.to(torch.bfloat16)
    wl = (w - wh.float()).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    rp = _split(torch.randn(M, N, generator=g).cuda())
    out = torch.full((2, M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
    sign = torch.zeros(M, N // 8, dtype=torch.uint8, device='cuda')
    bits = (torch.rand(M, N // 8, generator=g) * 256).to(torch.uint8).cuda()
    d.a_hi, d.a_lo, d.w_hi, d.w_lo = xp[0].data_ptr(), xp[1].data_ptr(), wh.data_ptr(), wl.data_ptr()
    d.bias, d.dst_hi, d.dst_lo = bias.data_ptr(), out[0].data_ptr(), out[1].data_ptr()
    if with_res:
        d.res_hi, d.res_lo = rp[0].data_ptr(), rp[1].data_ptr()
    d.N, d.ldw, d.ldc, d.w_rows, d.flags = N, K, N, N, (1 if flags_relu else 0)
    if conv:
        d.lda = C
        d.conv, d.batch, d.grid_h, d.grid_w, d.src_h, d.src_w, d.sy, d.sx, d.k_per_tap, d.n_taps = 1, B, Hh, Ww, Hh * stride, Ww * stride, stride, stride, C, 1
        d.tap_dy[0], d.tap_dx[0] = 0, 0
        d.dst_h, d.dst_w, d.dst_sy, d.dst_sx = Hh, Ww, 1, 1
        if with_mask:
            d.mask_bits = bits.data_ptr()
        else:
            d.sign_out = sign.data_ptr()
    else:
        d.M, d.K, d.lda = M, K, K
    return d, out, sign, [xp, wh, wl, bias, rp, bits]


@pytest.mark.parametrize('conv,M,K,N,relu,res,mask', [
    (None, 10317, 192, 1792, False, True, False),          # six K steps: exactly the 12 micro-steps; ragged last row tile; 41 x 14 tiles
    (None, 10240, 544, 1664, True, False, False),          # 17 K steps (odd: the stage-buffer parity flips per tile), 13 column tiles
    ((8, 56, 56, 1), 25088, 256, 768, True, True, False),  # the layer shape: 1x1, skip pair, ReLU, sign bits
    ((8, 28, 28, 2), 6272, 256, 2816, False, True, True),  # stride 2 with a 1-bit mask, 25 x 22 tiles
    (None, 131072 + 5, 256, 136, False, True, False),      # N = 136: a column tile with 8 valid columns
])
@pytest.mark.parametrize('schedule', [2, 3])
def test_pair_gemm_persistent_kernel_is_bit_identical(conv, M, K, N, relu, res, mask, schedule):
    """Schedule 3 (round 6, the default for launches of more than one tile per CU): one workgroup per CU WALKS the tiles of the ping-pong
    kernel, the next tile's first stage requested before the epilogue.  Schedule 2: the PERSISTENT form of the ping-pong GEMM (256 x 128 tiles, a workgroup per CU walks a run of tiles, the epilogue of a tile is
    deferred into the memory phases of the next tile's K loop) against round 4's two-stage loop with its ordinary epilogue: equal BIT FOR
    BIT, repeatedly, beside a bandwidth hog (the deferred epilogue mixes compiler-counted loads / stores with inline-asm LDS-DMA; a wrong
    wait shows as a rare wrong tile)."""
    from robustart_amd import _lib
    lib = _lib.load()
    d, out, sign, keep = _pair_ps_case(_lib, conv, M, K, N, relu, res, mask, seed=21)
    old = lib.rart_gemm_pair_get_schedule()
    try:
        _lib.check(lib.rart_gemm_pair_set_schedule(0))
        _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
        torch.cuda.synchronize()
        want, want_sign = out.view(torch.int16).clone(), sign.clone()
        assert torch.isfinite(out.float()).all()
        _lib.check(lib.rart_gemm_pair_set_schedule(schedule))
        hog_stream = torch.cuda.Stream()
        hog = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
        for rep in range(6):
            out.fill_(float('nan'))
            sign.zero_()
            if rep >= 2:
                with torch.cuda.stream(hog_stream):
                    for _ in range(4):
                        hog.add_(1.0)
            _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
            torch.cuda.synchronize()
            bad = (out.view(torch.int16) != want).any(0)
            assert not bad.any(), 'persistent kernel differs from the two-stage loop (repetition %d): %d elements, first rows %s' % (
                rep, int(bad.sum()), bad.any(1).nonzero().flatten()[:8].tolist())
            assert torch.equal(sign, want_sign)
    finally:
        lib.rart_gemm_pair_set_schedule(old)
