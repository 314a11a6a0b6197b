"""GPU parity of the hand-written ResNet-50 engine (rart_conv_igemm_bf16 + rart_engine_*):
forward logits and backward-to-input gradient vs plain PyTorch.

Two references, both PyTorch fp32 on the same weights:
  * "emulated": fp32 arithmetic with the engine's storage rounding points reproduced (bf16 weights,
    bf16 activations after every fused conv+bias(+residual)+ReLU, hi+lo input) -- differences are then
    only fp32 summation order, so the tolerance is tight;
  * "pure fp32": no rounding anywhere -- shows the bf16 engine's end-to-end error (loose, stated).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


rb = _RoundBF16.apply


def _emulated_forward(eng, x01):
    """torch restatement of the engine's dataflow (NCHW) in the dtype/device of x01 (fp64 on CPU for the
    tight pins: MIOpen's fp32 convolutions are themselves inexact at the 1e-5 level)."""
    dt = x01.dtype
    mean = torch.tensor(MEAN, device=x01.device, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(STD, device=x01.device, dtype=torch.float32).view(1, 3, 1, 1)
    v = ((x01.float() - mean) * (1.0 / std)).to(dt) if not x01.requires_grad else (x01 - mean.to(dt)) * (1.0 / std).to(dt)
    hi = rb(v)
    v = hi + rb(v - hi)

    def conv(c, t, relu, res=None):
        w = c.w_folded.to(t.device).to(torch.bfloat16).to(dt)
        o = F.conv2d(t, w, c.b_folded.to(t.device).to(dt), stride=c.stride, padding=c.pad)
        if res is not None:
            o = o + res
        if relu:
            o = torch.relu(o)
        return rb(o)

    t = conv(eng.stem, v, True)
    t = F.max_pool2d(t, 3, 2, 1)
    for ca, cb, cc, ds in eng.blocks:
        a = conv(ca, t, True)
        b = conv(cb, a, True)
        sk = conv(ds, t, False) if ds is not None else t
        t = conv(cc, b, True, res=sk)
    p = rb(t.mean((2, 3)))
    wfc = eng.fc_w[:eng.n_classes].to(t.device).to(dt)
    return p @ wfc.t() + eng.fc_b.to(t.device).to(dt)


def _rand_bf16(shape, seed, scale=1.0, relu=False):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(shape, generator=g) * scale
    if relu:
        t = torch.relu(t)
    return t.to(torch.bfloat16)


@pytest.mark.parametrize('cin,cout,k,stride,hw,B', [(64, 256, 1, 1, 14, 3), (256, 64, 1, 1, 14, 3), (64, 64, 3, 1, 14, 3),
                                                   (128, 128, 3, 2, 28, 2), (256, 512, 1, 2, 28, 2),
                                                   (512, 2048, 1, 1, 7, 5)])
def test_conv_kernel_forward_and_backward_layerwise(cin, cout, k, stride, hw, B):
    """rart_conv_igemm_bf16 alone vs fp64 conv of the same bf16 operands: forward with bias / residual /
    ReLU, backward-to-input with residual + ReLU mask (all parity classes for stride 2).  The only
    differences allowed: fp32 accumulation order, then one bf16 rounding of the result."""
    from robustart_amd.model.engine import _Conv, ResNet50Engine
    torch.manual_seed(cin + cout + k)
    conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
    bn = torch.nn.BatchNorm2d(cout).eval()
    bn.running_mean.normal_(0, 0.1)
    bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.1)
    c = _Conv(conv, bn, 'cuda')
    eng = ResNet50Engine.__new__(ResNet50Engine)
    eng.lib, eng.device, eng._buf = __import__('robustart_amd._lib', fromlist=['x']).load(), torch.device('cuda'), {}
    eng.profile = None
    oh = hw // stride
    x = _rand_bf16((B, hw, hw, cin), 1, relu=True)                       # NHWC
    res = _rand_bf16((B, oh, oh, cout), 2)
    out = torch.empty(B, oh, oh, cout, dtype=torch.bfloat16, device='cuda')
    eng._conv_fwd(c, x.cuda(), (hw, hw), out, True, res=res.cuda())
    wq = c.w_folded.to(torch.bfloat16).double()
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), wq, c.b_folded.double(), stride=stride, padding=k // 2)
    ref = torch.relu(ref + res.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    got = out.cpu().double()
    tol = ref.abs() * 2.0 ** -8 + 1e-6                                      # one bf16 ulp
    assert ((got - ref).abs() <= tol).all(), (got - ref).abs().max().item()
    assert ((got - ref.to(torch.bfloat16).double()).abs() > 0).double().mean() < 2e-3   # rounding flips are rare
    # backward to input: dx = conv_transpose(dz) + res2, masked by (xmask > 0)
    dz = _rand_bf16((B, oh, oh, cout), 3, 0.1)
    res2 = _rand_bf16((B, hw, hw, cin), 4, 0.1)
    xmask = _rand_bf16((B, hw, hw, cin), 5)
    dx = torch.zeros(B, hw, hw, cin, dtype=torch.bfloat16, device='cuda')
    if stride == 2 and k == 1:
        dx.copy_(res2.cuda())
        eng._conv_bwd(c, dz.cuda(), (oh, oh), dx, (hw, hw), res=dx, mask=xmask.cuda())   # accumulate form
    else:
        eng._conv_bwd(c, dz.cuda(), (oh, oh), dx, (hw, hw), res=res2.cuda(), mask=xmask.cuda())
    gref = torch.nn.grad.conv2d_input((B, cin, hw, hw), wq, dz.double().permute(0, 3, 1, 2), stride=stride,
                                      padding=k // 2).permute(0, 2, 3, 1)
    m = (xmask.double() > 0).double()
    if stride == 2 and k == 1:
        # only the even/even parity class is produced by a 1x1/2 conv; elsewhere dst keeps res2 (unmasked there)
        want = res2.double().clone()
        want[:, ::2, ::2] = ((gref + res2.double()) * m)[:, ::2, ::2]
    else:
        want = (gref + res2.double()) * m
    got = dx.cpu().double()
    tol = want.abs() * 2.0 ** -8 + 1e-6
    assert ((got - want).abs() <= tol).all(), (got - want).abs().max().item()


def _setup(seed=0):
    from robustart_amd.model import get_model
    from robustart_amd.model.resnet_torch import randomize_bn_stats
    from robustart_amd.model.engine import ResNet50Engine
    torch.manual_seed(seed)
    m = randomize_bn_stats(get_model({'type': 'resnet50_official'}), seed).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    eng = ResNet50Engine(m, 'cuda')
    return m.cuda(), eng


@pytest.fixture(scope='module')
def setup():
    return _setup()


@pytest.mark.parametrize('B,HW', [(3, 96), (2, 224)])
def test_forward_logits(setup, B, HW):
    m, eng = setup
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 3, HW, HW, generator=g).cuda()
    got = eng.logits(x, MEAN, STD)
    want = _emulated_forward(eng, x.cpu().double()).float().cuda()
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    print('forward B=%d HW=%d: logit scale %.3f, max |err| vs emulated(fp64) %.3g' % (B, HW, scale, err))
    # bf16 rounding flips (fp32 vs fp64 accumulation near a rounding boundary, ~1e-3 of the elements per
    # layer) compound over 53 layers; the layer-level test above is the tight pin
    assert err <= 1.5e-2 * scale + 1e-4
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    pure = m((x - mean) / std)
    perr = (got - pure).abs().max().item()
    print('   vs pure fp32: max |err| %.3g (%.2f%% of scale)' % (perr, 100 * perr / scale))
    assert perr <= 0.05 * scale                       # bf16 storage end to end, 53 layers
    assert (got.argmax(1) == pure.argmax(1)).all() or perr > 0


def test_u8_entry_matches_float_entry(setup):
    m, eng = setup
    u8 = torch.randint(0, 256, (2, 96, 96, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
    a = eng.logits_from_u8(u8, MEAN, STD)
    b = eng.logits(u8.permute(0, 3, 1, 2).float() / 255.0, MEAN, STD)
    torch.testing.assert_close(a, b, atol=1.5e-2 * b.abs().max().item(), rtol=0)


def _reference_backward_with_engine_masks(eng, acts, dl, std):
    """fp64 backward-to-input of the folded network, using the ENGINE's forward activations for every
    ReLU / max-pool decision and rounding gradients to bf16 where the engine stores them.  Isolates the
    backward arithmetic from the (chaotic) forward rounding differences."""
    dt = torch.float64
    q = lambda t: t.to(torch.bfloat16).to(dt)        # noqa: E731
    nchw = lambda t: t.detach().cpu().to(dt).permute(0, 3, 1, 2)   # noqa: E731

    def dgrad(c, dz, in_hw):
        w = c.w_folded.to(torch.bfloat16).to(dt)
        return torch.nn.grad.conv2d_input((dz.shape[0], c.cin, in_hw[0], in_hw[1]), w, dz, stride=c.stride,
                                          padding=c.pad)
    B = dl.shape[0]
    dlq = q(dl.detach().cpu().to(dt))
    dpool = q(dlq @ eng.fc_w[:eng.n_classes].cpu().to(dt))
    xl, xlhw = acts['last']
    y = nchw(xl)
    dz = q((y > 0).to(dt) * dpool.view(B, -1, 1, 1) / (xlhw[0] * xlhw[1]))
    for bi in range(len(eng.blocks) - 1, -1, -1):
        ca, cb, cc, ds = eng.blocks[bi]
        x, xhw, ya, yb, yc, ohw = acts['b%d' % bi]
        # the engine's own ReLU decisions: its 1-bit sign tensors (the fused Bottleneck kernels never write ya / yb), or the
        # bf16 activations when sign_bit_masks is off
        mx, ma, mb = acts['b%d_masks' % bi]
        sign = lambda t: (torch.from_numpy(np.unpackbits(t.cpu().numpy(), axis=-1, bitorder='little')).permute(0, 3, 1, 2).to(dt)  # noqa: E731
                          if t.dtype == torch.uint8 else (nchw(t) > 0).to(dt))
        dzb = q(dgrad(cc, dz, ohw) * sign(mb))
        dza = q(dgrad(cb, dzb, xhw) * sign(ma))
        m = sign(mx)
        if ds is None:
            dx = q((dgrad(ca, dza, xhw) + dz) * m)
        else:
            dx = q(dgrad(ca, dza, xhw) * m)
            dx = q((dx + dgrad(ds, dz, xhw)) * m)
        dz = dx
    y1 = nchw(acts['y1']).requires_grad_(True)
    p = F.max_pool2d(y1, 3, 2, 1)
    g1, = torch.autograd.grad(p, y1, grad_outputs=dz)
    dz1 = q(g1 * (y1.detach() > 0))
    g = dgrad(eng.stem, dz1, (y1.shape[2] * 2, y1.shape[3] * 2))
    return g / torch.tensor(std, dtype=dt).view(1, 3, 1, 1)


@pytest.mark.parametrize('B,HW,kind', [(3, 96, 0), (2, 224, 0), (3, 96, 1), (3, 96, 4), (2, 224, 4)])
def test_backward_to_input(setup, B, HW, kind):
    m, eng = setup
    g = torch.Generator().manual_seed(10 + B)
    x = torch.rand(B, 3, HW, HW, generator=g).cuda()
    y = torch.randint(0, 1000, (B,), generator=g).cuda()
    # kind 4 = FAB's gradient of a logit DIFFERENCE z_target - z_y (fab_pt.py:102-117): a +1 / -1 pair in dlogits
    yt = ((y + 1 + torch.randint(0, 998, (B,), generator=g).cuda()) % 1000) if kind == 4 else None
    logits, loss, grad, pred = eng.forward_backward(x, MEAN, STD, y, kind, yt)
    if kind == 4:
        dl = eng.last_dlogits
        rows = torch.arange(B, device='cuda')
        assert torch.equal(dl[rows, yt], torch.ones(B, device='cuda')) and torch.equal(dl[rows, y], -torch.ones(B, device='cuda'))
        assert float(dl.abs().sum()) == 2.0 * B
        torch.testing.assert_close(loss, logits[rows, yt] - logits[rows, y], rtol=0, atol=1e-4 * float(logits.abs().max()))
    assert torch.equal(pred.long(), logits.argmax(1))
    if eng.last_acts['y1'] is None:
        # the fused stem forward never materialises the stem output; the reference below needs it for the ReLU / max-pool
        # decisions: take it from the unfused stem (bit-identical, test_fused_stem_forward_is_bit_identical)
        acts, dl_keep = eng.last_acts, eng.last_dlogits
        try:
            eng.fused_stem_fwd = False
            _, acts_u = eng._forward(x.detach().float().contiguous(), False, MEAN, STD, keep=True)
            acts['y1'] = acts_u['y1'].clone()
            assert torch.equal(acts_u['p1'], acts['p1'])
        finally:
            eng.fused_stem_fwd = True
        eng.last_acts, eng.last_dlogits = acts, dl_keep
    # (1) rigorous: same masks as the engine's forward -> only fp32-accumulate / bf16-rounding noise remains
    ref = _reference_backward_with_engine_masks(eng, eng.last_acts, eng.last_dlogits, STD).cuda()
    for i in range(B):
        a, b = grad[i].flatten().double(), ref[i].flatten().double()
        cos = (a @ b / (a.norm() * b.norm())).item()
        rel = ((a - b).norm() / b.norm()).item()
        print('backward(engine masks) B=%d HW=%d kind=%d img %d: cos %.6f rel-L2 %.5f' % (B, HW, kind, i, cos, rel))
        assert cos > 0.9995 and rel < 0.03
    # (2) end to end vs autograd through the emulated network (its own forward, so its own ReLU masks):
    # bf16 rounding flips in the forward change ~0.5% of the masks per stage and the difference compounds
    # over 16 blocks, so this is a sanity bound, not a pin (see DESIGN.md "engine numerics").
    if kind != 0:
        return      # DLR depends on the ORDER of near-tied logits of a random-init net: not comparable end to end (4: pinned above)
    from robustart_amd.noise.adv import logit_loss
    xr = x.cpu().double().requires_grad_(True)
    out = _emulated_forward(eng, xr)
    _, dl, _ = logit_loss(out.detach().float().cuda(), y, kind)
    gw, = torch.autograd.grad(out, xr, grad_outputs=dl.cpu().double())
    gw = gw.cuda()
    for i in range(B):
        a, b = grad[i].flatten().double(), gw[i].flatten().double()
        cos = (a @ b / (a.norm() * b.norm())).item()
        print('backward(end to end) img %d: cos %.4f norm ratio %.4f' % (i, cos, (a.norm() / b.norm()).item()))
        assert cos > 0.85 and 0.9 < (a.norm() / b.norm()).item() < 1.1


def test_pgd_through_engine_matches_autograd_path(setup):
    """Drop-in check: pgd_linf with the engine-backed f_model vs the same attack driven through torch
    autograd on the emulated network, same injected start: identical sign steps almost everywhere."""
    from robustart_amd.noise import adv
    from robustart_amd.model.engine import EngineModel
    m, eng = setup
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 96, 96, generator=g).cuda()
    y = eng.logits(x, MEAN, STD).argmax(1)
    eps = 4 / 255
    u = ((torch.rand(x.shape, generator=g) * 2 - 1) * eps).cuda()
    f_eng = EngineModel(None, takes_normalized=False, engine=eng)
    a = adv.pgd_linf(x, y, f_eng, eps, 3 / 40, 3, init_u=u)
    b = adv.pgd_linf(x, y, lambda z: _emulated_forward(eng, z), eps, 3 / 40, 3, init_u=u)   # MIOpen fp32 autograd
    assert (a - x).abs().max() <= eps + 1e-6 and a.min() >= 0 and a.max() <= 1
    agree = ((a - b).abs() < 1e-6).float().mean().item()
    print('pgd engine-vs-autograd agreement: %.4f' % agree)
    assert agree > 0.6
    la, lb = eng.logits(a, MEAN, STD), eng.logits(b, MEAN, STD)
    assert (la - lb).abs().max() <= 0.05 * lb.abs().max()


@pytest.mark.gpu
def test_refold_reproduces_constructor_tables_and_tracks_new_weights():
    """ResNet50Engine.refold (GPU-side BatchNorm fold + packing) == the constructor's host-side fold, bit for bit; after
    the parameters change, the refolded engine matches a freshly constructed one."""
    import copy
    import torch
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import ResNet50Engine
    torch.manual_seed(3)
    model = get_model({'type': 'resnet50_official'}).cuda().eval()
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
    eng = ResNet50Engine(model)

    def tables(e):
        out = [e.stem_w, e.stem_wd, e.fc_w, e.fc_wd, e.fc_b, e.stem.bias]
        for blk in e.blocks:
            for c in blk:
                if c is not None:
                    out += [c.w_fwd, c.bias] + [w for _, _, w in c.bwd if w is not None]
        return [t.clone() for t in out]

    before = tables(eng)
    eng.refold(model)
    for a, b in zip(before, tables(eng)):
        assert torch.equal(a, b)
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.0 + 0.05 * torch.randn_like(p))
    eng.refold(model)
    fresh = tables(ResNet50Engine(model))
    for a, b in zip(fresh, tables(eng)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('B,H,W', [(2, 64, 96), (3, 32, 32), (1, 224, 224)])
def test_fused_stem_backward_kernel_vs_fp64(B, H, W):
    """rart_engine_stem_bwd_fused alone: random pooled gradient, random argmax codes (including the dead code 15 and
    out-of-image window positions), random bf16 weights.  Reference in fp64: scatter the pooled gradient to the recorded
    window positions (max-pool backward), round dz1 to bf16 as the kernel stores it, transposed 7x7/2 conv, / std."""
    import ctypes
    from robustart_amd import _lib
    from robustart_amd.model.engine import ResNet50Engine
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + H)
    oh, ow = H // 2, W // 2
    qh, qw = oh // 2, ow // 2
    wb = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(torch.bfloat16).float()
    dpool = torch.randn(B, qh, qw, 64, generator=g).to(torch.bfloat16)
    code = torch.randint(0, 10, (B, qh, qw, 64), generator=g, dtype=torch.uint8)
    code[code == 9] = 15
    # window positions that fall outside the stem-output grid can never be an argmax: remap them to a valid one
    ky, kx = code // 3, code % 3
    qy = torch.arange(qh).view(1, qh, 1, 1)
    qx = torch.arange(qw).view(1, 1, qw, 1)
    live = code != 15
    py = 2 * qy - 1 + ky.long()
    px = 2 * qx - 1 + kx.long()
    bad = live & ((py < 0) | (py >= oh) | (px < 0) | (px >= ow))
    code[bad] = 4                                                   # the window centre is always inside
    ky, kx = code // 3, code % 3
    py = (2 * qy - 1 + ky.long()).clamp(0, oh - 1)
    px = (2 * qx - 1 + kx.long()).clamp(0, ow - 1)
    live = code != 15
    dz1 = torch.zeros(B, oh, ow, 64, dtype=torch.float64)
    bi = torch.arange(B).view(B, 1, 1, 1).expand_as(code)
    ci = torch.arange(64).view(1, 1, 1, 64).expand_as(code)
    dz1.index_put_((bi[live], py.expand_as(code)[live], px.expand_as(code)[live], ci[live]), dpool.double()[live],
                   accumulate=True)
    dz1 = dz1.float().to(torch.bfloat16).double()
    ref = torch.nn.grad.conv2d_input((B, 3, H, W), wb.double(), dz1.permute(0, 3, 1, 2), stride=2, padding=3)
    ref = ref / torch.tensor(STD, dtype=torch.float64).view(1, 3, 1, 1)
    wt = ResNet50Engine._stem_bwd_table(wb).cuda()
    grad = torch.full((B, 3, H, W), float('nan'), device='cuda')
    stdf = (ctypes.c_float * 3)(*STD)
    dpool_d, code_d = dpool.cuda(), code.cuda()      # named: a temporary's block may be handed to the next .cuda() before the launch
    _lib.check(lib.rart_engine_stem_bwd_fused(_lib.ptr(dpool_d), _lib.ptr(code_d), _lib.ptr(wt), _lib.ptr(grad),
                                              B, H, W, stdf, _lib.stream_ptr()))
    got = grad.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    print('fused stem backward B=%d %dx%d: max abs err %.3e (ref max %.3f)' % (B, H, W, err, ref.abs().max().item()))
    assert err <= 2e-5 * max(1.0, ref.abs().max().item())


def test_fused_stem_backward_matches_unfused_chain(setup):
    m, eng = setup
    g = torch.Generator().manual_seed(77)
    x = torch.rand(3, 3, 96, 128, generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    try:
        eng.fused_stem_bwd = True
        _, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        ga = ga.clone()
        eng.fused_stem_bwd = eng.fused_stem_fwd = False
        _, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    finally:
        eng.fused_stem_bwd = eng.fused_stem_fwd = True
    a, b = ga.flatten().double(), gb.flatten().double()
    cos = (a @ b / (a.norm() * b.norm())).item()
    rel = ((a - b).norm() / b.norm()).item()
    print('fused vs unfused stem backward: cos %.6f rel-L2 %.5f' % (cos, rel))
    assert cos > 0.9999 and rel < 0.01      # the unfused chain rounds the 147 patch columns to bf16, the fused one does not


def test_sign_bit_masks_match_bf16_masks_exactly(setup):
    """The backward GEMMs read the ReLU masks as 1-bit tensors written by the forward epilogues (rart_conv_desc.sign_out,
    flag 16) instead of re-reading the bf16 activations: the decision (stored bf16 > 0) is the same, so the gradient must
    be BIT-identical to the bf16-mask path; also at a stride-2 / projection block boundary and odd batch."""
    m, eng = setup
    g = torch.Generator().manual_seed(91)
    x = torch.rand(3, 3, 96, 64, generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    try:
        eng.sign_bit_masks = True
        la, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        ga = ga.clone()
        eng.sign_bit_masks = False
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    finally:
        eng.sign_bit_masks = True
    assert torch.equal(la, lb)
    assert torch.equal(ga, gb)
    assert ga.abs().max() > 0


@pytest.mark.parametrize('C,B,H,W', [(64, 3, 56, 56), (64, 2, 24, 24), (64, 5, 7, 9), (128, 3, 28, 28), (128, 2, 12, 12),
                                     (128, 1, 5, 27), (256, 3, 14, 14), (256, 2, 6, 10), (256, 1, 1, 1)])
def test_halo_conv3x3_forward_and_backward_vs_fp64(C, B, H, W):
    """rart_conv3x3_halo_bf16 (input halo tile resident in LDS) against an fp64 evaluation of the same bf16 operands:
    forward (bias + ReLU + sign bits) and backward-to-input (flipped taps + 1-bit mask); image boundaries inside a
    workgroup's run of positions, odd sizes, positions past the end."""
    from robustart_amd import _lib
    from robustart_amd.model.engine import _Conv, _cints
    lib = _lib.load()
    assert lib.rart_conv3x3_halo_supported(C, H, W)
    g = torch.Generator().manual_seed(C + B + H)
    conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5)
    c = _Conv(conv, None, 'cuda')
    c.bias.copy_(torch.randn(C, generator=g) * 0.1)
    x = _rand_bf16((B, H, W, C), 5, relu=True).cuda()
    y = torch.empty_like(x)
    sign = torch.zeros(B, H, W, C // 8, dtype=torch.uint8, device='cuda')
    sp = _lib.stream_ptr()
    wf, wb_ = (torch.empty(9 * C * C, dtype=torch.bfloat16, device='cuda') for _ in range(2))
    _lib.check(lib.rart_conv3x3_pack_frag_bf16(_lib.ptr(c.w_fwd), _lib.ptr(wf), C, sp))
    _lib.check(lib.rart_conv3x3_pack_frag_bf16(_lib.ptr(c.bwd[0][2]), _lib.ptr(wb_), C, sp))
    _lib.check(lib.rart_conv3x3_halo_bf16(_lib.ptr(x), _lib.ptr(wf), _lib.ptr(c.bias), None, _lib.ptr(sign), _lib.ptr(y),
                                          B, H, W, C, _cints([t[0] for t in c.fwd_taps]), _cints([t[1] for t in c.fwd_taps]),
                                          1, sp))
    wq = conv.weight.detach().to(torch.bfloat16).double()
    ref = F.relu(F.conv2d(x.cpu().double().permute(0, 3, 1, 2), wq, c.bias.cpu().double(), padding=1)).permute(0, 2, 3, 1)
    got = y.cpu().double()
    ulp = ref.abs().clamp_min(2.0 ** -20) * 2.0 ** -8
    assert ((got - ref).abs() <= ulp + 1e-6).all(), (got - ref).abs().max()
    bits = torch.from_numpy(np.unpackbits(sign.cpu().numpy(), axis=-1, bitorder='little')).bool()
    assert torch.equal(bits, y.cpu() > 0)
    # backward to input with a random 1-bit mask
    dz = _rand_bf16((B, H, W, C), 6).cuda()
    mask = torch.randint(0, 256, (B, H, W, C // 8), generator=g, dtype=torch.uint8).cuda()
    dx = torch.empty_like(dz)
    taps = c.bwd[0][1]
    _lib.check(lib.rart_conv3x3_halo_bf16(_lib.ptr(dz), _lib.ptr(wb_), None, _lib.ptr(mask), None, _lib.ptr(dx),
                                          B, H, W, C, _cints([t[0] for t in taps]), _cints([t[1] for t in taps]), 0, sp))
    refg = torch.nn.grad.conv2d_input((B, C, H, W), wq, dz.cpu().double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    mb = torch.from_numpy(np.unpackbits(mask.cpu().numpy(), axis=-1, bitorder='little')).double()
    refg = refg * mb
    gotg = dx.cpu().double()
    ulp = refg.abs().clamp_min(2.0 ** -20) * 2.0 ** -8
    assert ((gotg - refg).abs() <= ulp + 1e-6).all(), (gotg - refg).abs().max()


def test_halo_conv3x3_engine_matches_generic_igemm(setup):
    m, eng = setup
    g = torch.Generator().manual_seed(123)
    x = torch.rand(3, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    try:
        eng.halo_conv3x3 = True
        la, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        ga = ga.clone()
        eng.halo_conv3x3 = False
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    finally:
        eng.halo_conv3x3 = True
    # same bf16 operands, fp32 accumulation in a different order: rare 1-ulp bf16 flips that then propagate
    assert (la - lb).abs().max() <= 0.02 * lb.abs().max()
    a, b = ga.flatten().double(), gb.flatten().double()
    cos = (a @ b / (a.norm() * b.norm())).item()
    print('halo vs generic 3x3: logits max diff %.4f (scale %.2f), grad cos %.6f' % ((la - lb).abs().max().item(), lb.abs().max().item(), cos))
    assert cos > 0.995


@pytest.mark.parametrize('B,H,W,u8', [(3, 224, 224, False), (2, 96, 160, True), (5, 32, 64, False)])
def test_fused_stem_forward_is_bit_identical(setup, B, H, W, u8):
    """rart_engine_stem_fwd_fused (normalise + hi/lo split + 7x7/2 conv + ReLU + max pool in one persistent kernel) against
    the three-kernel chain it replaces: same K order and operands, so the pooled activation, the argmax codes and the sign
    bits must be BIT-identical (image borders, partial pooled tiles, fp32 and uint8 entries)."""
    m, eng = setup
    g = torch.Generator().manual_seed(B * 7 + H)
    if u8:
        src = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).cuda()
    else:
        src = torch.rand(B, 3, H, W, generator=g).cuda()
    out = {}
    try:
        for fused in (True, False):
            eng.fused_stem_fwd = fused
            _, acts = eng._forward(src, u8, MEAN, STD, keep=True)
            out[fused] = (acts['p1'].clone(), acts['p1_argmax'].clone(), eng._buf['p1_sign'].clone())
    finally:
        eng.fused_stem_fwd = True
    for a, b, name in zip(out[True], out[False], ('p1', 'argmax', 'sign')):
        assert torch.equal(a, b), name
    assert (out[True][0] > 0).any() and (out[True][1] == 15).any()


@pytest.mark.parametrize('B', [1, 3])
def test_fused_bottleneck_forward_and_backward_vs_fp64(B):
    """rart_bottleneck_fused_bf16 (1x1 -> 3x3 -> 1x1 + input with both 64-channel intermediates on chip) against an fp64
    evaluation of the same bf16 operands with the intermediates rounded to bf16 where the kernel rounds them: forward
    (biases, ReLUs, the three sign tensors) and backward-to-input (three 1-bit masks, flipped taps, residual gradient).
    A 1-ulp flip of an intermediate (fp32 accumulation order) moves a few outputs by more than their own ulp, so the
    bound is 1 ulp for all but 1e-3 of the elements and 1 % of the tensor scale for every element."""
    from robustart_amd import _lib
    from robustart_amd.model.engine import _Conv, _cints
    lib = _lib.load()
    H = W = 56
    assert lib.rart_bottleneck_fused_supported(256, 64, H, W)
    assert not lib.rart_bottleneck_fused_supported(512, 128, 28, 28)
    g = torch.Generator().manual_seed(40 + B)

    def mk(cin, cout, k):
        conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5)
        c = _Conv(conv, None, 'cuda')
        c.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        return conv, c

    (c1, ca), (c2, cb), (c3, cc) = mk(256, 64, 1), mk(64, 64, 3), mk(64, 256, 1)
    wq = [c.weight.detach().to(torch.bfloat16).double() for c in (c1, c2, c3)]
    bq = [c.bias.cpu().double() for c in (ca, cb, cc)]
    rb = lambda t: t.to(torch.bfloat16).double()            # where the kernel rounds
    x = _rand_bf16((B, H, W, 256), 5, relu=True).cuda()
    y = torch.empty_like(x)
    s1 = torch.zeros(B, H, W, 8, dtype=torch.uint8, device='cuda')
    s2 = torch.zeros_like(s1)
    s3 = torch.zeros(B, H, W, 32, dtype=torch.uint8, device='cuda')
    sp = _lib.stream_ptr()
    dy, dx_ = _cints([t[0] for t in cb.fwd_taps]), _cints([t[1] for t in cb.fwd_taps])
    w2f, w2b = torch.empty(64 * 576, dtype=torch.bfloat16, device='cuda'), torch.empty(64 * 576, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_conv3x3_pack_frag_bf16(_lib.ptr(cb.w_fwd), _lib.ptr(w2f), 64, sp))
    _lib.check(lib.rart_conv3x3_pack_frag_bf16(_lib.ptr(cb.bwd[0][2]), _lib.ptr(w2b), 64, sp))
    _lib.check(lib.rart_bottleneck_fused_bf16(_lib.ptr(x), _lib.ptr(ca.w_fwd), _lib.ptr(w2f), _lib.ptr(cc.w_fwd),
                                              _lib.ptr(ca.bias), _lib.ptr(cb.bias), _lib.ptr(cc.bias), _lib.ptr(s1), _lib.ptr(s2),
                                              _lib.ptr(s3), _lib.ptr(y), B, H, W, 256, 64, dy, dx_, 0, sp))
    xd = x.cpu().double().permute(0, 3, 1, 2)
    a1 = rb(F.relu(F.conv2d(xd, wq[0], bq[0])))
    a2 = rb(F.relu(F.conv2d(a1, wq[1], bq[1], padding=1)))
    ref = F.relu(F.conv2d(a2, wq[2], bq[2]) + xd).permute(0, 2, 3, 1)
    got = y.cpu().double()

    def close(got, ref, what):
        err = (got - ref).abs()
        ulp = ref.abs().clamp_min(2.0 ** -20) * 2.0 ** -8 + 1e-6
        frac = (err > ulp).double().mean().item()
        print('%s: beyond 1 ulp %.2e of the elements, max err %.4f (scale %.2f)' % (what, frac, err.max().item(), ref.abs().max().item()))
        assert frac < 1e-3 and err.max() <= 0.01 * ref.abs().max(), what

    close(got, ref, 'forward')
    unpack = lambda t: torch.from_numpy(np.unpackbits(t.cpu().numpy(), axis=-1, bitorder='little')).bool()
    assert torch.equal(unpack(s3), y.cpu() > 0)
    for s, a, name in ((s1, a1, 'a1'), (s2, a2, 'a2')):
        mism = (unpack(s) != (a.permute(0, 2, 3, 1) > 0)).double().mean().item()
        assert mism < 1e-4, (name, mism)
    # forward without sign outputs (the evaluation path) writes the same activations
    y2 = torch.empty_like(x)
    _lib.check(lib.rart_bottleneck_fused_bf16(_lib.ptr(x), _lib.ptr(ca.w_fwd), _lib.ptr(w2f), _lib.ptr(cc.w_fwd),
                                              _lib.ptr(ca.bias), _lib.ptr(cb.bias), _lib.ptr(cc.bias), None, None, None,
                                              _lib.ptr(y2), B, H, W, 256, 64, dy, dx_, 0, sp))
    assert torch.equal(y, y2)
    # backward to input with random masks
    gz = _rand_bf16((B, H, W, 256), 6).cuda()
    mb = torch.randint(0, 256, (B, H, W, 8), generator=g, dtype=torch.uint8).cuda()      # sign of the conv2 output
    ma = torch.randint(0, 256, (B, H, W, 8), generator=g, dtype=torch.uint8).cuda()      # sign of the conv1 output
    mx = torch.randint(0, 256, (B, H, W, 32), generator=g, dtype=torch.uint8).cuda()     # sign of the block input
    dx = torch.empty_like(gz)
    taps = cb.bwd[0][1]
    _lib.check(lib.rart_bottleneck_fused_bf16(_lib.ptr(gz), _lib.ptr(cc.bwd[0][2]), _lib.ptr(w2b), _lib.ptr(ca.bwd[0][2]),
                                              None, None, None, _lib.ptr(mb), _lib.ptr(ma), _lib.ptr(mx), _lib.ptr(dx), B, H, W,
                                              256, 64, _cints([t[0] for t in taps]), _cints([t[1] for t in taps]), 1, sp))
    bits = lambda t: unpack(t).double().permute(0, 3, 1, 2)
    gd = gz.cpu().double().permute(0, 3, 1, 2)
    d2 = rb(torch.nn.grad.conv2d_input((B, 64, H, W), wq[2], gd) * bits(mb))
    d1 = rb(torch.nn.grad.conv2d_input((B, 64, H, W), wq[1], d2, padding=1) * bits(ma))
    refg = ((torch.nn.grad.conv2d_input((B, 256, H, W), wq[0], d1) + gd) * bits(mx)).permute(0, 2, 3, 1)
    close(dx.cpu().double(), refg, 'backward')


@pytest.mark.parametrize('B', [1, 3])
def test_fused_first_bottleneck_forward_and_backward_vs_fp64(B):
    """rart_bottleneck_first_bf16 (layer1 block 0: 64 -> 64 -> 256 with the projection shortcut folded into the last GEMM as
    extra K) against fp64 with bf16 rounding of the two intermediates; same bounds as the identity-block test."""
    from robustart_amd import _lib
    from robustart_amd.model.engine import _Conv, _cints
    lib = _lib.load()
    H = W = 56
    assert lib.rart_bottleneck_first_supported(64, 64, 256, H, W)
    g = torch.Generator().manual_seed(70 + B)

    def mk(cin, cout, k):
        conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5)
        c = _Conv(conv, None, 'cuda')
        c.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        return conv, c

    (c1, ca), (c2, cb), (c3, cc), (c4, ds) = mk(64, 64, 1), mk(64, 64, 3), mk(64, 256, 1), mk(64, 256, 1)
    wq = [c.weight.detach().to(torch.bfloat16).double() for c in (c1, c2, c3, c4)]
    bq = [c.bias.cpu().double() for c in (ca, cb, cc, ds)]
    rb = lambda t: t.to(torch.bfloat16).double()
    sp = _lib.stream_ptr()
    new = lambda n: torch.empty(n, dtype=torch.bfloat16, device='cuda')
    w2f, w2b, w4f = new(64 * 576), new(64 * 576), new(256 * 64)
    _lib.check(lib.rart_conv3x3_pack_frag_bf16(_lib.ptr(cb.w_fwd), _lib.ptr(w2f), 64, sp))
    _lib.check(lib.rart_conv3x3_pack_frag_bf16(_lib.ptr(cb.bwd[0][2]), _lib.ptr(w2b), 64, sp))
    _lib.check(lib.rart_pack_frag_bf16(_lib.ptr(ds.w_fwd), _lib.ptr(w4f), 256, 64, sp))
    b3 = (cc.bias + ds.bias).contiguous()
    x = _rand_bf16((B, H, W, 64), 5, relu=True).cuda()
    y = torch.empty(B, H, W, 256, dtype=torch.bfloat16, device='cuda')
    s1 = torch.zeros(B, H, W, 8, dtype=torch.uint8, device='cuda')
    s2 = torch.zeros_like(s1)
    s3 = torch.zeros(B, H, W, 32, dtype=torch.uint8, device='cuda')
    dy, dx_ = _cints([t[0] for t in cb.fwd_taps]), _cints([t[1] for t in cb.fwd_taps])
    _lib.check(lib.rart_bottleneck_first_bf16(_lib.ptr(x), _lib.ptr(ca.w_fwd), _lib.ptr(w2f), _lib.ptr(cc.w_fwd), _lib.ptr(w4f),
                                              _lib.ptr(ca.bias), _lib.ptr(cb.bias), _lib.ptr(b3), _lib.ptr(s1), _lib.ptr(s2),
                                              _lib.ptr(s3), _lib.ptr(y), B, H, W, 64, 64, 256, dy, dx_, 0, sp))
    xd = x.cpu().double().permute(0, 3, 1, 2)
    a1 = rb(F.relu(F.conv2d(xd, wq[0], bq[0])))
    a2 = rb(F.relu(F.conv2d(a1, wq[1], bq[1], padding=1)))
    ref = F.relu(F.conv2d(a2, wq[2], bq[2]) + F.conv2d(xd, wq[3], bq[3])).permute(0, 2, 3, 1)

    def close(got, ref, what):
        err = (got - ref).abs()
        ulp = ref.abs().clamp_min(2.0 ** -20) * 2.0 ** -8 + 1e-6
        frac = (err > ulp).double().mean().item()
        print('%s: beyond 1 ulp %.2e of the elements, max err %.4f (scale %.2f)' % (what, frac, err.max().item(), ref.abs().max().item()))
        assert frac < 1e-3 and err.max() <= 0.01 * ref.abs().max(), what

    close(y.cpu().double(), ref, 'first block forward')
    unpack = lambda t: torch.from_numpy(np.unpackbits(t.cpu().numpy(), axis=-1, bitorder='little')).bool()
    assert torch.equal(unpack(s3), y.cpu() > 0)
    for s_, a, name in ((s1, a1, 'a1'), (s2, a2, 'a2')):
        mism = (unpack(s_) != (a.permute(0, 2, 3, 1) > 0)).double().mean().item()
        assert mism < 1e-4, (name, mism)
    # backward to the 64-channel input
    gz = _rand_bf16((B, H, W, 256), 6).cuda()
    mb = torch.randint(0, 256, (B, H, W, 8), generator=g, dtype=torch.uint8).cuda()
    ma = torch.randint(0, 256, (B, H, W, 8), generator=g, dtype=torch.uint8).cuda()
    mx = torch.randint(0, 256, (B, H, W, 8), generator=g, dtype=torch.uint8).cuda()
    dx = torch.empty(B, H, W, 64, dtype=torch.bfloat16, device='cuda')
    taps = cb.bwd[0][1]
    _lib.check(lib.rart_bottleneck_first_bf16(_lib.ptr(gz), _lib.ptr(cc.bwd[0][2]), _lib.ptr(w2b), _lib.ptr(ds.bwd[0][2]),
                                              _lib.ptr(ca.bwd[0][2]), None, None, None, _lib.ptr(mb), _lib.ptr(ma), _lib.ptr(mx),
                                              _lib.ptr(dx), B, H, W, 64, 64, 256, _cints([t[0] for t in taps]),
                                              _cints([t[1] for t in taps]), 1, sp))
    bits = lambda t: unpack(t).double().permute(0, 3, 1, 2)
    gd = gz.cpu().double().permute(0, 3, 1, 2)
    d2 = rb(torch.nn.grad.conv2d_input((B, 64, H, W), wq[2], gd) * bits(mb))
    d1 = rb(torch.nn.grad.conv2d_input((B, 64, H, W), wq[1], d2, padding=1) * bits(ma))
    refg = ((torch.nn.grad.conv2d_input((B, 64, H, W), wq[0], d1) + torch.nn.grad.conv2d_input((B, 64, H, W), wq[3], gd))
            * bits(mx)).permute(0, 2, 3, 1)
    close(dx.cpu().double(), refg, 'first block backward')


def test_fused_bottleneck_engine_matches_three_launch_chain(setup):
    m, eng = setup
    g = torch.Generator().manual_seed(321)
    x = torch.rand(3, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    try:
        eng.fused_bottleneck = True
        eng.profile = []
        la, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        kinds = [p[3] for p in eng.profile]
        eng.profile = None
        assert kinds.count('bottleneck') == 6, kinds          # layer1 blocks 0, 1 and 2, forward and backward
        assert kinds.count('bottleneck14') == 10, kinds       # layer3 identity blocks, forward and backward
        assert kinds.count('bottleneck28') == 6, kinds        # layer2 identity blocks, forward and backward
        assert kinds.count('bottleneck7') == 4, kinds         # layer4 identity blocks, forward and backward
        ga = ga.clone()
        ea = eng.logits(x, MEAN, STD).clone()
        eng.fused_bottleneck = eng.fused_bottleneck14 = False
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        eb = eng.logits(x, MEAN, STD)
    finally:
        eng.fused_bottleneck = eng.fused_bottleneck14 = True
        eng.profile = None
    assert torch.equal(la, ea)                                # with and without sign outputs: same activations
    assert (la - lb).abs().max() <= 0.02 * lb.abs().max() and (ea - eb).abs().max() <= 0.02 * eb.abs().max()
    a, b = ga.flatten().double(), gb.flatten().double()
    cos = (a @ b / (a.norm() * b.norm())).item()
    print('fused bottleneck vs chain: logits max diff %.4f (scale %.2f), grad cos %.6f' % ((la - lb).abs().max().item(), lb.abs().max().item(), cos))
    # the identity blocks alone reproduce the chain bit for bit; the fused FIRST block adds the projection shortcut in fp32
    # where the chain rounds it to bf16 first, and on this random-init network that one rounding flips ReLU decisions through
    # all 16 blocks (the same effect test_backward_to_input bounds by 0.85 end to end).  The arithmetic itself is pinned by
    # the fp64 tests above and by test_backward_to_input's same-mask comparison; tests/test_outcome_gpu.py measures the
    # gradient direction on a fitted network.
    assert cos > 0.9


@pytest.mark.parametrize('B,geo', [(1, 14), (3, 14), (1, 28), (3, 28), (1, 7), (3, 7)])
def test_fused_bottleneck14_and_28_forward_and_backward_vs_fp64(B, geo):
    """rart_bottleneck14_fused_bf16 (layer3 identity block, one image per workgroup) and rart_bottleneck28_fused_bf16 (layer2,
    a quarter image per workgroup with its a1 halo recomputed), rart_bottleneck7_fused_bf16 (layer4, two 32-channel blocks per wave): x streamed through LDS slices, both intermediates LDS resident;
    against fp64 with bf16 rounding where the kernels round; same bounds as the layer1 test."""
    from robustart_amd import _lib
    from robustart_amd.model.engine import _Conv, _cints
    lib = _lib.load()
    H = W = geo
    CIO, CM = {14: (1024, 256), 28: (512, 128), 7: (2048, 512)}[geo]
    fused = {14: lib.rart_bottleneck14_fused_bf16, 28: lib.rart_bottleneck28_fused_bf16, 7: lib.rart_bottleneck7_fused_bf16}[geo]
    assert {14: lib.rart_bottleneck14_fused_supported, 28: lib.rart_bottleneck28_fused_supported,
            7: lib.rart_bottleneck7_fused_supported}[geo](CIO, CM, H, W)
    g = torch.Generator().manual_seed(140 + B)

    def mk(cin, cout, k):
        conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5)
        c = _Conv(conv, None, 'cuda')
        c.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        return conv, c

    (c1, ca), (c2, cb), (c3, cc) = mk(CIO, CM, 1), mk(CM, CM, 3), mk(CM, CIO, 1)
    wq = [c.weight.detach().to(torch.bfloat16).double() for c in (c1, c2, c3)]
    bq = [c.bias.cpu().double() for c in (ca, cb, cc)]
    rb = lambda t: t.to(torch.bfloat16).double()
    sp = _lib.stream_ptr()
    new = lambda n: torch.empty(n, dtype=torch.bfloat16, device='cuda')

    def frag(tab, rows, k):
        o = new(rows * k)
        _lib.check(lib.rart_pack_frag_bf16(_lib.ptr(tab), _lib.ptr(o), rows, k, sp))
        return o

    w1f, w3f = frag(ca.w_fwd, CM, CIO), frag(cc.w_fwd, CIO, CM)
    w1b, w3b = frag(cc.bwd[0][2], CM, CIO), frag(ca.bwd[0][2], CIO, CM)
    w2f, w2b = new(9 * CM * CM), new(9 * CM * CM)
    _lib.check(lib.rart_pack_frag_bf16(_lib.ptr(cb.w_fwd), _lib.ptr(w2f), CM, 9 * CM, sp))
    _lib.check(lib.rart_pack_frag_bf16(_lib.ptr(cb.bwd[0][2]), _lib.ptr(w2b), CM, 9 * CM, sp))
    x = _rand_bf16((B, H, W, CIO), 5, relu=True).cuda()
    y = torch.empty_like(x)
    s1 = torch.zeros(B, H, W, CM // 8, dtype=torch.uint8, device='cuda')
    s2 = torch.zeros_like(s1)
    s3 = torch.zeros(B, H, W, CIO // 8, dtype=torch.uint8, device='cuda')
    dy, dx_ = _cints([t[0] for t in cb.fwd_taps]), _cints([t[1] for t in cb.fwd_taps])
    _lib.check(fused(_lib.ptr(x), _lib.ptr(w1f), _lib.ptr(w2f), _lib.ptr(w3f), _lib.ptr(ca.bias),
                                                _lib.ptr(cb.bias), _lib.ptr(cc.bias), _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(s3),
                                                _lib.ptr(y), B, H, W, CIO, CM, dy, dx_, 0, sp))
    xd = x.cpu().double().permute(0, 3, 1, 2)
    a1 = rb(F.relu(F.conv2d(xd, wq[0], bq[0])))
    a2 = rb(F.relu(F.conv2d(a1, wq[1], bq[1], padding=1)))
    ref = F.relu(F.conv2d(a2, wq[2], bq[2]) + xd).permute(0, 2, 3, 1)

    def close(got, ref, what, frac_max=1e-3):
        err = (got - ref).abs()
        ulp = ref.abs().clamp_min(2.0 ** -20) * 2.0 ** -8 + 1e-6
        frac = (err > ulp).double().mean().item()
        print('%s: beyond 1 ulp %.2e of the elements, max err %.4f (scale %.2f)' % (what, frac, err.max().item(), ref.abs().max().item()))
        assert frac < frac_max and err.max() <= 0.01 * ref.abs().max(), what

    # (the K = 2048 / 4608 sums of the 7 x 7 block flip more 1-ulp roundings of the intermediates; the absolute bound is what matters)
    close(y.cpu().double(), ref, 'fused block forward', frac_max=3e-3 if geo != 7 else 3e-2)
    unpack = lambda t: torch.from_numpy(np.unpackbits(t.cpu().numpy(), axis=-1, bitorder='little')).bool()
    assert torch.equal(unpack(s3), y.cpu() > 0)
    for s_, a, name in ((s1, a1, 'a1'), (s2, a2, 'a2')):
        mism = (unpack(s_) != (a.permute(0, 2, 3, 1) > 0)).double().mean().item()
        assert mism < 1e-4, (name, mism)
    y2 = torch.empty_like(x)
    _lib.check(fused(_lib.ptr(x), _lib.ptr(w1f), _lib.ptr(w2f), _lib.ptr(w3f), _lib.ptr(ca.bias),
                                                _lib.ptr(cb.bias), _lib.ptr(cc.bias), None, None, None, _lib.ptr(y2), B, H, W,
                                                CIO, CM, dy, dx_, 0, sp))
    assert torch.equal(y, y2)
    gz = _rand_bf16((B, H, W, CIO), 6).cuda()
    mb = torch.randint(0, 256, (B, H, W, CM // 8), generator=g, dtype=torch.uint8).cuda()
    ma = torch.randint(0, 256, (B, H, W, CM // 8), generator=g, dtype=torch.uint8).cuda()
    mx = torch.randint(0, 256, (B, H, W, CIO // 8), generator=g, dtype=torch.uint8).cuda()
    dx = torch.empty_like(gz)
    taps = cb.bwd[0][1]
    _lib.check(fused(_lib.ptr(gz), _lib.ptr(w1b), _lib.ptr(w2b), _lib.ptr(w3b), None, None, None,
                                                _lib.ptr(mb), _lib.ptr(ma), _lib.ptr(mx), _lib.ptr(dx), B, H, W, CIO, CM,
                                                _cints([t[0] for t in taps]), _cints([t[1] for t in taps]), 1, sp))
    bits = lambda t: unpack(t).double().permute(0, 3, 1, 2)
    gd = gz.cpu().double().permute(0, 3, 1, 2)
    d2 = rb(torch.nn.grad.conv2d_input((B, CM, H, W), wq[2], gd) * bits(mb))
    d1 = rb(torch.nn.grad.conv2d_input((B, CM, H, W), wq[1], d2, padding=1) * bits(ma))
    refg = ((torch.nn.grad.conv2d_input((B, CIO, H, W), wq[0], d1) + gd) * bits(mx)).permute(0, 2, 3, 1)
    # K = 1024 sums of signed values: more 1-ulp flips of the two intermediates than in the 256-channel block, and many outputs
    # near zero where "1 ulp of the reference" is tiny; the absolute bound (1 % of the scale) is the meaningful one here
    close(dx.cpu().double(), refg, 'fused block backward', frac_max=2e-2 if geo != 7 else 5e-2)


@pytest.mark.parametrize('B,geo', [(1, 56), (3, 56), (1, 28), (2, 28)])
def test_fused_stride2_bottleneck_forward_vs_fp64(B, geo):
    """rart_bottleneck_s2_fwd_bf16: the stride-2 first block of layer2 (56 x 56, 256 -> 128 -> 512) and layer3 (28 x 28, 512 -> 256 ->
    1024) with its projection shortcut as one kernel: a1 on the 16 x 16 input grid behind a 7 x 7 output tile, the 3x3 / 2 as a
    stride in the slot -> address map, the shortcut as extra K of the last stage.  Against fp64 with bf16 rounding where the kernel
    rounds (a1, a2, the output); the three 1-bit sign tensors against the same reference."""
    from robustart_amd import _lib
    from robustart_amd.model.engine import _Conv
    lib = _lib.load()
    H = W = geo
    CIN, CM, COUT = {56: (256, 128, 512), 28: (512, 256, 1024)}[geo]
    assert lib.rart_bottleneck_s2_fwd_supported(CIN, CM, COUT, H, W)
    assert not lib.rart_bottleneck_s2_fwd_supported(1024, 512, 2048, 14, 14)        # layer4's block: a1 image does not fit LDS
    g = torch.Generator().manual_seed(300 + B + geo)

    def mk(cin, cout, k, stride):
        conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5)
        c = _Conv(conv, None, 'cuda')
        c.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        return conv, c

    (c1, ca), (c2, cb), (c3, cc), (c4, ds) = mk(CIN, CM, 1, 1), mk(CM, CM, 3, 2), mk(CM, COUT, 1, 1), mk(CIN, COUT, 1, 2)
    wq = [c.weight.detach().to(torch.bfloat16).double() for c in (c1, c2, c3, c4)]
    bq = [c.bias.cpu().double() for c in (ca, cb, cc, ds)]
    rb = lambda t: t.to(torch.bfloat16).double()      # noqa: E731
    sp = _lib.stream_ptr()

    def frag(tab, rows, k):
        o = torch.empty(rows * k, dtype=torch.bfloat16, device='cuda')
        _lib.check(lib.rart_pack_frag_bf16(_lib.ptr(tab), _lib.ptr(o), rows, k, sp))
        return o

    w1, w2, w3, wd = frag(ca.w_fwd, CM, CIN), frag(cb.w_fwd, CM, 9 * CM), frag(cc.w_fwd, COUT, CM), frag(ds.w_fwd, COUT, CIN)
    b3 = (cc.bias + ds.bias).contiguous()
    x = _rand_bf16((B, H, W, CIN), 7, relu=True).cuda()
    y = torch.full((B, H // 2, W // 2, COUT), float('nan'), dtype=torch.bfloat16, device='cuda')
    s1 = torch.zeros(B, H, W, CM // 8, dtype=torch.uint8, device='cuda')
    s2 = torch.zeros(B, H // 2, W // 2, CM // 8, dtype=torch.uint8, device='cuda')
    s3 = torch.zeros(B, H // 2, W // 2, COUT // 8, dtype=torch.uint8, device='cuda')
    _lib.check(lib.rart_bottleneck_s2_fwd_bf16(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(w3), _lib.ptr(wd), _lib.ptr(ca.bias),
                                               _lib.ptr(cb.bias), _lib.ptr(b3), _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(s3), _lib.ptr(y),
                                               B, H, W, CIN, CM, COUT, sp))
    xd = x.cpu().double().permute(0, 3, 1, 2)
    a1 = rb(F.relu(F.conv2d(xd, wq[0], bq[0])))
    a2 = rb(F.relu(F.conv2d(a1, wq[1], bq[1], stride=2, padding=1)))
    ref = F.relu(F.conv2d(a2, wq[2], bq[2]) + F.conv2d(xd, wq[3], bq[3], stride=2)).permute(0, 2, 3, 1)
    got = y.cpu().double()
    assert torch.isfinite(got).all()                                   # every output position was written
    err = (got - ref).abs()
    ulp = ref.abs().clamp_min(2.0 ** -20) * 2.0 ** -8 + 1e-6
    frac = (err > ulp).double().mean().item()
    print('fused stride-2 block forward %d x %d: beyond 1 ulp %.2e of the elements, max err %.4f (scale %.2f)'
          % (geo, geo, frac, err.max().item(), ref.abs().max().item()))
    assert frac < 3e-3 and err.max() <= 0.01 * ref.abs().max()
    unpack = lambda t: torch.from_numpy(np.unpackbits(t.cpu().numpy(), axis=-1, bitorder='little')).bool()      # noqa: E731
    assert torch.equal(unpack(s3), y.cpu() > 0)
    for s_, a, name in ((s1, a1, 'a1'), (s2, a2, 'a2')):
        mism = (unpack(s_) != (a.permute(0, 2, 3, 1) > 0)).double().mean().item()
        assert mism < 1e-4, (name, mism)
    y2 = torch.empty_like(y)
    _lib.check(lib.rart_bottleneck_s2_fwd_bf16(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(w3), _lib.ptr(wd), _lib.ptr(ca.bias),
                                               _lib.ptr(cb.bias), _lib.ptr(b3), None, None, None, _lib.ptr(y2), B, H, W, CIN, CM, COUT, sp))
    assert torch.equal(y, y2)                                          # the sign outputs are optional


def test_engine_with_and_without_the_fused_stride2_blocks(setup):
    """Engine switch fused_bottleneck_s2: the one-kernel stride-2 blocks against the four-launch chains end to end (logits, and the
    gradient w.r.t. the input, whose backward pass reads the sign tensors the fused forward wrote)."""
    m, eng = setup
    g = torch.Generator().manual_seed(77)
    x = torch.rand(3, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    try:
        eng.fused_bottleneck_s2 = eng.fused_bottleneck_s2_bwd = True
        la, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        la, ga = la.clone(), ga.clone()
        # same forward (same masks), the backward of the two blocks as seven conv launches each: only bf16 rounding of the chain's
        # extra intermediates differs
        eng.fused_bottleneck_s2_bwd = False
        _, _, gc, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        gc = gc.clone()
        eng.fused_bottleneck_s2 = False
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    finally:
        eng.fused_bottleneck_s2 = eng.fused_bottleneck_s2_bwd = True
    a_, c_ = ga.flatten(1).double(), gc.flatten(1).double()
    cos_bwd = (a_ * c_).sum(1) / (a_.norm(dim=1) * c_.norm(dim=1))
    print('   fused vs chained BACKWARD of the stride-2 blocks (same masks): gradient cosine %s' % cos_bwd.tolist())
    assert (cos_bwd > 0.999).all(), cos_bwd
    scale = lb.abs().max().item()
    print('fused stride-2 blocks vs chains: logits max diff %.3g of scale %.2f' % ((la - lb).abs().max().item(), scale))
    assert (la - lb).abs().max().item() <= 2e-2 * scale          # the chain rounds the shortcut to bf16 on its own, the fused block does not
    a, b = ga.flatten(1).double(), gb.flatten(1).double()
    cos = (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))
    # a random-init network amplifies the handful of ReLU decisions that differ between the two forwards (see the 'end to end' bound
    # of test_backward_to_input); the kernel itself is pinned by test_fused_stride2_bottleneck_forward_vs_fp64
    print('   gradient cosine fused vs chains: %s' % cos.tolist())
    assert (cos > 0.85).all(), cos


@pytest.mark.parametrize('B,geo', [(1, 56), (3, 56), (1, 28), (2, 28)])
def test_fused_stride2_bottleneck_backward_vs_fp64(B, geo):
    """rart_bottleneck_s2_bwd_bf16: backward-to-input of the stride-2 first block of layer2 / layer3 as one kernel (d_a2 on the 8 x 8
    output grid behind a 14 x 14 input tile, the transposed 3x3 / 2 by input-parity class, the projection shortcut's gradient as extra
    K of class (0, 0)), random masks; against fp64 with bf16 rounding of the two intermediates."""
    from robustart_amd import _lib
    from robustart_amd.model.engine import _Conv
    lib = _lib.load()
    H = W = geo
    CIN, CM, COUT = {56: (256, 128, 512), 28: (512, 256, 1024)}[geo]
    g = torch.Generator().manual_seed(500 + B + geo)

    def mk(cin, cout, k, stride):
        conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5)
        return conv, _Conv(conv, None, 'cuda')

    (c1, ca), (c2, cb), (c3, cc), (c4, ds) = mk(CIN, CM, 1, 1), mk(CM, CM, 3, 2), mk(CM, COUT, 1, 1), mk(CIN, COUT, 1, 2)
    wq = [c.weight.detach().to(torch.bfloat16).double() for c in (c1, c2, c3, c4)]
    rb = lambda t: t.to(torch.bfloat16).double()      # noqa: E731
    sp = _lib.stream_ptr()

    def frag(tab, rows, k, out=None):
        o = torch.empty(rows * k, dtype=torch.bfloat16, device='cuda') if out is None else out
        _lib.check(lib.rart_pack_frag_bf16(_lib.ptr(tab), _lib.ptr(o), rows, k, sp))
        return o

    w3t, w1t, wdt = frag(cc.bwd[0][2], CM, COUT), frag(ca.bwd[0][2], CIN, CM), frag(ds.bwd[0][2], CIN, COUT)
    assert [len(t) for _, t, _ in cb.bwd] == [1, 2, 2, 4] and [p_ for p_, _, _ in cb.bwd] == [(0, 0), (0, 1), (1, 0), (1, 1)]
    w2t = torch.empty(9 * CM * CM, dtype=torch.bfloat16, device='cuda')
    off = 0
    for _, taps, tab in cb.bwd:
        frag(tab, CM, len(taps) * CM, w2t[off:])
        off += len(taps) * CM * CM
    gz = _rand_bf16((B, H // 2, W // 2, COUT), 8).cuda()
    mb = torch.randint(0, 256, (B, H // 2, W // 2, CM // 8), generator=g, dtype=torch.uint8).cuda()
    ma = torch.randint(0, 256, (B, H, W, CM // 8), generator=g, dtype=torch.uint8).cuda()
    mx = torch.randint(0, 256, (B, H, W, CIN // 8), generator=g, dtype=torch.uint8).cuda()
    dx = torch.full((B, H, W, CIN), float('nan'), dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_bottleneck_s2_bwd_bf16(_lib.ptr(gz), _lib.ptr(w3t), _lib.ptr(w2t), _lib.ptr(w1t), _lib.ptr(wdt), _lib.ptr(mb),
                                               _lib.ptr(ma), _lib.ptr(mx), _lib.ptr(dx), B, H, W, CIN, CM, COUT, sp))
    unpack = lambda t: torch.from_numpy(np.unpackbits(t.cpu().numpy(), axis=-1, bitorder='little')).bool()      # noqa: E731
    bits = lambda t: unpack(t).double().permute(0, 3, 1, 2)      # noqa: E731
    gd = gz.cpu().double().permute(0, 3, 1, 2)
    ci = torch.nn.grad.conv2d_input
    d2 = rb(ci((B, CM, H // 2, W // 2), wq[2], gd) * bits(mb))
    d1 = rb(ci((B, CM, H, W), wq[1], d2, stride=2, padding=1) * bits(ma))
    ref = ((ci((B, CIN, H, W), wq[0], d1) + ci((B, CIN, H, W), wq[3], gd, stride=2)) * bits(mx)).permute(0, 2, 3, 1)
    got = dx.cpu().double()
    assert torch.isfinite(got).all()                                   # every input position was written
    err = (got - ref).abs()
    ulp = ref.abs().clamp_min(2.0 ** -20) * 2.0 ** -8 + 1e-6
    frac = (err > ulp).double().mean().item()
    print('fused stride-2 block backward %d x %d: beyond 1 ulp %.2e of the elements, max err %.4f (scale %.2f)'
          % (geo, geo, frac, err.max().item(), ref.abs().max().item()))
    assert frac < 2e-2 and err.max() <= 0.01 * ref.abs().max()
    # without the input mask (the first block of a network has none)
    dx2 = torch.empty_like(dx)
    _lib.check(lib.rart_bottleneck_s2_bwd_bf16(_lib.ptr(gz), _lib.ptr(w3t), _lib.ptr(w2t), _lib.ptr(w1t), _lib.ptr(wdt), _lib.ptr(mb),
                                               _lib.ptr(ma), None, _lib.ptr(dx2), B, H, W, CIN, CM, COUT, sp))
    keep = unpack(mx)
    assert torch.equal(dx2.cpu()[keep], dx.cpu()[keep])


def test_small_m_gemm_for_the_classifier_head():
    """rart_gemm_small_m_bf16 (fc forward: fp32 logits + bias; fc backward: bf16 dpool) against fp64 on the same bf16 operands, ragged
    m and n (rows / columns past the edge are never stored), and the engine switch small_m_fc against the implicit-GEMM head."""
    from robustart_amd import _lib
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import ResNet50Engine
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    for m, n, k, f32 in ((256, 1000, 2048, True), (37, 1000, 2048, True), (256, 2048, 1024, False), (5, 70, 64, False)):
        a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).cuda()
        w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).cuda()
        bias = torch.randn(n, generator=g).cuda() if f32 else None
        out = torch.full((m + 3, n + 8), 7.0, dtype=torch.float32 if f32 else torch.bfloat16, device='cuda')
        _lib.check(lib.rart_gemm_small_m_bf16(_lib.ptr(a), k, _lib.ptr(w), k, _lib.ptr(bias) if f32 else None, _lib.ptr(out), n + 8,
                                              1 if f32 else 0, m, n, k, _lib.stream_ptr()))
        ref = a.double() @ w.double().t() + (bias.double() if f32 else 0.0)
        got = out[:m, :n].double()
        tol = 1e-4 if f32 else 2.0 ** -8
        assert ((got - ref).abs() <= tol * ref.abs() + 1e-3).all(), (m, n, k)
        assert (out[m:] == 7.0).all() and (out[:, n:] == 7.0).all()
    torch.manual_seed(0)
    eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda')
    x = torch.rand(4, 3, 64, 64, generator=g).cuda()
    y = torch.tensor([1, 2, 3, 4]).cuda()
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    l1, _, g1, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    l1, g1 = l1.clone(), g1.clone()
    eng.small_m_fc = False
    l0, _, g0, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    assert (l1 - l0).abs().max() <= 1e-4 * l0.abs().max()             # same bf16 operands, different fp32 summation order
    a_, b_ = g1.flatten().double(), g0.flatten().double()
    assert (a_ @ b_ / (a_.norm() * b_.norm())).item() > 0.9999
