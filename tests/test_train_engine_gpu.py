"""Train-mode HIP kernels and the ResNet-50 train engine against PyTorch (fp32 on the GPU).

The reference trains with PyTorch (cifar10/code/train.py:96-127), so torch's conv2d / BatchNorm2d autograd IS the
reference arithmetic; tolerances reflect bf16 activation storage with fp32 accumulation."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _L():
    from robustart_amd import _lib
    return _lib, _lib.load()


def _ints(v):
    return (ctypes.c_int * len(v))(*v)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize('rows,C,relu,with_res', [(64, 64, 1, 0), (3136, 256, 1, 1), (50176, 64, 0, 0), (777, 2048, 1, 0)])
def test_bn_train_forward_and_backward_kernels(rows, C, relu, with_res):
    L, lib = _L()
    torch.manual_seed(rows + C)
    z = (torch.randn(rows, C, device='cuda') * 1.5 + 0.3).to(torch.bfloat16)
    res = torch.randn(rows, C, device='cuda').to(torch.bfloat16) if with_res else None
    gamma = torch.rand(C, device='cuda') + 0.5
    beta = torch.randn(C, device='cuda') * 0.1
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    y = torch.empty_like(z)
    mean, invstd, ss = (torch.empty(C, device='cuda'), torch.empty(C, device='cuda'), torch.empty(2, C, device='cuda'))
    need = lib.rart_bn_workspace_bytes(rows, C)
    ws = torch.empty(need, dtype=torch.uint8, device='cuda')
    sign = torch.full((rows, C // 8), 0xAA, dtype=torch.uint8, device='cuda')
    L.check(lib.rart_bn_train_forward_bf16(z.data_ptr(), res.data_ptr() if with_res else None, y.data_ptr(), sign.data_ptr(), rows, C,
                                           gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, relu,
                                           mean.data_ptr(), invstd.data_ptr(), ss.data_ptr(), None, 0, ws.data_ptr(), need,
                                           L.stream_ptr()))
    # torch reference on the same bf16-rounded z
    zt = z.float().requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    yt = F.batch_norm(zt, rm2, rv2, gt, bt, training=True, momentum=0.1, eps=1e-5)
    if with_res:
        yt = yt + res.float()
    if relu:
        yt = yt.relu()
    torch.testing.assert_close(mean, zt.detach().mean(0), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm, rm2, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rv, rv2, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(y.float(), yt.detach(), rtol=1e-2, atol=1e-2)          # bf16 output rounding
    # backward: dy arbitrary bf16; mask from the kernel's own y
    dy = torch.randn(rows, C, device='cuda').to(torch.bfloat16)
    dz, g = torch.empty_like(z), torch.empty_like(z)
    dgam, dbet, coef = torch.empty(C, device='cuda'), torch.empty(C, device='cuda'), torch.empty(3, C, device='cuda')
    # the sign tensor is the 1-bit form of (y > 0)
    sh = torch.arange(8, device='cuda', dtype=torch.uint8)
    assert torch.equal(((sign.unsqueeze(-1) >> sh) & 1).reshape(rows, C), (y > 0).to(torch.uint8))
    L.check(lib.rart_bn_train_backward_bf16(dy.data_ptr(), y.data_ptr() if relu else None, 0, z.data_ptr(), dz.data_ptr(),
                                            g.data_ptr(), rows, C, gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                            dgam.data_ptr(), dbet.data_ptr(), 0, coef.data_ptr(), ws.data_ptr(), need,
                                            L.stream_ptr()))
    if relu:      # the same backward with the mask read as bits: bit-identical results
        dz2, g2, dgam2, dbet2 = torch.empty_like(z), torch.empty_like(z), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        L.check(lib.rart_bn_train_backward_bf16(dy.data_ptr(), sign.data_ptr(), 1, z.data_ptr(), dz2.data_ptr(), g2.data_ptr(), rows, C,
                                                gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dgam2.data_ptr(), dbet2.data_ptr(), 0,
                                                coef.data_ptr(), ws.data_ptr(), need, L.stream_ptr()))
        assert torch.equal(dz, dz2) and torch.equal(g, g2) and torch.equal(dgam, dgam2) and torch.equal(dbet, dbet2)
    gmask = dy.float() * ((y.float() > 0).float() if relu else 1.0)
    torch.testing.assert_close(g.float(), gmask, rtol=0, atol=0)
    # torch: BatchNorm backward for upstream gradient gmask
    ybn = F.batch_norm(zt, None, None, gt, bt, training=True, eps=1e-5)
    ybn.backward(gmask)
    assert _cos(dz.float(), zt.grad) > 0.9999
    torch.testing.assert_close(dgam, gt.grad, rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(dbet, bt.grad, rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(dz.float(), zt.grad, rtol=2e-2, atol=2e-2 * float(zt.grad.abs().max()))


@pytest.mark.parametrize('cin,cout,r,stride,H,B', [(64, 64, 3, 1, 16, 4), (64, 256, 1, 1, 10, 3), (256, 128, 1, 1, 28, 9), (128, 128, 3, 2, 16, 4),
                                                   (1024, 2048, 1, 2, 4, 16), (64, 256, 1, 1, 56, 96)])     # the last: 2 352 row tiles -> the folded finaliser
def test_conv_epilogue_batchnorm_statistics(cin, cout, r, stride, H, B):
    """rart_conv_desc.bn_stats_out: the igemm's per-tile column sums / sums of squares of the bf16 output equal the sums of the stored
    tensor (fp32 order noise), for 64- and 128-column tiles, shallow and deep K pipelines and a ragged last row tile; fed to
    rart_bn_train_forward_bf16 as stats_partial they give the statistics of its own pass over z."""
    from robustart_amd.model.train_engine import ResNet50TrainEngine, _TConv
    L, lib = _L()
    model, _, _ = _tiny_resnet_inputs(2, 32)
    eng = ResNet50TrainEngine(model)
    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, r, stride=stride, padding=r // 2, bias=False).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda()
    tc = _TConv(conv, bn, torch.device('cuda'), torch)
    tc.repack(lib, L.stream_ptr())
    x = torch.randn(B, H, H, cin, device='cuda').to(torch.bfloat16)
    oh = H // stride
    z = torch.empty(B, oh, oh, cout, dtype=torch.bfloat16, device='cuda')
    assert eng.conv_bn_stats
    st = eng._conv_fwd(tc, x, (H, H), z)
    rows = B * oh * oh
    part = st[0][:st[1] * 2 * cout * 4].view(torch.float32).view(st[1], 2, cout)
    zf = z.float().view(rows, cout)
    s_ref, q_ref = zf.double().sum(0), (zf.double() ** 2).sum(0)
    assert (part[:, 0].double().sum(0) - s_ref).abs().max().item() <= 1e-4 * (zf.abs().double().sum(0).max().item() + 1)
    assert (part[:, 1].double().sum(0) - q_ref).abs().max().item() <= 1e-5 * q_ref.max().item()
    outs = []
    for use in (st, None):
        y = torch.empty_like(z)
        tc.bn.running_mean.zero_(); tc.bn.running_var.fill_(1)
        eng._bn_fwd(tc, z, y, rows, True, stats=use)
        outs.append((y.clone(), tc.mean.clone(), tc.invstd.clone(), tc.bn.running_var.clone()))
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(outs[0][3], outs[1][3], rtol=1e-5, atol=1e-6)
    assert (outs[0][0].float() - outs[1][0].float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize('B,H,C,r,stride', [(3, 8, 64, 3, 1), (2, 12, 128, 3, 2), (5, 7, 256, 1, 1), (2, 8, 512, 1, 2)])
def test_transpose_gather_matches_unfold(B, H, C, r, stride):
    L, lib = _L()
    torch.manual_seed(B * H + C)
    x = torch.randn(B, H, H, C, device='cuda').to(torch.bfloat16)
    pad = r // 2
    oh = H // stride
    taps = [(a - pad, b - pad) for a in range(r) for b in range(r)]
    M = B * oh * oh
    m_pad = (M + 63) // 64 * 64 + 64
    out = torch.full((len(taps) * C, m_pad), 7.0, device='cuda', dtype=torch.bfloat16)
    L.check(lib.rart_transpose_gather_bf16(x.data_ptr(), out.data_ptr(), B, H, H, C, oh, oh, stride, stride, len(taps),
                                           _ints([t[0] for t in taps]), _ints([t[1] for t in taps]), m_pad, 0, 0, L.stream_ptr()))
    cols = F.unfold(x.float().permute(0, 3, 1, 2), r, padding=pad, stride=stride)     # [B][C*r*r][oh*ow], (c, tap) order
    cols = cols.view(B, C, r * r, oh * oh).permute(2, 1, 0, 3).reshape(r * r * C, M)  # -> (tap, c) rows, m = (b, pix)
    assert torch.equal(out[:, :M].float(), cols)
    assert torch.count_nonzero(out[:, M:]).item() == 0
    # slab layout: [m_pad / chunk][rows_total][chunk] with extra (untouched) padding rows per slab
    chunk, rows_total = 64, len(taps) * C + 8
    slabs = torch.full((m_pad // chunk, rows_total, chunk), 7.0, device='cuda', dtype=torch.bfloat16)
    L.check(lib.rart_transpose_gather_bf16(x.data_ptr(), slabs.data_ptr(), B, H, H, C, oh, oh, stride, stride, len(taps),
                                           _ints([t[0] for t in taps]), _ints([t[1] for t in taps]), m_pad, chunk,
                                           rows_total, L.stream_ptr()))
    assert torch.equal(slabs[:, :len(taps) * C, :].permute(1, 0, 2).reshape(len(taps) * C, m_pad), out)
    assert bool((slabs[:, len(taps) * C:, :] == 7.0).all())


def test_transpose_gather_c4_and_pack_weight():
    L, lib = _L()
    torch.manual_seed(1)
    B, H = 2, 16
    x = torch.randn(B, H + 8, H + 8, 4, device='cuda').to(torch.bfloat16)
    taps = [(a, b) for a in range(7) for b in range(7)]
    oh = H // 2
    M = B * oh * oh
    m_pad = (M + 63) // 64 * 64
    out = torch.empty(49 * 4, m_pad, device='cuda', dtype=torch.bfloat16)
    L.check(lib.rart_transpose_gather_bf16(x.data_ptr(), out.data_ptr(), B, H + 8, H + 8, 4, oh, oh, 2, 2, 49,
                                           _ints([t[0] for t in taps]), _ints([t[1] for t in taps]), m_pad, 0, 0, L.stream_ptr()))
    ref = torch.empty(49, 4, B, oh, oh, device='cuda')
    for ti, (a, b) in enumerate(taps):
        ref[ti] = x.float()[:, a:a + 2 * oh:2, b:b + 2 * oh:2, :].permute(3, 0, 1, 2)
    assert torch.equal(out[:, :M].float(), ref.reshape(196, M))
    # weight packing against the eval engine's host-side layout
    from robustart_amd.model.engine import _Conv
    for cin, cout, r, stride in [(64, 64, 3, 1), (128, 256, 3, 2), (256, 64, 1, 1), (256, 512, 1, 2)]:
        conv = torch.nn.Conv2d(cin, cout, r, stride=stride, padding=r // 2, bias=False).cuda()
        ref_c = _Conv(conv, None, 'cuda')
        from robustart_amd.model.train_engine import _TConv
        tc = _TConv(conv, None, torch.device('cuda'), torch)
        tc.repack(lib, L.stream_ptr())
        assert torch.equal(tc.w_fwd, ref_c.w_fwd)
        for (p1, t1, _, w1), (p2, t2, w2) in zip(tc.bwd, ref_c.bwd):
            assert p1 == p2 and t1 == t2
            assert (w1 is None and w2 is None) or torch.equal(w1, w2)


def _tiny_resnet_inputs(B, S):
    torch.manual_seed(0)
    from robustart_amd.model import get_model
    model = get_model({'type': 'resnet50_official'}).cuda().train()
    x01 = torch.rand(B, 3, S, S, device='cuda')
    y = torch.randint(0, 1000, (B,), device='cuda')
    return model, x01, y


@pytest.mark.parametrize('direct', [True, False])
def test_wgrad_single_layers_match_torch(direct):
    """Weight gradient of single convs (3x3/1, 3x3/2, 1x1/1, 1x1/2) vs torch autograd, on both paths: rart_wgrad_direct_bf16 (tiles
    position-major in LDS, fragments through ds_read_b64_tr_b16; two taps per tile for 64 channels, 64- and 128-column tiles, position
    counts that are not a multiple of the 32-position K step, several K splits) and the transpose_gather + implicit-GEMM path it replaces."""
    from robustart_amd.model.train_engine import ResNet50TrainEngine, _TConv
    model, _, _ = _tiny_resnet_inputs(2, 32)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    eng = ResNet50TrainEngine(model)
    eng.direct_wgrad = direct
    torch.manual_seed(5)
    for cin, cout, r, stride, H, B in [(64, 64, 3, 1, 16, 4), (128, 128, 3, 2, 16, 4), (256, 64, 1, 1, 8, 8),
                                       (256, 512, 1, 2, 8, 8), (512, 2048, 1, 1, 4, 16), (64, 256, 1, 1, 10, 3), (64, 64, 1, 1, 10, 3),
                                       (128, 128, 3, 1, 14, 5), (64, 64, 3, 1, 56, 6), (1024, 256, 1, 1, 14, 4), (256, 256, 3, 2, 14, 3)]:
        conv = torch.nn.Conv2d(cin, cout, r, stride=stride, padding=r // 2, bias=False).cuda()
        conv.weight.grad = torch.zeros_like(conv.weight)
        tc = _TConv(conv, None, torch.device('cuda'), torch)
        x = torch.randn(B, H, H, cin, device='cuda').to(torch.bfloat16)
        oh = H // stride
        dz = torch.randn(B, oh, oh, cout, device='cuda').to(torch.bfloat16)
        eng._conv_wgrad(tc, dz, (oh, oh), x, (H, H))
        w = conv.weight.detach().clone().requires_grad_(True)
        F.conv2d(x.float().permute(0, 3, 1, 2), w, stride=stride, padding=r // 2).backward(dz.float().permute(0, 3, 1, 2))
        assert _cos(conv.weight.grad, w.grad) > 0.99999, (cin, cout, r, stride, H, B)
        torch.testing.assert_close(conv.weight.grad, w.grad, rtol=1e-3, atol=1e-3 * float(w.grad.abs().max()))


def test_stem_wgrad_direct_matches_the_transposing_path_and_torch():
    """The stem's weight gradient (7x7 / 2, 49 taps on the 4-channel padded plane, 3 valid channels) on rart_wgrad_direct_bf16's
    4-channel form against the transpose_gather path and torch, through a full engine step at 64 x 64 and 96 x 128 inputs."""
    from robustart_amd.model.train_engine import ResNet50TrainEngine
    for S_ in ((64, 64), (96, 128)):
        model, _, _ = _tiny_resnet_inputs(2, 32)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        eng = ResNet50TrainEngine(model)
        g = torch.Generator().manual_seed(3)
        x01 = torch.rand(3, 3, S_[0], S_[1], generator=g).cuda()
        dl = (torch.randn(3, 1000, generator=g) * 1e-2).cuda()
        grads = []
        for direct in (True, False):
            eng.direct_wgrad = direct
            eng.forward(x01, False, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
            eng.backward(dl)
            grads.append(model.conv1.weight.grad.clone())
        assert _cos(grads[0], grads[1]) > 0.99999
        torch.testing.assert_close(grads[0], grads[1], rtol=2e-3, atol=2e-3 * float(grads[1].abs().max()))


def test_wgrad_direct_at_the_benchmark_size():
    """rart_wgrad_direct_bf16 at B = 256 on layer1's 3x3 (802 816 positions, 1 020 K splits, two taps per tile) and layer3's 1x1
    (1 024 -> 256): against torch's fp32 weight gradient of the same bf16 tensors."""
    from robustart_amd.model.train_engine import ResNet50TrainEngine, _TConv
    model, _, _ = _tiny_resnet_inputs(2, 32)
    eng = ResNet50TrainEngine(model)
    assert eng.direct_wgrad
    torch.manual_seed(7)
    for cin, cout, r, H in ((64, 64, 3, 56), (1024, 256, 1, 14)):
        conv = torch.nn.Conv2d(cin, cout, r, padding=r // 2, bias=False).cuda()
        conv.weight.grad = torch.zeros_like(conv.weight)
        tc = _TConv(conv, None, torch.device('cuda'), torch)
        x = torch.randn(256, H, H, cin, device='cuda').to(torch.bfloat16)
        dz = torch.randn(256, H, H, cout, device='cuda').to(torch.bfloat16)
        eng._conv_wgrad(tc, dz, (H, H), x, (H, H))
        want = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), conv.weight.shape, dz.float().permute(0, 3, 1, 2), padding=r // 2)
        assert _cos(conv.weight.grad, want) > 0.999999, (cin, cout, r)
        torch.testing.assert_close(conv.weight.grad, want, rtol=2e-3, atol=2e-3 * float(want.abs().max()))


def _rb(t):
    """bf16 rounding with a straight-through gradient: the storage points of the engine."""
    return t + (t.to(torch.bfloat16).float() - t).detach()


def _nchw(t):
    return t.float().permute(0, 3, 1, 2)


def _forced_forward(ref, xn, acts):
    """The engine's train-mode forward restated with torch fp32 ops, bf16 rounding at the engine's storage points, and
    every stage's VALUE pinned to what the engine stored (straight-through), so torch autograd differentiates the same
    function at the same point: identical ReLU masks, batch statistics and max-pool winners.  Without the pinning a
    random-init train-mode ResNet-50 amplifies single-ulp bf16 differences chaotically (logits agree only to cos 0.98
    at this size although every stage matches to 1e-6 when teacher-forced), which would hide real backward bugs."""
    stage_err = []

    def pin(t, stored):
        v = _nchw(stored) if stored.dim() == 4 else stored.float()
        stage_err.append(float((t.detach() - v).norm() / (v.norm() + 1e-30)))
        return t + (v - t).detach()

    def conv(x, m):
        return _rb(F.conv2d(x, _rb(m.weight), stride=m.stride, padding=m.padding))

    def bn(z, m, relu, res=None):
        y = F.batch_norm(z, None, None, m.weight, m.bias, training=True, eps=m.eps)
        if res is not None:
            y = y + res
        return _rb(y.relu() if relu else y)

    z1 = pin(conv(xn, ref.conv1), acts['z1'])
    y1 = pin(bn(z1, ref.bn1, True), acts['y1'])
    x = pin(F.max_pool2d(y1, 3, 2, 1), acts['p1'])
    k = 0
    for layer in (ref.layer1, ref.layer2, ref.layer3, ref.layer4):
        for blk in layer:
            _, _, za, ya, zb, yb, zc, zd, out, _ = acts['b%d' % k]
            a = pin(bn(pin(conv(x, blk.conv1), za), blk.bn1, True), ya)
            b = pin(bn(pin(conv(a, blk.conv2), zb), blk.bn2, True), yb)
            if blk.downsample is not None:
                sk = bn(pin(conv(x, blk.downsample[0]), zd), blk.downsample[1], False)
            else:
                sk = x
            x = pin(bn(pin(conv(b, blk.conv3), zc), blk.bn3, True, res=sk), out)
            k += 1
    pooled = pin(_rb(x.mean((2, 3))), acts['pooled'])
    return F.linear(pooled, _rb(ref.fc.weight), ref.fc.bias), max(stage_err)


def test_resnet50_train_engine_matches_torch_autograd():
    from robustart_amd.model.train_engine import ResNet50TrainEngine
    from robustart_amd.train.arena import label_smooth_ce
    import copy
    B, S = 16, 64
    model, x01, y = _tiny_resnet_inputs(B, S)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    ref = copy.deepcopy(model)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    ready = []
    eng = ResNet50TrainEngine(model, on_grad_ready=lambda p: ready.append(id(p)))
    logits = eng.forward(x01, False, mean, std)
    loss_rows, dl = label_smooth_ce(logits, y, 0.1, 1.0 / B)
    eng.backward(dl)
    assert sorted(ready) == sorted(id(p) for p in model.parameters())         # every gradient announced exactly once
    mt = torch.tensor(mean, device='cuda').view(1, 3, 1, 1)
    st = torch.tensor(std, device='cuda').view(1, 3, 1, 1)
    out, worst_stage = _forced_forward(ref, (x01 - mt) / st, eng.acts)
    print('largest relative stage error (teacher-forced forward): %.3g' % worst_stage)
    assert worst_stage < 2e-3                                                  # each stage reproduces the engine's value
    loss = F.cross_entropy(out, y, label_smoothing=0.1)
    loss.backward()
    assert _cos(logits, out.detach()) > 0.99999
    assert abs(float(loss_rows.mean()) - float(loss.detach())) < 1e-3
    # running statistics against plain torch BatchNorm on the fp32 model (first layer: no accumulated rounding)
    plain = copy.deepcopy(ref)
    plain((x01 - mt) / st)
    torch.testing.assert_close(model.bn1.running_mean, plain.bn1.running_mean, rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(model.bn1.running_var, plain.bn1.running_var, rtol=2e-2, atol=2e-3)
    assert int(model.bn1.num_batches_tracked) == 1
    rep = sorted((_cos(p.grad, q.grad), n) for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()))
    cs = np.array([c for c, _ in rep])
    print('gradient cos vs torch autograd at the same point: median %.5f mean %.5f; lowest: %s' %
          (np.median(cs), cs.mean(), rep[:5]))
    # the engine stores every intermediate gradient in bf16 (2^-9 relative per stage, ~100 stages deep)
    assert np.median(cs) > 0.999 and cs.min() > 0.99, rep[:8]
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        ratio = float(p.grad.norm() / (q.grad.norm() + 1e-30))
        assert 0.97 < ratio < 1.03, (n, ratio)
