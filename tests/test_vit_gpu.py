"""GPU parity of the ViT-B/16 forward engine vs the plain PyTorch module (fp32) on the same random weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


@pytest.fixture(scope='module')
def setup():
    from robustart_amd.model import get_model
    from robustart_amd.model.vit_engine import ViTEngine
    torch.manual_seed(0)
    m = get_model({'type': 'vit_base', 'kwargs': {'num_classes': 1000, 'drop_path_rate': 0.1}}).eval()
    # non-trivial biases / norms so that every epilogue term is exercised
    g = torch.Generator().manual_seed(1)
    for n, p in m.named_parameters():
        if n.endswith('bias'):
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.05)
        if 'norm' in n and n.endswith('weight'):
            p.data.copy_(1 + torch.randn(p.shape, generator=g) * 0.1)
    for p in m.parameters():
        p.requires_grad_(False)
    return m.cuda(), ViTEngine(m, 'cuda')


def test_layernorm_softmax_kernels():
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(37, 768, generator=g) * 2 + 0.3).to(torch.bfloat16).cuda()
    gam, bet = torch.randn(768, generator=g).cuda(), torch.randn(768, generator=g).cuda()
    out = torch.empty_like(x)
    _lib.check(lib.rart_layernorm_bf16(_lib.ptr(x), _lib.ptr(gam), _lib.ptr(bet), _lib.ptr(out), 37, 768, 768, 768, 1e-6,
                                       _lib.stream_ptr()))
    ref = torch.nn.functional.layer_norm(x.float(), (768,), gam, bet, 1e-6)
    torch.testing.assert_close(out.float(), ref, atol=3e-2, rtol=1e-2)          # one bf16 rounding of the result
    s = (torch.randn(50, 200, generator=g) * 8).to(torch.bfloat16).cuda()
    p = torch.empty(50, 224, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_softmax_rows_bf16(_lib.ptr(s), _lib.ptr(p), 50, 197, 200, 224, 0.125, _lib.stream_ptr()))
    ref = torch.softmax(s[:, :197].float() * 0.125, -1)
    torch.testing.assert_close(p[:, :197].float(), ref, atol=4e-3, rtol=1e-2)
    assert (p[:, 197:] == 0).all()


@pytest.mark.parametrize('B', [3, 8])
def test_vit_forward_logits(setup, B):
    m, eng = setup
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 3, 224, 224, generator=g).cuda()
    got = eng.logits(x, MEAN, STD)
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    ref = m((x - mean) / std)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    print('vit forward B=%d: logit scale %.3f, max |err| vs fp32 torch %.4f (%.2f%%)' % (B, scale, err, 100 * err / scale))
    assert err < 0.03 * scale                     # bf16 storage through 12 blocks
    assert (got.argmax(1) == ref.argmax(1)).float().mean() >= 0.6
    u8 = (x * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    a = eng.logits_from_u8(u8, MEAN, STD)
    b = eng.logits(u8.permute(0, 3, 1, 2).float() / 255, MEAN, STD)
    torch.testing.assert_close(a, b, atol=0.02 * scale, rtol=0)


def test_fused_attention_matches_torch_and_unfused_path(setup):
    """rart_vit_attention vs torch softmax attention on the same bf16 qkv, and vs the batched-igemm decomposition."""
    from robustart_amd import _lib
    m, eng = setup
    lib = _lib.load()
    B, T, H, hd = 3, 197, 12, 64
    D = H * hd
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * T + 256, 3 * D, generator=g) * 1.5).to(torch.bfloat16).cuda()
    qkv[B * T:] = 0
    out = torch.empty(B, T, D, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_vit_attention(_lib.ptr(qkv), _lib.ptr(out), B, T, H, hd, _lib.stream_ptr()))
    q3 = qkv[:B * T].float().view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q3[0] @ q3[1].transpose(-2, -1) * hd ** -0.5, -1) @ q3[2]).transpose(1, 2).reshape(B, T, D)
    err = (out.float() - ref).abs().max().item()
    print('fused attention max err %.4g (ref scale %.3f)' % (err, ref.abs().max().item()))
    assert err < 0.02 * ref.abs().max().item() + 2e-2
    s_ld, t_pad = 200, 224
    scores = torch.empty(B * H, T, s_ld, dtype=torch.bfloat16, device='cuda')
    probs = torch.empty(B * H, T, t_pad, dtype=torch.bfloat16, device='cuda')
    vt = torch.zeros(B * H * hd + 128, t_pad, dtype=torch.bfloat16, device='cuda')
    out2 = torch.empty_like(out)
    eng._attention_unfused(qkv, scores, probs, vt, out2, B, T, s_ld, t_pad)
    assert (out2.float() - ref).abs().max().item() < 0.03 * ref.abs().max().item() + 2e-2


@pytest.mark.parametrize('T', [1, 17, 32, 33, 65, 96, 128, 129, 160, 192, 193, 224])
def test_fused_attention_every_key_tile_count(T):
    """rart_vit_attention instantiates k_vit_attention for ceil(T / 32) key tiles and masks only the last one: token counts on both
    sides of every tile boundary (and a full last tile, where nothing is masked) against torch's soft-max attention."""
    from robustart_amd import _lib
    lib = _lib.load()
    B, H, hd = 2, 3, 64
    D = H * hd
    g = torch.Generator().manual_seed(100 + T)
    qkv = (torch.randn(B * T + 256, 3 * D, generator=g) * 1.5).to(torch.bfloat16).cuda()
    qkv[B * T:] = 7.0                                     # rows past the batch must never be read into a result
    out = torch.full((B * T + 8, D), 3.0, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_vit_attention(_lib.ptr(qkv), _lib.ptr(out), B, T, H, hd, _lib.stream_ptr()))
    q3 = qkv[:B * T].float().view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q3[0] @ q3[1].transpose(-2, -1) * hd ** -0.5, -1) @ q3[2]).transpose(1, 2).reshape(B * T, D)
    err = (out[:B * T].float() - ref).abs().max().item()
    assert err < 0.02 * ref.abs().max().item() + 2e-2, (T, err)
    assert (out[B * T:] == 3.0).all()                     # nothing written past the last token


def test_vit_backward_kernels():
    """GELU, GELU', LayerNorm backward, soft-max backward rows, un-patchify vs torch."""
    from robustart_amd import _lib
    import ctypes
    lib = _lib.load()
    g = torch.Generator().manual_seed(2)
    u = (torch.randn(64, 3072, generator=g) * 1.5).to(torch.bfloat16).cuda()
    dh = torch.randn(64, 3072, generator=g).to(torch.bfloat16).cuda()
    out, du = torch.empty_like(u), torch.empty_like(u)
    _lib.check(lib.rart_gelu_bf16(_lib.ptr(u), _lib.ptr(out), u.numel(), _lib.stream_ptr()))
    _lib.check(lib.rart_gelu_bwd_bf16(_lib.ptr(dh), _lib.ptr(u), _lib.ptr(du), u.numel(), _lib.stream_ptr()))
    ut = u.float().requires_grad_(True)
    yt = torch.nn.functional.gelu(ut)
    yt.backward(dh.float())
    torch.testing.assert_close(out.float(), yt.detach(), atol=2e-2, rtol=1e-2)
    torch.testing.assert_close(du.float(), ut.grad, atol=2e-2, rtol=1e-2)
    # LayerNorm backward with residual
    x = (torch.randn(37, 768, generator=g) * 2 + 0.3).to(torch.bfloat16).cuda()
    dy = torch.randn(37, 768, generator=g).to(torch.bfloat16).cuda()
    res = torch.randn(37, 768, generator=g).to(torch.bfloat16).cuda()
    gam = (1 + 0.1 * torch.randn(768, generator=g)).cuda()
    dx = torch.empty_like(x)
    _lib.check(lib.rart_layernorm_bwd_bf16(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(gam), _lib.ptr(res), _lib.ptr(dx), 37, 768, 768,
                                           768, 768, 768, 1e-6, _lib.stream_ptr()))
    xt = x.float().requires_grad_(True)
    torch.nn.functional.layer_norm(xt, (768,), gam, torch.zeros(768, device='cuda'), 1e-6).backward(dy.float())
    torch.testing.assert_close(dx.float(), xt.grad + res.float(), atol=3e-2, rtol=2e-2)
    # soft-max backward rows
    s = (torch.randn(50, 197, generator=g) * 3).cuda()
    p = torch.softmax(s * 0.125, 1)
    pb = torch.zeros(50, 224, dtype=torch.bfloat16, device='cuda')
    pb[:, :197] = p.to(torch.bfloat16)
    dp = torch.zeros(50, 200, dtype=torch.bfloat16, device='cuda')
    dp[:, :197] = torch.randn(50, 197, generator=g).to(torch.bfloat16).cuda()
    ds = torch.full((50, 224), 9.0, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_softmax_bwd_rows_bf16(_lib.ptr(pb), _lib.ptr(dp), _lib.ptr(ds), 50, 197, 224, 200, 224, 0.125,
                                              _lib.stream_ptr()))
    pf, df = pb[:, :197].float(), dp[:, :197].float()
    ref = 0.125 * pf * (df - (pf * df).sum(1, keepdim=True))
    torch.testing.assert_close(ds[:, :197].float(), ref, atol=2e-3, rtol=2e-2)
    assert torch.count_nonzero(ds[:, 197:]).item() == 0
    # un-patchify: inverse layout of rart_vit_patchify, scaled by 1/std
    B, ps, H = 2, 16, 64
    P, kk = (H // ps) ** 2, 3 * ps * ps
    dpat = torch.randn(B * P, kk, generator=g).to(torch.bfloat16).cuda()
    grad = torch.empty(B, 3, H, H, device='cuda')
    _lib.check(lib.rart_vit_unpatchify_f32(_lib.ptr(dpat), _lib.ptr(grad), B, H, H, ps, kk, (ctypes.c_float * 3)(*STD),
                                           _lib.stream_ptr()))
    ref = dpat.float().view(B, H // ps, H // ps, 3, ps, ps).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, H, H)
    ref = ref / torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    torch.testing.assert_close(grad, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('kind', [0, 1])
def test_vit_backward_to_input_matches_torch_autograd(setup, kind):
    """forward_backward (every GEMM and attention product on the igemm kernel) vs torch autograd through the fp32 module."""
    from robustart_amd.noise.adv import logit_loss
    m, eng = setup
    torch.manual_seed(5)
    B = 4
    x01 = torch.rand(B, 3, 224, 224, device='cuda')
    y = torch.randint(0, 1000, (B,), device='cuda')
    logits, loss, grad, pred = eng.forward_backward(x01, MEAN, STD, y, kind)
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    xr = x01.clone().requires_grad_(True)
    out = m((xr - mean) / std)
    # same upstream gradient on both sides: the DLR loss picks the top-3 logits, and on random-init weights the
    # engine's bf16 logits (within 2 % of the scale) order near-ties differently from fp32
    g_ref, = torch.autograd.grad(out, xr, grad_outputs=eng.last_dlogits)
    if kind == 0:
        l_ref, _, _ = logit_loss(out, y, kind, None, 1.0)
        torch.testing.assert_close(loss, l_ref, rtol=2e-2, atol=2e-2)
    scale = out.detach().abs().max()
    assert (logits - out.detach()).abs().max() < 0.02 * scale
    assert pred.tolist() == out.argmax(1).tolist()
    a, b = grad.double().flatten(), g_ref.double().flatten()
    cos = float((a @ b) / (a.norm() * b.norm()))
    rel = float((a - b).norm() / b.norm())
    print('ViT grad cos %.5f rel %.4f' % (cos, rel))
    assert cos > 0.995 and rel < 0.1
    for i in range(B):                                      # per-sample directions too
        ai, bi = grad[i].double().flatten(), g_ref[i].double().flatten()
        assert float((ai @ bi) / (ai.norm() * bi.norm())) > 0.99


def test_pgd_on_vit_runs_on_the_hip_engine(setup):
    from robustart_amd.model.engine import EngineModel
    from robustart_amd.noise import AddNoise
    m, eng = setup
    torch.manual_seed(6)
    x01 = torch.rand(3, 3, 224, 224, device='cuda')
    y = torch.randint(0, 1000, (3,), device='cuda')
    f_model = EngineModel(None, takes_normalized=False, engine=eng)
    an = AddNoise('pgd_linf')
    an.set_config(f_model=f_model, eps=4 / 255, steps=3)
    adv_x = an.add_noise(x01, y)
    assert (adv_x - x01).abs().max() <= 4 / 255 + 1e-6 and adv_x.min() >= 0 and adv_x.max() <= 1
    clean = torch.nn.functional.cross_entropy(eng.logits(x01, MEAN, STD), y)
    attacked = torch.nn.functional.cross_entropy(eng.logits(adv_x, MEAN, STD), y)
    assert attacked > clean                                  # the gradient points uphill


def test_vit_refold_tracks_new_weights(setup):
    """ViTEngine.refold (GPU-side packing from live parameters) == a freshly constructed engine, bit for bit."""
    import copy
    from robustart_amd.model.vit_engine import ViTEngine
    m, _ = setup
    m2 = copy.deepcopy(m)
    eng = ViTEngine(m2, 'cuda')
    torch.manual_seed(9)
    x = torch.rand(2, 3, 224, 224, device='cuda')
    with torch.no_grad():
        for p in m2.parameters():
            p.mul_(1.0 + 0.05 * torch.randn_like(p))
    eng.refold(m2)
    fresh = ViTEngine(m2, 'cuda')
    assert torch.equal(eng.logits(x, MEAN, STD), fresh.logits(x, MEAN, STD))
    for a, b in zip(eng.layers, fresh.layers):
        for k in a:
            assert (a[k] == b[k]) if isinstance(a[k], int) else torch.equal(a[k], b[k])


def test_fused_attention_backward_matches_unfused_and_torch(setup):
    """rart_vit_attention_bwd vs torch autograd of softmax(QK^T/sqrt(d))V on the same bf16 q, k, v, and vs the
    batched-GEMM decomposition through the whole network."""
    from robustart_amd import _lib
    lib = _lib.load()
    torch.manual_seed(11)
    B, T, H, hd = 3, 197, 12, 64
    D = H * hd
    qkv = (torch.randn(B * T + 256, 3 * D, device='cuda') * 0.7).to(torch.bfloat16)
    dout = torch.randn(B * T, D, device='cuda').to(torch.bfloat16)
    q, k, v = [qkv[:B * T].float().view(B, T, 3, H, hd)[:, :, i].permute(0, 2, 1, 3).requires_grad_(True) for i in range(3)]
    p = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1)
    o = p @ v                                                          # [B][H][T][hd]
    o.backward(dout.float().view(B, T, H, hd).permute(0, 2, 1, 3))
    att = o.detach().permute(0, 2, 1, 3).reshape(B * T, D).to(torch.bfloat16).contiguous()
    dqkv = torch.zeros(B * T, 3 * D, device='cuda', dtype=torch.bfloat16)
    _lib.check(lib.rart_vit_attention_bwd(_lib.ptr(qkv), _lib.ptr(att), _lib.ptr(dout), _lib.ptr(dqkv), B, T, H, hd,
                                          _lib.stream_ptr()))
    got = dqkv.float().view(B, T, 3, H, hd)
    for i, ref in enumerate((q.grad, k.grad, v.grad)):
        g = got[:, :, i].permute(0, 2, 1, 3)
        a, b2 = g.double().flatten(), ref.double().flatten()
        cos = float((a @ b2) / (a.norm() * b2.norm()))
        rel = float((a - b2).norm() / b2.norm())
        assert cos > 0.9995 and rel < 0.03, (i, cos, rel)
    # whole network: fused vs unfused backward give the same input gradient (up to bf16 rounding of dS / P)
    m, eng = setup
    x01 = torch.rand(2, 3, 224, 224, device='cuda')
    y = torch.randint(0, 1000, (2,), device='cuda')
    eng.fused_attention_bwd = True
    g1 = eng.forward_backward(x01, MEAN, STD, y, 0)[2]
    eng.fused_attention_bwd = False
    g2 = eng.forward_backward(x01, MEAN, STD, y, 0)[2]
    eng.fused_attention_bwd = True
    a, b2 = g1.double().flatten(), g2.double().flatten()
    assert float((a @ b2) / (a.norm() * b2.norm())) > 0.9995


def test_vit_train_engine_parameter_gradients_match_torch_autograd():
    """ViTTrainEngine (forward + backward to all 152 parameters on HIP) vs torch autograd through the fp32 module."""
    import copy
    from robustart_amd import _lib
    from robustart_amd.model import get_model
    from robustart_amd.model.vit_train_engine import ViTTrainEngine
    from robustart_amd.train.arena import label_smooth_ce
    lib = _lib.load()
    torch.manual_seed(3)
    model = get_model({'type': 'vit_base', 'kwargs': {'num_classes': 1000, 'drop_path_rate': 0.0}}).cuda().train()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bias'):
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).cuda())
            if 'norm' in n and n.endswith('weight'):
                p.copy_((1 + torch.randn(p.shape, generator=g) * 0.1).cuda())
    ref = copy.deepcopy(model)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    ready = []
    eng = ViTTrainEngine(model, 'cuda', on_grad_ready=lambda p: ready.append(id(p)))
    B = 4
    x01 = torch.rand(B, 3, 224, 224, device='cuda')
    y = torch.randint(0, 1000, (B,), device='cuda')
    logits = eng.forward(x01, False, MEAN, STD)
    loss_rows, dl = label_smooth_ce(logits, y, 0.1, 1.0 / B)
    eng.backward(dl)
    assert sorted(ready) == sorted(id(p) for p in model.parameters())
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    out = ref((x01 - mean) / std)
    out.backward(dl)                                              # the same upstream gradient on both sides
    assert (logits - out.detach()).abs().max() < 0.02 * out.detach().abs().max()
    rep = []
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        a, b2 = p.grad.double().flatten(), q.grad.double().flatten()
        rep.append((float((a @ b2) / (a.norm() * b2.norm() + 1e-30)), float(a.norm() / (b2.norm() + 1e-30)), n))
    rep.sort()
    print('lowest parameter-gradient cosines:', [(round(c, 4), round(r, 3), n) for c, r, n in rep[:6]])
    import numpy as np
    cs = np.array([c for c, _, _ in rep])
    assert np.median(cs) > 0.999 and cs.min() > 0.98, rep[:8]
    assert all(0.9 < r < 1.1 for _, r, _ in rep), [x for x in rep if not 0.9 < x[1] < 1.1][:8]


def test_gemm256_gelu_keep_equals_the_two_launch_form():
    """Flag 64 of the 256 x 256 GEMM (dst = gelu(u), `mask` receives the bf16 pre-activation u: ViT fc1 when the backward will need u)
    against the plain product followed by rart_gelu_bf16: u bit-identical, gelu(u) within bf16 rounding; the 128 x 128 kernel refuses the flag and
    rart_gemm256_supported tells the engine which launches qualify."""
    import ctypes
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    M, K, N = 256 * 32 + 37, 192, 4096
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()

    def run(flags, keep_out):
        out = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
        d = _lib.ConvDesc()
        d.src, d.wgt, d.dst = a.data_ptr(), w.data_ptr(), out.data_ptr()
        d.bias = bias.data_ptr()
        d.mask = keep_out.data_ptr() if keep_out is not None else None
        d.batch, d.grid_h, d.grid_w = 1, M, 1
        d.src_h, d.src_w, d.src_pix_stride = M, 1, K
        d.k_per_tap, d.n_taps, d.sy, d.sx = K, 1, 1, 1
        d.n_cols, d.dst_h, d.dst_w = N, M, 1
        d.dst_sy, d.dst_sx, d.dst_oy, d.dst_ox, d.dst_pix_stride = 1, 1, 0, 0, N
        d.flags = flags
        st = lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr())
        torch.cuda.synchronize()
        return st, out

    assert lib.rart_gemm256_supported(M, K, N, K, N) == 1 and lib.rart_gemm256_supported(64, K, N, K, N) == 0
    st, u_ref = run(0, None)
    assert st == 0
    hid_ref = torch.empty_like(u_ref)
    _lib.check(lib.rart_gelu_bf16(_lib.ptr(u_ref), _lib.ptr(hid_ref), u_ref.numel(), _lib.stream_ptr()))
    u = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
    st, hid = run(64, u)
    assert st == 0 and torch.equal(u, u_ref)              # the pre-activation is the plain product's output, bit for bit
    # gelu of the ROUNDED u: within bf16 rounding of the exact function, and within one bf16 step of the stand-alone kernel
    # (which evaluates erff; the GEMM epilogues use the 1.5e-7 Abramowitz-Stegun form shared with flag 4)
    exact = torch.nn.functional.gelu(u_ref.double())
    assert ((hid.double() - exact).abs() <= exact.abs() * 2.0 ** -8 + 1e-6).all()
    assert ((hid.double() - hid_ref.double()).abs() <= hid_ref.double().abs() * 2.0 ** -7 + 1e-6).all()
    try:
        lib.rart_igemm_set_gemm256(0)
        st, _ = run(64, u)
        assert st == 2 and b'flag 64' in lib.rart_last_error_string()
    finally:
        lib.rart_igemm_set_gemm256(2)


@pytest.mark.parametrize('flags,use_res', [(0, False), (0, True), (4, False), (8, False)])
def test_gemm256_kernel_vs_torch_and_the_128_kernel(flags, use_res):
    """The 256 x 256 x 64 direct-to-LDS GEMM (k_gemm256_bf16, taken by plain products with >= 512 tiles) against fp64 and
    against the 128 x 128 kernel on the same descriptor: ragged M (rows past M come from the zero page and are never stored),
    bias, residual, GELU, GELU' epilogues.  Same operands, same K order, same epilogue arithmetic: bit-identical outputs."""
    import ctypes
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(flags + 3)
    M, K, N = 256 * 32 + 37, 192, 4096
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if use_res else None
    u = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if flags == 8 else None

    def run(enable):
        lib.rart_igemm_set_gemm256(enable)
        out = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
        d = _lib.ConvDesc()
        d.src, d.wgt, d.dst = a.data_ptr(), w.data_ptr(), out.data_ptr()
        d.bias = bias.data_ptr()
        d.res = res.data_ptr() if res is not None else None
        d.mask = u.data_ptr() if u is not None else None
        d.batch, d.grid_h, d.grid_w = 1, M, 1
        d.src_h, d.src_w, d.src_pix_stride = M, 1, K
        d.k_per_tap, d.n_taps, d.sy, d.sx = K, 1, 1, 1
        d.n_cols, d.dst_h, d.dst_w = N, M, 1
        d.dst_sy, d.dst_sx, d.dst_oy, d.dst_ox, d.dst_pix_stride = 1, 1, 0, 0, N
        d.flags = flags
        _lib.check(lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr()))
        torch.cuda.synchronize()
        return out

    try:
        big, small = run(1), run(0)
        # round 6: the same kernel on the ping-pong schedule (k_gemm256_pp, the library's default): bit-identical, also on repetition beside
        # a bandwidth hog (a missing wait of the counted-vmcnt pipeline shows as a rare wrong tile)
        hog_stream, hog = torch.cuda.Stream(), torch.empty(64 << 20, dtype=torch.float32, device='cuda')
        for rep in range(4):
            if rep >= 2:
                with torch.cuda.stream(hog_stream):
                    for _ in range(4):
                        hog.add_(1.0)
            pp = run(2)
            assert torch.equal(pp.view(torch.int16), big.view(torch.int16)), 'ping-pong 256 x 256 GEMM differs (repetition %d)' % rep
    finally:
        lib.rart_igemm_set_gemm256(2)
    assert torch.isfinite(big.float()).all()
    assert torch.equal(big, small)
    ref = a.double() @ w.double().t() + bias.double()
    if res is not None:
        ref = ref + res.double()
    if flags == 4:
        ref = torch.nn.functional.gelu(ref)
    if flags == 8:
        ud = u.double()
        ref = ref * (0.5 * (1 + torch.erf(ud / 2 ** 0.5)) + ud * torch.exp(-ud * ud / 2) / (2 * torch.pi) ** 0.5)
    err = (big.double() - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -7 + 2e-3).all(), err.max().item()
