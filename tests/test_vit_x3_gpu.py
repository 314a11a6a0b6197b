"""Reference-precision mode of the ViT-B/16 engine (ViTEngine(precision='bf16x3'), alias 'fp32x').

The reference evaluates and attacks ViT in fp32 (exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:1-9: no precision key;
RobustART/noise/utils/adv/attack.py:20-23; Attacks/autoattack/autopgd_base.py:271-289) and the north star asks for logits within
1e-4 of it; the bf16 engine is ~3e-3 away.  This mode keeps every activation / gradient / weight as a hi + lo pair of bf16 planes,
forms every contraction as lo.hi + hi.lo + hi.hi on the bf16 MFMA with fp32 accumulation (rart_gemm_pair_bf16) and evaluates
LayerNorm / soft-max / GELU in fp32 on hi + lo (csrc/vit_pair.hip).  Tolerances stated here:
  * the pair GEMM vs fp64 of the same pair operands: <= 1e-5 of the output scale (fp32 out), 2.5e-5 (pair out: the output split);
  * logits vs the fp32 torch module AND vs an fp64 evaluation: <= 1e-4 of the logit scale;
  * gradient w.r.t. the input vs fp64 autograd through the module: relative L2 <= 5e-4 (GELU is smooth: no decision flips).
"""
import copy
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _split(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()


def _f64(p):
    return p[0].double() + p[1].double()


@pytest.fixture(scope='module')
def setup():
    from robustart_amd.model import get_model
    from robustart_amd.model.vit_engine import ViTEngine
    torch.manual_seed(0)
    m = get_model({'type': 'vit_base', 'kwargs': {'num_classes': 1000, 'drop_path_rate': 0.1}}).eval()
    g = torch.Generator().manual_seed(1)
    for n, p in m.named_parameters():                      # non-trivial biases / norms so that every epilogue term is exercised
        if n.endswith('bias'):
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.05)
        if 'norm' in n and n.endswith('weight'):
            p.data.copy_(1 + torch.randn(p.shape, generator=g) * 0.1)
    for p in m.parameters():
        p.requires_grad_(False)
    return m.cuda(), ViTEngine(m, 'cuda', precision='fp32x')


def _gemm_pair(lib, a, w, dst, M, N, K, lda, ldc, ldw=None, bias=None, res=None, flags=0, aux=None, w_rows=None, **kw):
    from robustart_amd import _lib
    d = _lib.GemmPairDesc()
    d.a_hi, d.a_lo, d.w_hi, d.w_lo = a[0].data_ptr(), a[1].data_ptr(), w[0].data_ptr(), w[1].data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    if res is not None:
        d.res_hi, d.res_lo = res[0].data_ptr(), res[1].data_ptr()
    if flags & 2:
        d.dst_hi = dst.data_ptr()
    else:
        d.dst_hi, d.dst_lo = dst[0].data_ptr(), dst[1].data_ptr()
    if aux is not None:
        d.aux_hi, d.aux_lo = aux[0].data_ptr(), aux[1].data_ptr()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = M, N, K, lda, ldw or K, ldc
    d.w_rows = w_rows if w_rows is not None else w.shape[-2]
    d.flags = flags
    for k, v in kw.items():
        setattr(d, k, v)
    return lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr())


def _gelu64(v):
    return 0.5 * v * (1 + torch.erf(v / 2 ** 0.5))


def test_pair_epilogue_gelu_and_its_derivative_on_a_dense_grid():
    """Round 6: the GELU / GELU' of the pair GEMM's epilogue are a fitted erfc form (csrc/rart_gemm_pair_dev.h) instead of libm's erff.  Pin
    the functions THEMSELVES on the GPU (v_rcp_f32 / v_exp_f32 included): the product is arranged to reproduce a controlled pre-activation u
    exactly (one non-zero per row of A, unit weights: u is a pair, u . 1 has no rounding), u runs over a dense grid of [-12, 12] plus the
    points around 0 and the tails; against fp64: |gelu error| <= 5e-7 + 2^-16 |gelu| (the output pair's rounding), same for gelu'."""
    from robustart_amd import _lib
    lib = _lib.load()
    M, N, K = 1 << 16, 8, 32
    u64 = torch.cat([torch.linspace(-12, 12, M - 2048, dtype=torch.float64), torch.linspace(-1e-3, 1e-3, 1024, dtype=torch.float64),
                     torch.linspace(-6.7, -5.0, 512, dtype=torch.float64), torch.linspace(5.0, 6.7, 512, dtype=torch.float64)]).cuda()
    a32 = torch.zeros(M, K, device='cuda')
    a32[:, 0] = u64.float()
    a = _split(a32)
    u = _f64(a)[:, 0]                                        # the value the pair represents: what the epilogue sees
    w32 = torch.zeros(N, K, device='cuda')
    w32[:, 0] = 1.0
    w = _split(w32)
    out = torch.full((2, M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
    assert _gemm_pair(lib, a, w, out, M, N, K, K, N, flags=4) == 0
    g = _f64(out)
    want = _gelu64(u)[:, None]
    err = (g - want).abs()
    print('gelu: max |error| %.3e' % err.max().item())
    assert (err <= 5e-7 + want.abs() * 2.0 ** -16).all()
    assert (g[:, :1] == g).all()                             # every column saw the same u
    # GELU': dst = (A . W) * gelu'(aux) with A . W = 1
    ones32 = torch.zeros(M, K, device='cuda')
    ones32[:, 0] = 1.0
    aux = torch.stack([a[0][:, :1].expand(M, N).contiguous(), a[1][:, :1].expand(M, N).contiguous()])
    assert _gemm_pair(lib, _split(ones32), w, out, M, N, K, K, N, flags=8, aux=aux) == 0
    gp = 0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5
    err = (_f64(out) - gp[:, None]).abs()
    print("gelu': max |error| %.3e" % err.max().item())
    assert (err <= 5e-7 + gp.abs()[:, None] * 2.0 ** -16).all()


@pytest.mark.parametrize('M,N,K', [(700, 768, 768), (5000, 512, 96), (197, 200, 64)])
def test_gemm_pair_kernel_vs_fp64(M, N, K):
    """rart_gemm_pair_bf16: plain product with bias + residual pair, the three GELU modes, fp32 output, against fp64 of the same hi / lo
    operands (only the dropped lo.lo term, fp32 accumulation and -- for pair outputs -- the output split differ).  Shapes: ragged M
    (a partial last row tile), >= 16 row tiles (the XCD remap), N below one tile with weight rows that do not exist (w_rows < N)."""
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M)
    a = _split((torch.randn(M, K, generator=g)).cuda())
    w_rows = N if N % 256 == 0 else N - 3                                    # the attention shape: 197 key rows for 200 columns
    w = _split((torch.randn(w_rows, K, generator=g) * 0.05).cuda())
    bias = torch.randn(N, generator=g).cuda()
    res = _split(torch.randn(M, N, generator=g).cuda())
    wz = torch.zeros(N, K, dtype=torch.float64, device='cuda')
    wz[:w_rows] = _f64(w)
    base = _f64(a) @ wz.t() + bias.double()
    scale = base.abs().max().item()
    # (1) bias + residual, pair out
    out = torch.full((2, M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
    assert _gemm_pair(lib, a, w, out, M, N, K, K, N, bias=bias, res=res, w_rows=w_rows) == 0
    err = (_f64(out) - (base + _f64(res))).abs().max().item()
    print('pair GEMM %dx%dx%d: scale %.2f, pair out err %.2e of scale' % (M, N, K, scale, err / scale))
    assert err <= 2.5e-5 * scale
    # (2) fp32 out, no residual
    o32 = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm_pair(lib, a, w, o32, M, N, K, K, N, bias=bias, flags=2, w_rows=w_rows) == 0
    err = (o32.double() - base).abs().max().item()
    print('   fp32 out err %.2e of scale' % (err / scale))
    assert err <= 1e-5 * scale
    # (3) GELU in the epilogue; GELU with the pre-activation kept; GELU' of a kept pre-activation
    small = base * (3.0 / scale)
    a2 = _split((_f64(a) * (3.0 / scale)).float())
    b2 = (bias.double() * (3.0 / scale)).float()
    small = _f64(a2) @ wz.t() + b2.double()
    assert _gemm_pair(lib, a2, w, out, M, N, K, K, N, bias=b2, flags=4, w_rows=w_rows) == 0
    assert (_f64(out) - _gelu64(small)).abs().max().item() <= 3e-5 * 3.0
    u = torch.full((2, M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
    assert _gemm_pair(lib, a2, w, out, M, N, K, K, N, bias=b2, flags=64, aux=u, w_rows=w_rows) == 0
    assert (_f64(u) - small).abs().max().item() <= 3e-5 * 3.0
    assert (_f64(out) - _gelu64(_f64(u))).abs().max().item() <= 2e-5 * 3.0      # gelu of the value the pair REPRESENTS
    assert _gemm_pair(lib, a, w, out, M, N, K, K, N, bias=bias, flags=8, aux=u, w_rows=w_rows) == 0
    uu = _f64(u)
    gp = 0.5 * (1 + torch.erf(uu / 2 ** 0.5)) + uu * torch.exp(-0.5 * uu * uu) / (2 * torch.pi) ** 0.5
    assert (_f64(out) - base * gp).abs().max().item() <= 3e-5 * scale
    # argument checks: K not a multiple of 32, missing lo plane
    assert _gemm_pair(lib, a, w, out, M, N, K - 8, K, N) != 0


def test_gemm_pair_row_rebasing_and_batched_products():
    """The class-token slot (destination rows re-based per image), the un-patchify source (source rows re-based) and a batched
    problem with strided operand planes (attention: Q K^T per (image, head) out of the qkv tensor)."""
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    B, P, T, D, K = 3, 196, 197, 256, 64
    a = _split(torch.randn(B * P, K, generator=g).cuda())
    w = _split((torch.randn(D, K, generator=g) * 0.1).cuda())
    out = torch.zeros(2, B * T, D, dtype=torch.bfloat16, device='cuda')
    assert _gemm_pair(lib, a, w, out, B * P, D, K, K, D, rows_per_image=P, dst_rows_per_image=T, dst_row_off=1) == 0
    want = (_f64(a) @ _f64(w).t()).view(B, P, D)
    got = _f64(out).view(B, T, D)
    assert (got[:, 0] == 0).all()                                            # the class-token rows are not touched
    assert (got[:, 1:] - want).abs().max().item() <= 2.5e-5 * want.abs().max().item()
    # source rows re-based: rows 1.. of every image
    src = _split(torch.randn(B * T, K, generator=g).cuda())
    o32 = torch.zeros(B * P, D, device='cuda')
    assert _gemm_pair(lib, src, w, o32, B * P, D, K, K, D, flags=2, rows_per_image=P, src_rows_per_image=T, src_row_off=1) == 0
    want = _f64(src).view(B, T, K)[:, 1:] @ _f64(w).t()
    assert (o32.double().view(B, P, D) - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    # batched Q K^T: qkv [B*T][3*Dh*H], z = (image, head)
    H, hd = 4, 64
    Dm = H * hd
    qkv = _split(torch.randn(B * T, 3 * Dm, generator=g).cuda())
    s_ld = 200
    sc = torch.full((B * H, T, s_ld), float('nan'), device='cuda')
    d = _lib.GemmPairDesc()
    d.a_hi, d.a_lo = qkv[0].data_ptr(), qkv[1].data_ptr()
    d.w_hi, d.w_lo = qkv[0].data_ptr() + Dm * 2, qkv[1].data_ptr() + Dm * 2
    d.dst_hi = sc.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc, d.w_rows, d.flags = T, s_ld, hd, 3 * Dm, 3 * Dm, s_ld, T, 2
    d.n_batched, d.z_inner = B * H, H
    d.a_z_outer, d.a_z_inner, d.w_z_outer, d.w_z_inner = T * 3 * Dm, hd, T * 3 * Dm, hd
    d.c_z_outer, d.c_z_inner = H * T * s_ld, T * s_ld
    _lib.check(lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
    q = _f64(qkv).view(B, T, 3, H, hd)
    want = torch.einsum('bqhd,bkhd->bhqk', q[:, :, 0], q[:, :, 1]).reshape(B * H, T, T)
    assert (sc[:, :, :T].double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    assert (sc[:, :, T:] == 0).all()                                         # key rows that do not exist read as zeros


def test_pair_row_kernels_vs_fp64():
    """LayerNorm forward / backward-to-input, soft-max rows forward / backward, class-token + position add, un-patchify from fp32:
    csrc/vit_pair.hip against torch fp64 on the values the pairs represent."""
    from robustart_amd import _lib
    lib = _lib.load()
    sp = _lib.stream_ptr()
    g = torch.Generator().manual_seed(0)
    rows, D = 37, 768
    x = _split((torch.randn(rows, D, generator=g) * 2 + 0.3).cuda())
    gam, bet = (1 + 0.1 * torch.randn(D, generator=g)).cuda(), torch.randn(D, generator=g).cuda()
    out = torch.empty_like(x)
    _lib.check(lib.rart_layernorm_pair(_lib.ptr(x[0]), _lib.ptr(x[1]), _lib.ptr(gam), _lib.ptr(bet), _lib.ptr(out[0]), _lib.ptr(out[1]),
                                       rows, D, D, D, 1e-6, sp))
    xr = _f64(x).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gam.double(), bet.double(), 1e-6)
    assert (_f64(out) - ref.detach()).abs().max().item() <= 2e-5 * ref.abs().max().item()
    dy = _split(torch.randn(rows, D, generator=g).cuda())
    res = _split(torch.randn(rows, D, generator=g).cuda())
    dx = torch.empty_like(x)
    _lib.check(lib.rart_layernorm_bwd_pair(_lib.ptr(dy[0]), _lib.ptr(dy[1]), _lib.ptr(x[0]), _lib.ptr(x[1]), _lib.ptr(gam), _lib.ptr(res[0]),
                                           _lib.ptr(res[1]), _lib.ptr(dx[0]), _lib.ptr(dx[1]), rows, D, D, D, D, D, 1e-6, sp))
    gx, = torch.autograd.grad(ref, xr, grad_outputs=_f64(dy))
    want = gx + _f64(res)
    assert (_f64(dx) - want).abs().max().item() <= 3e-5 * want.abs().max().item()
    _lib.check(lib.rart_layernorm_bwd_pair(_lib.ptr(dy[0]), _lib.ptr(dy[1]), _lib.ptr(x[0]), _lib.ptr(x[1]), _lib.ptr(gam), None, None,
                                           _lib.ptr(dx[0]), _lib.ptr(dx[1]), rows, D, D, D, 0, D, 1e-6, sp))
    assert (_f64(dx) - gx).abs().max().item() <= 3e-5 * gx.abs().max().item()
    # soft-max rows: fp32 scores [rows][200] -> pair probabilities [rows][224], 197 valid keys
    R, T, s_ld, t_pad = 50, 197, 200, 224
    s = (torch.randn(R, s_ld, generator=g) * 8).cuda()
    p = torch.full((2, R, t_pad), float('nan'), dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_softmax_rows_pair(_lib.ptr(s), _lib.ptr(p[0]), _lib.ptr(p[1]), R, T, s_ld, t_pad, 0.125, sp))
    sr = s[:, :T].double().requires_grad_(True)
    pref = torch.softmax(sr * 0.125, -1)
    assert (_f64(p)[:, :T] - pref.detach()).abs().max().item() <= 1e-5
    assert (_f64(p)[:, T:] == 0).all()
    dp = torch.randn(R, s_ld, generator=g).cuda()
    ds = torch.full((2, R, t_pad), float('nan'), dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_softmax_bwd_rows_pair(_lib.ptr(p[0]), _lib.ptr(p[1]), _lib.ptr(dp), _lib.ptr(ds[0]), _lib.ptr(ds[1]), R, T, t_pad,
                                              s_ld, t_pad, 0.125, sp))
    gs, = torch.autograd.grad(pref, sr, grad_outputs=dp[:, :T].double())
    assert (_f64(ds)[:, :T] - gs).abs().max().item() <= 3e-5 * gs.abs().max().item() + 1e-7
    assert (_f64(ds)[:, T:] == 0).all()
    # class token + position embedding
    n, t, d = 3, 5, 64
    xx = _split(torch.randn(n, t, d, generator=g).cuda())
    before = _f64(xx).clone()
    cls0, pos = torch.randn(d, generator=g).cuda(), torch.randn(t, d, generator=g).cuda()
    _lib.check(lib.rart_vit_add_pos_cls_pair(_lib.ptr(xx[0]), _lib.ptr(xx[1]), _lib.ptr(cls0), _lib.ptr(pos), n, t, d, sp))
    want = before + pos.double()
    want[:, 0] = cls0.double()
    assert (_f64(xx) - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    # un-patchify from fp32
    B, Himg, ps = 2, 32, 16
    kk = 3 * ps * ps
    dpat = torch.randn(B * (Himg // ps) ** 2, kk, generator=g).cuda()
    grad = torch.empty(B, 3, Himg, Himg, device='cuda')
    _lib.check(lib.rart_vit_unpatchify_from_f32(_lib.ptr(dpat), _lib.ptr(grad), B, Himg, Himg, ps, kk, (ctypes.c_float * 3)(*STD), sp))
    ref = dpat.view(B, Himg // ps, Himg // ps, 3, ps, ps).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, Himg, Himg)
    ref = ref / torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    torch.testing.assert_close(grad, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('B', [3, 8])
def test_vit_x3_logits_within_1e4_of_fp32_and_fp64(setup, B):
    m, eng = setup
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 3, 224, 224, generator=g).cuda()
    got = eng.logits(x, MEAN, STD)
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    pure = m((x - mean) / std)
    m64 = copy.deepcopy(m).cpu().double()                     # fp64 on the host: independent of rocBLAS
    ref = m64((x.cpu().double() - mean.cpu().double()) / std.cpu().double()).cuda()
    scale = ref.abs().max().item()
    e32 = (got.double() - pure.double()).abs().max().item() / scale
    e64 = (got.double() - ref).abs().max().item() / scale
    t32 = (pure.double() - ref).abs().max().item() / scale
    print('ViT x3 logits B=%d: scale %.2f; |x3 - fp32 module| %.2e, |x3 - fp64| %.2e, |fp32 module - fp64| %.2e (of scale)'
          % (B, scale, e32, e64, t32))
    assert e32 <= 1e-4 and e64 <= 1e-4                       # the north star's tolerance
    assert (got.argmax(1) == ref.argmax(1)).all()
    # the uint8 entry (the corruption kernels' output) and batch invariance
    u8 = (x * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    a = eng.logits_from_u8(u8, MEAN, STD).clone()
    b = eng.logits(u8.permute(0, 3, 1, 2).float() / 255, MEAN, STD).clone()
    assert (a - b).abs().max().item() <= 2e-5 * scale        # u8 / 255 by multiply vs divide: one fp32 ulp on the pixel
    one = eng.logits_from_u8(u8[1:2].contiguous(), MEAN, STD)
    assert torch.equal(one[0], a[1])                          # per-element arithmetic does not depend on the batch


@pytest.mark.parametrize('kind', [0, 1])
def test_vit_x3_gradient_vs_fp64_autograd(setup, kind):
    """forward_backward (CE and DLR) on the pair engine against fp64 autograd through the module with the engine's loss gradient
    pushed through; torch's own fp32 autograd measured by the same yardstick beside it."""
    m, eng = setup
    torch.manual_seed(5)
    B = 4
    x = torch.rand(B, 3, 224, 224, device='cuda')
    y = torch.randint(0, 1000, (B,), device='cuda')
    logits, loss, grad, pred = eng.forward_backward(x, MEAN, STD, y, kind)
    dl = eng.last_dlogits
    mean = torch.tensor(MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float64).view(1, 3, 1, 1)
    m64 = copy.deepcopy(m).cpu().double()
    xr = x.cpu().double().requires_grad_(True)
    lg = m64((xr - mean) / std)
    want, = torch.autograd.grad((lg * dl.double().cpu()).sum(), xr)
    want, lg = want.cuda(), lg.detach().cuda()
    xt = x.clone().requires_grad_(True)
    lt = m((xt - mean.float().cuda()) / std.float().cuda())
    gt, = torch.autograd.grad((lt * dl).sum(), xt)
    a, b = grad.double().flatten(1), want.flatten(1)
    rel = ((a - b).norm(dim=1) / b.norm(dim=1)).cpu()
    rel_t = ((gt.double().flatten(1) - b).norm(dim=1) / b.norm(dim=1)).cpu()
    cos = ((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).cpu()
    lerr = (logits.double() - lg).abs().max().item() / lg.abs().max().item()
    print('ViT x3 gradient kind=%d: rel L2 %s; torch fp32 autograd by the same yardstick %s; 1 - cos %s; logits %.2e'
          % (kind, rel.tolist(), rel_t.tolist(), (1 - cos).tolist(), lerr))
    assert lerr <= 1e-4
    assert (rel <= 5e-4).all() and (cos >= 1 - 2e-7).all()
    assert torch.equal(pred.long(), lg.argmax(1))


def test_vit_x3_through_engine_model_and_pgd(setup):
    """make_engine / EngineModel(precision=...) carry the mode for ViT; a PGD run stays in the eps ball and its final logits agree
    with the fp32 module on the same adversarial examples to 1e-4; a bf16 EngineModel hands FAB its reference-precision engine."""
    from robustart_amd.model.engine import EngineModel, make_engine
    from robustart_amd.noise import adv
    m, eng = setup
    assert make_engine(m, 'cuda', 'fp32x').precision == 'bf16x3'
    f = EngineModel(None, takes_normalized=False, engine=eng)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (2,), generator=g).cuda()
    eps = 4 / 255
    xa = adv.pgd_linf(x, y, f, eps, 3 / 40, 2, seed=2)
    assert (xa - x).abs().max().item() <= eps + 1e-6 and xa.min().item() >= 0 and xa.max().item() <= 1
    assert not torch.equal(xa, x)
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    want = m((xa - mean) / std)
    assert (f(xa) - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    fb = EngineModel(m, takes_normalized=False)                                # bf16 ViT engine from the module
    assert fb.rart_engine.precision == 'bf16' and fb.rart_reference_engine().precision == 'bf16x3'
    with pytest.raises(ValueError):
        from robustart_amd.model.vit_engine import ViTEngine
        ViTEngine(m, 'cuda', precision='fp16')


@pytest.mark.parametrize('T', [197, 33, 224, 64, 130])
def test_fused_pair_attention_vs_fp64_every_key_tile_count(T):
    """rart_vit_attention_pair (one workgroup per (image, head), K / V^T pairs resident in LDS, soft-max in fp32 registers, every
    contraction three MFMA products) against fp64 soft-max attention on the values the pair REPRESENTS; key-tile counts 2 .. 7 with
    full and partial last tiles."""
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(T)
    B, H, hd = 2, 3, 64
    D = H * hd
    qkv = _split((torch.randn(B * T, 3 * D, generator=g) * 1.5).cuda())
    out = torch.full((2, B * T, D), float('nan'), dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_vit_attention_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(out[0]), _lib.ptr(out[1]), B, T, H, hd,
                                           _lib.stream_ptr()))
    q = _f64(qkv).view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    want = (torch.softmax(q[0] @ q[1].transpose(-1, -2) / 8.0, -1) @ q[2]).permute(0, 2, 1, 3).reshape(B * T, D)
    err = (_f64(out) - want).abs().max().item() / want.abs().max().item()
    print('pair attention T=%d: max err %.2e of scale' % (T, err))
    assert err <= 2.5e-5


@pytest.mark.parametrize('T', [197, 33, 224, 64, 130])
def test_fused_pair_attention_backward_vs_fp64_autograd(T):
    """rart_vit_attention_bwd_pair (query-side launch: dQ + per-query statistics; key-side launch: dK, dV) against fp64 autograd of
    soft-max attention on the values the pairs represent -- O is the fp64 forward's output rounded to a pair, as the engine hands it."""
    from robustart_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(100 + T)
    B, H, hd = 2, 3, 64
    D = H * hd
    qkv = _split((torch.randn(B * T, 3 * D, generator=g) * 1.5).cuda())
    dout = _split(torch.randn(B * T, D, generator=g).cuda())
    q64 = _f64(qkv).requires_grad_(True)
    qh = q64.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    o64 = (torch.softmax(qh[0] @ qh[1].transpose(-1, -2) / 8.0, -1) @ qh[2]).permute(0, 2, 1, 3).reshape(B * T, D)
    (o64 * _f64(dout)).sum().backward()
    out = _split(o64.detach().float())
    dqkv = torch.full((2, B * T, 3 * D), float('nan'), dtype=torch.bfloat16, device='cuda')
    stats = torch.empty(B * H * ((T + 31) // 32 * 32) * 4, dtype=torch.float32, device='cuda')
    _lib.check(lib.rart_vit_attention_bwd_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(dout[0]),
                                               _lib.ptr(dout[1]), _lib.ptr(dqkv[0]), _lib.ptr(dqkv[1]), _lib.ptr(stats), B, T, H, hd,
                                               _lib.stream_ptr()))
    got, want = _f64(dqkv), q64.grad
    assert torch.isfinite(got).all()
    for name, lo in (('dQ', 0), ('dK', D), ('dV', 2 * D)):
        w = want[:, lo:lo + D]
        err = (got[:, lo:lo + D] - w).abs().max().item() / w.abs().max().item()
        print('pair attention backward T=%d %s: max err %.2e of scale' % (T, name, err))
        assert err <= 3e-5, (name, err)


def test_vit_x3_fused_attention_backward_matches_the_unfused_path(setup):
    """forward_backward with the fused pair attention backward against the same engine on the decomposition into batched pair
    products: the same logits bit for bit (the forward is shared), the input gradient to summation-order noise."""
    m, eng = setup
    g = torch.Generator().manual_seed(13)
    x = torch.rand(3, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (3,), generator=g).cuda()
    assert eng.fused_attention_bwd
    la, _, ga, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    la, ga = la.clone(), ga.clone()
    eng.fused_attention_bwd = False
    try:
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
    finally:
        eng.fused_attention_bwd = True
    assert torch.equal(la, lb)
    err = (ga - gb).abs().max().item() / gb.abs().max().item()
    print('ViT x3 input gradient, fused vs unfused attention backward: %.2e of scale' % err)
    assert err <= 5e-5


def test_vit_x3_fused_attention_matches_the_unfused_path(setup):
    """The forward with the fused pair attention against the same engine on the decomposition into batched pair products (fp32 scores,
    soft-max rows, V transposes): same operands, other summation order."""
    m, eng = setup
    g = torch.Generator().manual_seed(12)
    x = torch.rand(3, 3, 224, 224, generator=g).cuda()
    assert eng.fused_attention
    a = eng.logits(x, MEAN, STD).clone()
    eng.fused_attention = False
    try:
        b = eng.logits(x, MEAN, STD).clone()
    finally:
        eng.fused_attention = True
    err = (a - b).abs().max().item() / b.abs().max().item()
    print('ViT x3 logits, fused vs unfused attention: %.2e of scale' % err)
    assert err <= 2e-5


def test_vit_engines_b256_match_small_batches_bit_for_bit(setup):
    """parity batches are 2-3 images.  At the benchmark's B = 256 every 32nd image of a forward / forward + backward of BOTH
    ViT engines equals the same image run in a batch of 2, bit for bit (the kernels' arithmetic per output element does not depend on
    the batch: same K order, same tiles) -- so the small-batch parity against the fp32 module / fp64 carries to the full size."""
    from robustart_amd.model.vit_engine import ViTEngine
    m, eng3 = setup
    g = torch.Generator().manual_seed(5)
    x = torch.rand(256, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 1000, (256,), generator=g).cuda()
    for name, eng in (('fp32x', eng3), ('bf16', ViTEngine(m, 'cuda'))):
        big = eng.logits(x, MEAN, STD).clone()
        lb, _, gb, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        lb, gb = lb.clone(), gb.clone()
        if name == 'fp32x':          # (the bf16 engine's keep-mode fc1 applies GELU to the ROUNDED pre-activation, its forward-only fc1 to the fp32 one)
            assert torch.equal(big, lb), name
        for i in range(0, 256, 32):
            xs, ys = x[i:i + 2].contiguous(), y[i:i + 2].contiguous()
            small = eng.logits(xs, MEAN, STD)
            assert torch.equal(small[0], big[i]) and torch.equal(small[1], big[i + 1]), (name, i)
            ls, _, gs, _ = eng.forward_backward(xs, MEAN, STD, ys, 0)
            assert torch.equal(ls[0], lb[i]) and torch.equal(gs[0], gb[i]) and torch.equal(gs[1], gb[i + 1]), (name, i)
    # and the accuracy statement at the full size on a sample of rows: the fp32 module on 8 of the 256 images
    mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
    rows = torch.arange(0, 256, 32, device='cuda')
    ref = m((x[rows] - mean) / std)
    big3 = eng3.logits(x, MEAN, STD)
    assert (big3[rows] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize('B,T', [(256, 197), (100, 197), (90, 224), (88, 193)])
def test_walking_pair_attention_is_bit_identical(B, T, monkeypatch):
    """Round 6: the forward pair attention with one workgroup per CU walking a run of (image, head) items -- wave 7 brings the next item's K
    (during soft-max + P V) and V (during S = K Q^T) with buffer_load ... lds, V row-major and gathered -- against the one-item-per-workgroup
    kernel: the same products in the same order, equal bit for bit; repeated beside a bandwidth hog (a missing wait of the loader shows as a
    rare wrong item).  Uneven runs (100 x 12 items over 256 workgroups), full and minimal seventh key tiles."""
    from robustart_amd import _lib
    lib = _lib.load()
    H, hd = 12, 64
    g = torch.Generator().manual_seed(B + T)
    qkv = _split((torch.randn(B * T, 3 * H * hd, generator=g) * 1.5).cuda())
    sp = _lib.stream_ptr()

    def run():
        out = torch.full((2, B * T, H * hd), float('nan'), dtype=torch.bfloat16, device='cuda')
        _lib.check(lib.rart_vit_attention_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(out[0]), _lib.ptr(out[1]), B, T, H, hd, sp))
        torch.cuda.synchronize()
        return out
    monkeypatch.setenv('RART_ATT_WALK', '0')
    want = run()
    assert torch.isfinite(want.float()).all()
    monkeypatch.setenv('RART_ATT_WALK', '1')
    hog_stream, hog = torch.cuda.Stream(), torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    for rep in range(5):
        if rep >= 2:
            with torch.cuda.stream(hog_stream):
                for _ in range(4):
                    hog.add_(1.0)
        got = run()
        bad = (got.view(torch.int16) != want.view(torch.int16)).any(0)
        assert not bad.any(), 'walking attention differs (repetition %d): %d elements, first rows %s' % (
            rep, int(bad.sum()), bad.any(1).nonzero().flatten()[:8].tolist())


@pytest.mark.parametrize('B,T', [(256, 197), (100, 224), (90, 193)])
def test_walking_pair_attention_backward_is_bit_identical(B, T, monkeypatch):
    """Round 6: the backward pair attention with both kernels walking their items -- backward-q: loader wave, K by LDS-DMA, V under S + soft-max;
    backward-kv: the next item's Q / dO arrive in 64-row slices as the current item's query tiles release them -- against the
    one-item-per-workgroup kernels: dQ / dK / dV and the per-query statistics equal bit for bit, repeatedly, beside a hog."""
    from robustart_amd import _lib
    lib = _lib.load()
    H, hd = 12, 64
    g = torch.Generator().manual_seed(7 * B + T)
    qkv = _split((torch.randn(B * T, 3 * H * hd, generator=g) * 1.2).cuda())
    dout = _split((torch.randn(B * T, H * hd, generator=g) * 0.1).cuda())
    sp = _lib.stream_ptr()
    out = torch.empty(2, B * T, H * hd, dtype=torch.bfloat16, device='cuda')
    _lib.check(lib.rart_vit_attention_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(out[0]), _lib.ptr(out[1]), B, T, H, hd, sp))
    TP = (T + 31) // 32 * 32

    def run():
        dq = torch.full((2, B * T, 3 * H * hd), float('nan'), dtype=torch.bfloat16, device='cuda')
        stats = torch.zeros(B * H * TP, 4, device='cuda')
        _lib.check(lib.rart_vit_attention_bwd_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(dout[0]),
                                                   _lib.ptr(dout[1]), _lib.ptr(dq[0]), _lib.ptr(dq[1]), _lib.ptr(stats), B, T, H, hd, sp))
        torch.cuda.synchronize()
        return dq, stats
    monkeypatch.setenv('RART_ATT_WALK', '0')
    want, wstats = run()
    assert torch.isfinite(want.float()).all()
    monkeypatch.setenv('RART_ATT_WALK', '2')          # 2 (= the default with the variable unset): backward-q AND backward-kv walk; 1: backward-q alone
    hog_stream, hog = torch.cuda.Stream(), torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    for rep in range(4):
        if rep >= 2:
            with torch.cuda.stream(hog_stream):
                for _ in range(4):
                    hog.add_(1.0)
        got, gstats = run()
        bad = (got.view(torch.int16) != want.view(torch.int16)).any(0)
        assert not bad.any(), 'walking backward differs (repetition %d): %d elements, first rows %s' % (
            rep, int(bad.sum()), bad.any(1).nonzero().flatten()[:8].tolist())
        assert torch.equal(gstats.view(torch.int32)[:, :3], wstats.view(torch.int32)[:, :3])
