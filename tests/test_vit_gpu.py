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
