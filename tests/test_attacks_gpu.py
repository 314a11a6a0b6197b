"""GPU parity tests: HIP attack step kernels and attack drivers vs the CPU oracle
(oracle/attacks_ref.py, itself pinned against unmodified reference APGD / MIM)."""
import os

import numpy as np
import pytest
import torch

from _tinynet import make_tinynet
from oracle import attacks_ref as A

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _rand(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


@pytest.mark.parametrize('shape', [(4, 3, 32, 32), (3, 3, 7, 5), (2, 3, 224, 224)])
def test_pgd_linf_step_bit_exact(shape):
    from robustart_amd.noise import adv
    x0 = _rand(shape, 1)
    x = torch.clamp(x0 + _rand(shape, 2, -0.03, 0.03), 0, 1)
    g = _rand(shape, 3, -1, 1)
    g.view(-1)[::17] = 0.0                      # sign(0) = 0
    eps, alpha = 8 / 255, 8 / 255 * 3 / 40
    want = A.pgd_linf_step(x, g, x0, eps, alpha)
    got = adv.pgd_step_linf_(x.cuda().contiguous(), g.cuda(), x0.cuda(), eps, alpha).cpu()
    assert torch.equal(got, want)
    assert (got - x0).abs().max() <= eps + 1e-7 and got.min() >= 0 and got.max() <= 1
    # idempotent projection: a zero-gradient step leaves a feasible point unchanged
    again = adv.pgd_step_linf_(got.cuda().contiguous(), torch.zeros_like(g).cuda(), x0.cuda(), eps, alpha).cpu()
    assert torch.allclose(again, got, atol=1e-7)


def test_pgd_l2_and_mim_steps():
    from robustart_amd.noise import adv
    shape = (3, 3, 32, 32)
    x0 = _rand(shape, 1)
    x = torch.clamp(x0 + _rand(shape, 2, -0.01, 0.01), 0, 1)
    g = _rand(shape, 3, -1, 1)
    want = A.pgd_l2_step(x, g, x0, 0.5, 0.05)
    got = adv.pgd_step_l2_(x.cuda().contiguous(), g.cuda(), x0.cuda(), 0.5, 0.05).cpu()
    torch.testing.assert_close(got, want, atol=1e-6, rtol=0)
    assert ((got - x0).flatten(1).norm(dim=1) <= 0.5 + 1e-5).all()
    m = _rand(shape, 4, -1, 1)
    wx, wm = A.mim_step(x, g, m, x0, 8 / 255, 0.002, 1.0)
    mg = m.cuda().contiguous()
    gx = adv.mim_step_(x.cuda().contiguous(), mg, g.cuda(), x0.cuda(), 8 / 255, 0.002, 1.0).cpu()
    torch.testing.assert_close(gx, wx, atol=1e-6, rtol=0)
    torch.testing.assert_close(mg.cpu(), wm, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('norm,eps', [('Linf', 8 / 255), ('L2', 0.5)])
def test_apgd_step_and_init(norm, eps):
    from robustart_amd.noise import adv
    shape = (4, 3, 32, 32)
    x0 = _rand(shape, 1)
    t = _rand(shape, 5, -1, 1)
    want = (x0 + eps * torch.ones_like(x0) * A._apgd_normalize(t, norm)).clamp(0, 1)
    got = adv.apgd_init(x0.cuda(), norm, eps, injected_t=t.cuda().contiguous()).cpu()
    torch.testing.assert_close(got, want, atol=1e-7, rtol=0)
    xa = want.clone()
    xold = torch.clamp(xa + _rand(shape, 6, -0.01, 0.01), 0, 1)
    grad = _rand(shape, 7, -1, 1)
    step = torch.tensor([2 * eps, eps, eps / 2, 2 * eps])
    for a in (1.0, 0.75):
        fn = A.apgd_step_linf if norm == 'Linf' else A.apgd_step_l2
        w = fn(xa, xold, grad, x0, eps, step.view(-1, 1, 1, 1), a)
        xa_g, xo_g = xa.cuda().contiguous(), xold.cuda().contiguous()
        adv.apgd_step_(xa_g, xo_g, grad.cuda(), x0.cuda(), step.cuda(), norm, eps, a)
        torch.testing.assert_close(xa_g.cpu(), w, atol=(0 if norm == 'Linf' else 2e-6), rtol=0)
        assert torch.equal(xo_g.cpu(), xa)
    # native random start stays inside the eps-ball and the box, and touches the ball's surface
    nat = adv.apgd_init(x0.cuda(), norm, eps, seed=3).cpu()
    d = (nat - x0)
    if norm == 'Linf':
        assert d.abs().max() <= eps + 1e-7
    else:
        assert (d.flatten(1).norm(dim=1) <= eps + 1e-5).all()


def test_select_rows_and_logit_losses():
    from robustart_amd.noise import adv
    src, dst = _rand((5, 3, 8, 8), 1), _rand((5, 3, 8, 8), 2)
    mask = torch.tensor([1, 0, 1, 0, 0], dtype=torch.bool)
    got = adv.select_rows_(dst.cuda().contiguous(), src.cuda(), mask.cuda()).cpu()
    want = dst.clone()
    want[mask] = src[mask]
    assert torch.equal(got, want)
    for C in (10, 1000):
        z = (_rand((16, C), 3, -4, 4)).requires_grad_(True)
        y = torch.randint(0, C, (16,), generator=torch.Generator().manual_seed(1))
        y[:4] = z[:4].argmax(1)                     # mix of correct / incorrect rows
        yt = (y + 1 + torch.randint(0, C - 1, (16,), generator=torch.Generator().manual_seed(2))) % C
        for kind, fn in ((0, lambda: A.ce_indiv(z, y)), (1, lambda: A.dlr_loss(z, y)),
                         (2, lambda: A.dlr_loss_targeted(z, y, yt)), (3, lambda: A.margin_loss(z, y))):
            li = fn()
            gw, = torch.autograd.grad(li.sum() * 0.5, z)
            loss, dl, pred = adv.logit_loss(z.detach().cuda(), y.cuda(), kind, yt.cuda(), 0.5)
            torch.testing.assert_close(loss.cpu(), li.detach(), atol=2e-5, rtol=2e-5)
            torch.testing.assert_close(dl.cpu(), gw, atol=2e-5, rtol=2e-4)
            assert torch.equal(pred.cpu().long(), z.argmax(1))


def test_ce_gradient_of_saturated_rows_sums_to_zero():
    """Rows whose softmax is saturated (1 - p_y between 1e-9 and 1e-5, logits around 20-30): d(CE)/dz must keep dl_y = -(sum of the
    others) -- lse = log(s) + max formed first loses log(s) below the ulp of the maximum and the input gradient of a network with a
    common-mode logit Jacobian flips its sign (found on the fitted ResNet-50 of test_outcome_gpu.py).  Against fp64."""
    from robustart_amd.noise import adv
    g = torch.Generator().manual_seed(9)
    C, B = 1000, 64
    z = torch.randn(B, C, generator=g) * 2.0
    y = torch.randint(0, C, (B,), generator=g)
    margin = torch.linspace(12.0, 24.0, B)                # 1 - p_y from ~1e-3 down to ~1e-9 (1 000 classes)
    z[torch.arange(B), y] = z.max(1)[0] + margin
    z = z + 20.0                                          # logits at the scale where an fp32 ulp is 2e-6
    loss, dl, pred = adv.logit_loss(z.cuda(), y.cuda(), 0, None, 1.0)
    z64 = z.double().requires_grad_(True)
    l64 = torch.nn.functional.cross_entropy(z64, y, reduction='none')
    g64, = torch.autograd.grad(l64.sum(), z64)
    dl = dl.cpu().double()
    assert (dl.sum(1).abs() <= 1e-6 * dl.abs().sum(1)).all()                     # the gradient has no common-mode component
    rel = (dl - g64).abs().max(1)[0] / g64.abs().max(1)[0]
    print('CE gradient of saturated rows vs fp64: max relative error %.2e; loss %.2e' % (rel.max().item(), ((loss.cpu().double() - l64.detach()).abs() / l64.detach()).max().item()))
    assert rel.max().item() <= 1e-5
    assert ((loss.cpu().double() - l64.detach()).abs() <= 1e-5 * l64.detach()).all()
    assert torch.equal(pred.cpu().long(), y)


def _gold_model():
    g = np.load(os.path.join(GOLD, 'attacks_ref.npz'))
    net = make_tinynet({k[4:]: g[k] for k in g.files if k.startswith('net/')})
    return g, net


def test_full_attacks_match_oracle_and_reference_goldens():
    """End to end on the tiny CNN: HIP step kernels + torch-autograd gradients vs (a) the oracle run on
    CPU with the same injected starts, (b) the reference's own outputs (golden)."""
    from robustart_amd.noise import adv
    g, net = _gold_model()
    netc = make_tinynet({k[4:]: g[k] for k in g.files if k.startswith('net/')}).cuda()
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    f_cpu = lambda z: net(A.normalize(z))  # noqa: E731
    mean = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()
    std = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()
    f_gpu = lambda z: netc((z - mean) / std)  # noqa: E731
    tol = dict(atol=2e-5, rtol=0)

    # PGD-Linf / FGSM / PGD-L2 (foolbox semantics, unpinned): same injected start as the oracle
    eps = 8 / 255
    u = _rand(x.shape, 9, -eps, eps)
    want = A.pgd_linf(f_cpu, x, y, eps, 3 / 40, 7, init_u=u)
    got = adv.pgd_linf(x.cuda(), y.cuda(), f_gpu, eps, 3 / 40, 7, init_u=u.cuda().contiguous()).cpu()
    torch.testing.assert_close(got, want, **tol)
    torch.testing.assert_close(adv.fgsm(x.cuda(), y.cuda(), f_gpu, eps).cpu(), A.fgsm(f_cpu, x, y, eps), **tol)
    d0 = A.l2_ball_start(torch.randn(4, 3 * 32 * 32 + 2, generator=torch.Generator().manual_seed(4)), 0.5).view_as(x)
    want = A.pgd_l2(f_cpu, x, y, 0.5, 3 / 40, 5, init_delta=d0)
    got = adv.pgd_l2(x.cuda(), y.cuda(), f_gpu, 0.5, 3 / 40, 5, init_delta=d0.cuda()).cpu()
    torch.testing.assert_close(got, want, **tol)

    # MIM vs the reference's golden output
    torch.manual_seed(11)
    noise = torch.FloatTensor(*x.shape).uniform_(-8 / 255, 8 / 255)
    got = adv.mim_linf(x.cuda(), y.cuda(), netc, 8 / 255, 5, 0.002, 1.0, init_noise=noise.cuda()).cpu()
    torch.testing.assert_close(got, torch.from_numpy(g['mim/adv']), **tol)

    # APGD (Linf / L2, ce / dlr) vs the reference's golden outputs
    for norm, e in (('Linf', 8 / 255), ('L2', 0.5)):
        for loss in ('ce', 'dlr'):
            torch.random.manual_seed(0)
            t = 2 * torch.rand(x.shape) - 1 if norm == 'Linf' else torch.randn(x.shape)
            got = adv.apgd_perturb(f_gpu, x.cuda(), y.cuda(), norm, e, 10, loss, 1, init_ts=[t.cuda().contiguous()])
            torch.testing.assert_close(got.cpu(), torch.from_numpy(g[f'apgd/{norm}/{loss}/adv']), **tol)


def test_square_attack_matches_reference_golden():
    """Square (Linf): HIP proposal / margin / select kernels, injected window + sign draws replaying the
    reference's torch stream -> the reference's own output (golden) within fp32 noise."""
    from robustart_amd.noise import adv
    g, net = _gold_model()
    netc = make_tinynet({k[4:]: g[k] for k in g.files if k.startswith('net/')}).cuda()
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    mean = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()
    std = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()
    f_gpu = lambda z: netc((z - mean) / std)  # noqa: E731
    torch.random.manual_seed(0)
    c, h, w = 3, 32, 32
    init = torch.sign(2 * torch.rand([4, c, 1, w]) - 1)
    draws = []
    for i in range(40):
        p = A.square_p_selection(i, 0.8, 40, False)
        s = max(int(round((p * h * w) ** 0.5)), 1)
        vh = int((0 + (h - s) * torch.rand([1])).long())
        vw = int((0 + (w - s) * torch.rand([1])).long())
        draws.append((vh, vw, torch.sign(2 * torch.rand([c, 1, 1]) - 1).view(c)))
    got = adv.square_perturb(f_gpu, x.cuda(), y.cuda(), 8 / 255, 40, 0.8, False, init_sign=init.view(4, c, w).cuda(),
                             draws=draws, check_every=1)
    torch.testing.assert_close(got.cpu(), torch.from_numpy(g['square/Linf/adv']), atol=2e-6, rtol=0)
    # native draws: stays in the eps-ball / box, deterministic
    a = adv.square_perturb(f_gpu, x.cuda(), y.cuda(), 8 / 255, 30, 0.8, False, seed=4, sample_offset=0)
    b = adv.square_perturb(f_gpu, x.cuda(), y.cuda(), 8 / 255, 30, 0.8, False, seed=4, sample_offset=0)
    assert torch.equal(a, b) and (a - x.cuda()).abs().max() <= 8 / 255 + 1e-6 and a.min() >= 0 and a.max() <= 1


def test_fab_projection_and_fab_t_match_reference():
    """rart_fab_project_linf (bisection, no sort) vs the reference's sort-based projection output (golden), then the
    full targeted FAB run vs the reference's adversarial examples."""
    from robustart_amd.noise import adv
    g, net = _gold_model()
    t, w, b = (torch.from_numpy(g['fabproj/' + k]).cuda() for k in ('t', 'w', 'b'))
    d, rm = adv.fab_project_linf(t.contiguous(), w.contiguous(), b.contiguous())
    ref = torch.from_numpy(g['fabproj/d'])
    torch.testing.assert_close(d.cpu(), ref, atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(rm.cpu(), ref.abs().max(1)[0], atol=2e-6, rtol=1e-5)
    # ImageNet-sized rows: the step lands on the hyperplane (or the box stops it) and stays inside the box
    gen = torch.Generator().manual_seed(2)
    n = 3 * 224 * 224
    t2 = torch.rand(4, n, generator=gen).cuda()
    w2 = torch.randn(4, n, generator=gen).cuda() * 1e-3
    b2 = ((w2 * t2).sum(1) + torch.tensor([0.5, -0.3, 2.0, -1e3]).cuda()).contiguous()
    d2, _ = adv.fab_project_linf(t2, w2, b2)
    want = A.fab_projection_linf(t2.cpu(), w2.cpu(), b2.cpu())
    assert (d2.cpu() - want).abs().max() < 5e-5
    y2 = t2 + d2
    assert y2.min() >= -1e-6 and y2.max() <= 1 + 1e-6
    resid = ((w2 * y2).sum(1) - b2).abs().cpu()
    assert (resid[:3] < 2e-3).all()                     # reachable rows sit on the hyperplane (fp32 sums of 150k terms)
    netc = make_tinynet({k[4:]: g[k] for k in g.files if k.startswith('net/')}).cuda()
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    mean = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()
    std = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()
    f_gpu = lambda z: netc((z - mean) / std)  # noqa: E731
    got = adv.fab_targeted_perturb(f_gpu, x.cuda(), y.cuda(), 8 / 255, 6, 3)
    torch.testing.assert_close(got.cpu(), torch.from_numpy(g['fabt/Linf/adv']), atol=5e-5, rtol=0)


def test_fab_l2_l1_projections_and_fab_t_match_reference():
    """rart_fab_project, norm L2 / L1 (bit-pattern bisection, no sort) vs the reference's sort-based projection_l2 / projection_l1
    outputs (golden: near / far / unreachable hyperplanes, zero and sub-1e-8 gradient entries, points on the box faces), the same at
    ImageNet row length vs the oracle with the constraint properties, then the whole targeted FAB run vs the reference's output."""
    from robustart_amd.noise import adv
    g = np.load(os.path.join(GOLD, 'fab_l2_l1_ref.npz'))
    t, w, b = (torch.from_numpy(g['proj/' + k]).cuda().contiguous() for k in 'twb')
    for norm, key in (('L2', 'd_l2'), ('L1', 'd_l1')):
        d, rn = adv.fab_project(t, w, b, norm)
        ref = torch.from_numpy(g['proj/' + key])
        torch.testing.assert_close(d.cpu(), ref, atol=3e-6 if norm == 'L2' else 3e-5, rtol=2e-5)   # (L1: the partly-moved coordinate is a residual / w)
        want = ref.abs().sum(1) if norm == 'L1' else ref.pow(2).sum(1).sqrt()
        torch.testing.assert_close(rn.cpu(), want, atol=1e-5, rtol=1e-5)
    gen = torch.Generator().manual_seed(2)
    n = 3 * 224 * 224
    t2 = torch.rand(5, n, generator=gen)
    w2 = torch.randn(5, n, generator=gen) * 1e-3
    w2[4] = (w2[4] * 4e3).round() / 4e3                  # multiples of 2.5e-4: many tied |w| values: the L1 greedy order is then by index
    b2 = ((w2 * t2).sum(1) + torch.tensor([0.5, -0.3, 2.0, -1e3, 0.7])).contiguous()
    t2c, w2c, b2c = t2.cuda(), w2.cuda(), b2.cuda()
    for norm, fn in (('L2', A.fab_projection_l2), ('L1', A.fab_projection_l1)):
        d2, rn2 = adv.fab_project(t2c, w2c, b2c, norm)
        y2 = t2c + d2
        assert y2.min() >= -1e-6 and y2.max() <= 1 + 1e-6                                  # inside the box
        resid = ((w2c.double() * y2.double()).sum(1) - b2c.double()).abs().cpu()
        assert (resid[:3] < 2e-3).all() and resid[4] < 2e-3                                # reachable rows sit on the hyperplane
        assert resid[3] > 100                                                               # the unreachable one went to the bounds
        want = fn(t2.double(), w2.double(), b2.double()).float()           # the oracle in fp64: the kernel accumulates in fp64
        wn = want.abs().sum(1) if norm == 'L1' else want.pow(2).sum(1).sqrt()
        torch.testing.assert_close(rn2.cpu(), wn, atol=1e-4, rtol=2e-4)                    # the same minimal step length
        rows = [0, 1, 2, 3] if norm == 'L1' else [0, 1, 2, 3, 4]                            # (tied row: any tie order is a minimiser)
        assert (d2.cpu()[rows] - want[rows]).abs().max() < 5e-5
    netc = make_tinynet().cuda()
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    mean = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()
    std = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()
    f_gpu = lambda z: netc((z - mean) / std)  # noqa: E731
    for norm, eps, tol in (('L2', 1.0, 5e-5), ('L1', 12.0, 2e-4)):
        got = adv.fab_targeted_perturb(f_gpu, x.cuda(), y.cuda(), eps, 6, 3, norm=norm)
        torch.testing.assert_close(got.cpu(), torch.from_numpy(g[f'fabt/{norm}/adv']), atol=tol, rtol=0)


SQUARE_LP_CASES = (('L2', 0.5, 60), ('L2', 2.0, 25), ('L1', 12.0, 60), ('L1', 40.0, 25))


def test_square_l2_l1_match_reference():
    """rart_square_init_lp / rart_square_propose_lp (+ rart_l1_project for L1) driven by adv.square_lp_perturb, with the reference's
    torch random stream replayed (window origins, eta's transposition, per-subset sign rows): the best point of EVERY image after the
    run and perturb()'s output vs the unmodified SquareAttack(norm='L2' / 'L1') (square.py:296-530)."""
    from robustart_amd.noise import adv
    g = np.load(os.path.join(GOLD, 'square_lp_ref.npz'))
    netc = make_tinynet().cuda()
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    mean = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()
    std = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()
    f_gpu = lambda z: netc((z - mean) / std)  # noqa: E731
    for norm, eps, nq in SQUARE_LP_CASES:
        d = A.TorchStreamDraws(0)
        d.reseed()
        got, xb, ind = adv.square_lp_perturb(f_gpu, x.cuda(), y.cuda(), norm, eps, nq, draws=d, _return_best=True)
        assert ind.numel() == len(x)
        torch.testing.assert_close(xb.cpu(), torch.from_numpy(g[f'square/{norm}/{eps}/x_best']), atol=5e-5, rtol=0)
        torch.testing.assert_close(got.cpu(), torch.from_numpy(g[f'square/{norm}/{eps}/adv']), atol=5e-5, rtol=0)
    # native draws at ImageNet size: the ball, the box, and determinism in (seed, sample_offset)
    gen = torch.Generator().manual_seed(4)
    xi = torch.rand(3, 3, 224, 224, generator=gen).cuda()
    wv = torch.randn(3 * 224 * 224, 10, generator=gen).cuda() * 0.01
    lin = lambda z: z.flatten(1) @ wv  # noqa: E731
    yi = lin(xi).argmax(1)
    for norm, eps in (('L2', 3.0), ('L1', 60.0)):
        a, xb, _ = adv.square_lp_perturb(lin, xi, yi, norm, eps, 12, seed=5, sample_offset=9, _return_best=True)
        b = adv.square_lp_perturb(lin, xi, yi, norm, eps, 12, seed=5, sample_offset=9)
        assert torch.equal(a, b)
        r = (xb - xi).flatten(1)
        nr = r.norm(dim=1) if norm == 'L2' else r.abs().sum(1)
        assert (nr <= eps * (1 + 1e-4)).all() and (nr >= eps * 0.5).all() and xb.min() >= -1e-6 and xb.max() <= 1 + 1e-6


def test_l2_l1_attack_edge_batches():
    """FAB-T / Square (L2, L1) and AutoAttack on degenerate batches: every label already wrong (nothing is attacked: the input comes
    back), a batch of one, and a batch where only one image is still correct (the reference squeezes such index tensors, fab_base.py:112,
    square.py:572-574)."""
    from robustart_amd.noise import adv
    gen = torch.Generator().manual_seed(12)
    x = torch.rand(3, 3, 32, 32, generator=gen).cuda()
    wv = torch.randn(3 * 32 * 32, 10, generator=gen).cuda() * 0.05
    lin = lambda z: z.flatten(1) @ wv  # noqa: E731
    y = lin(x).argmax(1)
    wrong = (y + 1) % 10
    for norm, eps in (('L2', 0.5), ('L1', 8.0)):
        assert torch.equal(adv.fab_targeted_perturb(lin, x, wrong, eps, 3, 2, norm=norm), x)
        assert torch.equal(adv.square_lp_perturb(lin, x, wrong, norm, eps, 5, seed=1, sample_offset=0), x)
        one = adv.fab_targeted_perturb(lin, x[:1], y[:1], eps, 3, 2, norm=norm)
        assert one.shape == (1, 3, 32, 32) and torch.isfinite(one).all()
        one = adv.square_lp_perturb(lin, x[:1], y[:1], norm, eps, 5, seed=1, sample_offset=0)
        assert one.shape == (1, 3, 32, 32) and torch.isfinite(one).all()
        mixed = torch.stack([wrong[0], y[1], wrong[2]])
        for out in (adv.fab_targeted_perturb(lin, x, mixed, eps, 3, 2, norm=norm),
                    adv.square_lp_perturb(lin, x, mixed, norm, eps, 5, seed=1, sample_offset=0)):
            assert torch.equal(out[0], x[0]) and torch.equal(out[2], x[2]) and torch.isfinite(out).all()
            r = (out[1] - x[1]).flatten()
            assert (r.norm() if norm == 'L2' else r.abs().sum()) <= eps * (1 + 1e-4)


def test_native_pgd_linf_invariants_at_imagenet_size():
    """BASELINE-size property checks (no oracle run needed): eps-ball, box, determinism, sharding."""
    from robustart_amd.noise import adv
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 7, stride=4), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1),
                              torch.nn.Flatten(), torch.nn.Linear(8, 1000)).cuda().eval()
    x = _rand((8, 3, 224, 224), 1).cuda()
    y = torch.randint(0, 1000, (8,)).cuda()
    eps = 2 / 255
    a = adv.pgd_linf(x, y, net, eps, 3 / 40, 3, seed=5, sample_offset=100)
    assert (a - x).abs().max() <= eps + 1e-7 and a.min() >= 0 and a.max() <= 1
    b = adv.pgd_linf(x, y, net, eps, 3 / 40, 3, seed=5, sample_offset=100)
    assert torch.equal(a, b)
    # sharding invariance of the random start: samples 4..7 as their own "rank"
    s = adv.attack_init_linf(x, eps, True, 5, 100)
    s2 = adv.attack_init_linf(x[4:].contiguous(), eps, True, 5, 104)
    assert torch.equal(s[4:], s2)
    u = (s - x)[(x > eps) & (x < 1 - eps)] / eps
    assert abs(u.mean().item()) < 5e-3 and abs(u.std().item() - 1 / np.sqrt(3)) < 5e-3


def test_pgd_l1_art_matches_oracle_and_stays_in_the_l1_ball():
    """pgd_l1 (ART PGD norm=1; parity unpinned -- ART is absent, the oracle restates its published algorithm): HIP kernels +
    autograd gradients on the tiny CNN vs the oracle with the same injected start; AddNoise('pgd_l1') dispatch; L1-ball
    and box invariants at ImageNet size with the native RNG."""
    from robustart_amd.noise import AddNoise, adv
    g, net = _gold_model()
    netc = make_tinynet({k[4:]: g[k] for k in g.files if k.startswith('net/')}).cuda()
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    B, n = x.shape[0], x[0].numel()
    rs = np.random.RandomState(5)
    se = (rs.exponential(size=(B, n)) * rs.choice([-1.0, 1.0], size=(B, n))).astype(np.float32)
    eps, eps_step, iters = 12.0, 1.5, 6
    rad = np.sqrt(rs.uniform(0, eps ** 2, B)).astype(np.float32)

    def loss_grad(xa, yy):
        z = torch.from_numpy(xa).requires_grad_(True)
        torch.nn.functional.cross_entropy(net(A.normalize(z)), torch.from_numpy(np.asarray(yy))).backward()
        return z.grad.numpy()
    want = A.pgd_l1_art(loss_grad, x.numpy(), y.numpy(), eps, eps_step, iters, se, rad)
    got = adv.pgd_l1(x.cuda(), y.cuda(), netc, eps, 32, eps_step, iters, 16, init_signed_exp=torch.from_numpy(se),
                     init_radius=torch.from_numpy(rad)).cpu()
    torch.testing.assert_close(got, torch.from_numpy(want), atol=2e-5, rtol=0)
    d = (got - x).flatten(1).abs().sum(1)
    assert (d <= eps * (1 + 1e-5)).all() and got.min() >= 0 and got.max() <= 1
    assert (d > 0.5 * eps).all()                                  # the steps (1.5 each) do fill the ball
    # the plugin entry: same registry name / config keys as the reference (add_noise_utils.py:7-18)
    an = AddNoise('pgd_l1')
    an.set_config(model=netc, eps=eps, input_size=32, eps_step=eps_step, max_iter=3, batch_size=16)
    out = an.add_noise(x.cuda(), y.cuda())
    assert out.shape == x.shape and out.is_cuda
    assert ((out.cpu() - x).flatten(1).abs().sum(1) <= eps * (1 + 1e-5)).all()
    # native start at ImageNet size: radius <= eps, clipped to the box, different per sample, reproducible
    x224 = _rand((3, 3, 224, 224), 1).cuda()
    s1 = adv.random_start_l1(x224, 1600.0, seed=3)
    s2 = adv.random_start_l1(x224, 1600.0, seed=3)
    assert torch.equal(s1, s2)
    dn = (s1 - x224).flatten(1).abs().sum(1)
    assert (dn <= 1600.0 * (1 + 1e-5)).all() and (dn > 0).all() and s1.min() >= 0 and s1.max() <= 1
    assert dn[0] != dn[1]
    signs = torch.sign(s1 - x224).flatten(1)
    assert abs(float(signs.mean())) < 0.02                        # random signs


def _f_gpu_and_gold():
    g, net = _gold_model()
    netc = make_tinynet({k[4:]: g[k] for k in g.files if k.startswith('net/')}).cuda()
    mean = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()
    std = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()
    return g, netc, (lambda z: netc((z - mean) / std))


def test_apgd_targeted_matches_reference_golden():
    """APGD-T on the HIP step kernels (rart_apgd_init / rart_apgd_step / rart_logit_loss kind 2 / rart_select_rows) vs the
    reference's own APGDAttack_targeted output (attacks_ref.npz 'apgdt/Linf/adv', autopgd_base.py:571-690): the start
    direction of every target class is drawn lazily from the replayed torch stream for the still-robust subset, exactly
    as tests/test_oracle_golden.py::_apgdt_with_stream does for the oracle."""
    from robustart_amd.noise import adv
    g, netc, f_gpu = _f_gpu_and_gold()
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    torch.random.manual_seed(0)
    got = adv.apgd_targeted_perturb(f_gpu, x.cuda(), y.cuda(), 'Linf', 4 / 255, 8, 3,
                                    init_ts=lambda j, shape: 2 * torch.rand(shape) - 1)
    torch.testing.assert_close(got.cpu(), torch.from_numpy(g['apgdt/Linf/adv']), atol=2e-5, rtol=0)
    assert (got.cpu() - x).abs().max() <= 4 / 255 + 1e-6


AA_CASES = {'standard': (1 / 255, ('apgd-ce', 'apgd-t', 'fab-t', 'square'), 2, 2, 2, 10, 3, 60),
            'reordered': (1 / 255, ('square', 'fab-t', 'apgd-t', 'apgd-ce'), 4, 4, 2, 6, 3, 40),
            'standard_L2': (0.12, ('apgd-ce', 'apgd-t', 'fab-t', 'square'), 2, 2, 2, 6, 3, 40, 'L2'),
            'reordered_L2': (0.12, ('square', 'fab-t', 'apgd-t', 'apgd-ce'), 3, 3, 2, 5, 2, 30, 'L2'),
            'rand': (1 / 255, ('apgd-ce', 'apgd-dlr'), 4, 0, 0, 0, 0, 0, 'Linf', 'rand', 3)}      # version 'rand', eot_iter 20 -> 3


@pytest.mark.parametrize('case', sorted(AA_CASES))
def test_autoattack_linf_orchestrator_matches_reference_golden(case):
    """adv.autoattack_linf (the function AddNoise('autoattack_linf') dispatches to) vs the reference's
    AutoAttack.run_standard_evaluation on the tiny CNN (tests/golden/autoattack_ref.npz; autoattack.py:90-211): robust-flag
    bookkeeping, per-attack robust subset, x_adv[non_robust] update and early exit, in two attack orders, with every
    sub-attack on the HIP kernels and the reference's torch random stream replayed through _overrides['draws']."""
    from robustart_amd.noise import adv
    g, netc, f_gpu = _f_gpu_and_gold()
    ga = np.load(os.path.join(GOLD, 'autoattack_ref.npz'))
    x, y = torch.from_numpy(ga['x']), torch.from_numpy(ga['y'])
    eps, plan, ai, ti, tc, fi, fc, sq = AA_CASES[case][:8]
    norm = AA_CASES[case][8] if len(AA_CASES[case]) > 8 else 'Linf'       # L2: APGD / APGD-T / FAB-T / Square all in their L2 forms
    version = AA_CASES[case][9] if len(AA_CASES[case]) > 9 else 'standard'
    ov = dict(plan=plan, apgd_iter=ai, apgdt_iter=ti, apgdt_classes=tc, fab_iter=fi, fab_classes=fc, square_queries=sq,
              draws=A.TorchStreamDraws(0))
    if version == 'rand':
        ov['eot_iter'] = AA_CASES[case][10]                  # (20 in the reference; the golden run shrank it like the iteration counts)
    got = adv.autoattack_linf(x.cuda(), y.cuda(), netc, norm, eps, version, False, _overrides=ov).cpu()
    want = torch.from_numpy(ga[f'{case}/adv'])
    torch.testing.assert_close(got, want, atol=5e-5, rtol=0)
    robust = (netc((got.cuda() - torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()) /
                   torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()).argmax(1).cpu() == y)
    assert np.array_equal(robust.numpy(), ga[f'{case}/robust'])
    assert torch.equal(got[3], x[3])                    # misclassified from the start: never attacked, never changed
    # native (counter-based) draws through the AddNoise plugin entry: eps-ball, box, reproducible
    from robustart_amd.noise import AddNoise, rng
    an = AddNoise('autoattack_linf')
    an.set_config(model=netc, norm=norm, eps=eps, version=version, verbose=False)
    rng.manual_seed(5)
    a = adv.autoattack_linf(x.cuda(), y.cuda(), netc, norm, eps, version, False,
                            _overrides=dict(apgd_iter=3, apgdt_iter=2, apgdt_classes=2, fab_iter=3, fab_classes=2, square_queries=20, eot_iter=2))
    r = (a.cpu() - x).flatten(1)
    assert ((r.abs().max(1)[0] if norm == 'Linf' else r.norm(dim=1)) <= eps * (1 + 1e-5) + 1e-6).all() and a.min() >= 0 and a.max() <= 1


def test_l1_projection_and_kth_select_kernels():
    """rart_l1_project (bisection + exact segment solve, no sort) vs the reference's sort-based L1_projection outputs
    (tests/golden/apgd_l1_ref.npz, autopgd_base.py:19-83); at ImageNet row size against the oracle and the two
    constraints; rart_row_kth_abs (radix select) exactly equal to sort()[k]; rart_row_count_diff."""
    from robustart_amd import _lib
    from robustart_amd.noise import adv
    gl = np.load(os.path.join(GOLD, 'apgd_l1_ref.npz'))
    x, y = torch.from_numpy(gl['proj/x']).cuda(), torch.from_numpy(gl['proj/y']).cuda()
    for e in (0.5, 4.0, 12.0):
        d = adv.l1_projection(x, y, e)
        torch.testing.assert_close(d.cpu(), torch.from_numpy(gl[f'proj/delta/{e}']), atol=2e-6, rtol=1e-5)
    g = torch.Generator().manual_seed(12)
    n = 3 * 224 * 224
    xb = torch.rand(3, n, generator=g)
    yb = torch.randn(3, n, generator=g) * torch.tensor([0.002, 0.05, 0.5]).view(3, 1)
    for e in (12.0, 60.0):
        want = A.l1_projection(xb, yb, e)
        got = adv.l1_projection(xb.cuda(), yb.cuda(), e).cpu()
        # the reference's own fp32 cumsum over 300k sorted breakpoints carries ~1e-5 relative error in alpha (the kernel
        # sums in fp64); the budget check below shows which side is exact
        assert (got - want).abs().max() <= 1e-4, (got - want).abs().max()
        spent = (yb.double() + got.double()).abs().sum(1)
        active = yb.abs().sum(1) > e                      # rows that had to be projected land ON the sphere
        assert ((spent[active] - e).abs() <= e * 2e-6).all(), spent
        z = xb + yb + got
        assert z.min() >= -1e-6 and z.max() <= 1 + 1e-6
        assert ((yb + got).abs().sum(1) <= e * (1 + 1e-5) + 1e-4).all()
    pt = adv.l1_projection(xb.cuda(), yb.cuda(), 12.0, point_out=True, clamp01=True).cpu()
    # (x + y) + delta is rounded at ulp(x) ~ 6e-8 per element: 150k residues of coordinates projected to zero add up
    assert pt.min() >= 0 and pt.max() <= 1 and ((pt.double() - xb.double()).abs().sum(1) <= 12.0 * (1 + 1e-3)).all()
    # exact k-th smallest |g|
    lib = _lib.load()
    gr = torch.randn(5, n, generator=g)
    gr[1, ::3] = 0.0
    gr[2] = gr[2].round()                                         # many ties
    ks = torch.tensor([0, n - 1, 12345, int(0.8 * n), 77], dtype=torch.int64)
    thr = torch.empty(5, device='cuda')
    grc, ksc = gr.cuda(), ks.cuda()
    _lib.check(lib.rart_row_kth_abs(_lib.ptr(grc), _lib.ptr(ksc), _lib.ptr(thr), 5, n, _lib.stream_ptr()))
    want = gr.abs().sort(-1)[0][torch.arange(5), ks]
    assert torch.equal(thr.cpu(), want)
    cnt = torch.empty(5, device='cuda')
    _lib.check(lib.rart_row_count_diff(_lib.ptr(grc), _lib.ptr(torch.zeros_like(grc)), _lib.ptr(cnt), 5, n, _lib.stream_ptr()))
    assert torch.equal(cnt.cpu(), (gr != 0).sum(1).float())


def test_apgd_l1_matches_reference_golden():
    """APGD with the L1 threat model on the HIP kernels vs the reference's APGDAttack(norm='L1') outputs on the tiny CNN
    (tests/golden/apgd_l1_ref.npz): plain runs with both losses, and the larger-eps schedule with two restarts that
    AutoAttack's L1 'standard' version uses; the reference's torch.randn stream is replayed through `draws`."""
    from robustart_amd.noise import adv
    g, netc, f_gpu = _f_gpu_and_gold()
    gl = np.load(os.path.join(GOLD, 'apgd_l1_ref.npz'))
    x, y = torch.from_numpy(gl['x']), torch.from_numpy(gl['y'])
    for loss in ('ce', 'dlr'):
        torch.random.manual_seed(0)
        got = adv.apgd_l1_perturb(f_gpu, x.cuda(), y.cuda(), 3.0, 25, loss, draws=lambda j, shape: torch.randn(shape)).cpu()
        want = torch.from_numpy(gl[f'perturb/{loss}/adv'])
        print('apgd-l1 %s: max |diff| %.3g' % (loss, (got - want).abs().max().item()))
        torch.testing.assert_close(got, want, atol=1e-4, rtol=0)
        assert ((got - x).abs().flatten(1).sum(1) <= 3.0 * (1 + 1e-4)).all() and got.min() >= 0 and got.max() <= 1
    torch.random.manual_seed(0)
    got = adv.apgd_l1_perturb(f_gpu, x.cuda(), y.cuda(), 2.0, 20, 'ce', n_restarts=2, use_largereps=True,
                              draws=lambda j, shape: torch.randn(shape)).cpu()
    torch.testing.assert_close(got, torch.from_numpy(gl['largereps/ce/adv']), atol=1e-4, rtol=0)
    # native draws at a larger size: L1 ball, box, reproducible with pinned offsets
    xl = _rand((3, 3, 64, 64), 2).cuda()
    yl = torch.zeros(3, dtype=torch.int64).cuda()
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 5, stride=4), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1),
                              torch.nn.Flatten(), torch.nn.Linear(8, 10)).cuda().eval()
    a = adv.apgd_l1_perturb(net, xl, yl, 8.0, 10, 'ce', seed=2, sample_offset=0)
    b = adv.apgd_l1_perturb(net, xl, yl, 8.0, 10, 'ce', seed=2, sample_offset=0)
    assert torch.equal(a, b)
    assert ((a - xl).abs().flatten(1).sum(1) <= 8.0 * (1 + 1e-4)).all() and a.min() >= 0 and a.max() <= 1


def test_fab_refuses_a_bare_bf16_engine_and_switches_when_it_has_the_module():
    """FAB's projections need reference-precision logits (fab_pt.py:102-117).  A bf16 EngineModel built from a
    module runs FAB on the cached 'fp32x' engine of the same module (no warning); a bare bf16 engine (no module to fold from) is
    refused unless allow_bf16_fab=True."""
    import warnings
    from robustart_amd.noise import adv
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import EngineModel
    from robustart_amd.model.resnet_torch import randomize_bn_stats
    torch.manual_seed(0)
    m = randomize_bn_stats(get_model({'type': 'resnet50_official'})).eval()
    f = EngineModel(m, takes_normalized=False)
    assert f.rart_engine.precision == 'bf16'
    ref = f.rart_reference_engine()
    assert ref is not None and ref.precision != 'bf16' and f.rart_reference_engine() is ref          # built once, cached
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 64, 64, generator=g).cuda()
    y = f(x).argmax(1)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        out = adv.fab_targeted_perturb(f, x, y, 4 / 255, 2, 1)
    assert out.shape == x.shape and torch.isfinite(out).all()
    bare = EngineModel(None, takes_normalized=False, engine=f.rart_engine)
    with pytest.raises(RuntimeError, match='reference-precision'):
        adv.fab_targeted_perturb(bare, x, y, 4 / 255, 2, 1)
    out2 = adv.fab_targeted_perturb(bare, x, y, 4 / 255, 2, 1, allow_bf16_fab=True)
    assert out2.shape == x.shape


@pytest.mark.parametrize('norm,eps', [('Linf', 1.0 / 255), ('L2', 0.25)])
def test_autoattack_is_invariant_to_batch_splitting(norm, eps):
    """every sub-attack of AutoAttack receives the still-robust SUBSET of the batch together with its
    rows' GLOBAL sample indices (rart_apgd_init / rart_square_init_linf / rart_rng_signs_f32 `row_samples`), so a sample's random
    starts and Square sign rows do not depend on which other samples survived: autoattack on 16 images == the concatenation of two
    8-image calls (the process-wide sample counter gives the second call the offset 8).  Driven through the HIP ResNet-50 engines,
    whose per-image arithmetic does not depend on the batch (tests/test_outcome_gpu.py::test_b256_matches_small_batches_bit_for_bit).
    Reference: autoattack.py:117-136 (robust-subset bookkeeping), autopgd_base.py:502-503 (re-seeding per perturb() call)."""
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import EngineModel
    from robustart_amd.model.resnet_torch import randomize_bn_stats
    from robustart_amd.noise import adv, rng
    torch.manual_seed(0)
    m = randomize_bn_stats(get_model({'type': 'resnet50_official'})).eval()
    model = EngineModel(m, takes_normalized=True)
    g = torch.Generator().manual_seed(21)
    x = torch.rand(16, 3, 64, 64, generator=g).cuda()
    mean = torch.tensor(A.IMAGENET_MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(A.IMAGENET_STD, device='cuda').view(1, 3, 1, 1)
    y = model((x - mean) / std).argmax(1)
    y[5] = (y[5] + 1) % 1000                         # one image misclassified from the start
    ov = dict(apgd_iter=4, apgdt_iter=3, apgdt_classes=2, fab_iter=3, fab_classes=2, square_queries=25)

    def run(xs, ys):
        return adv.autoattack_linf(xs, ys, model, norm, eps, 'standard', False, seed=3, _overrides=dict(ov))
    rng.manual_seed(3, 0)
    full = run(x, y)
    rng.manual_seed(3, 0)
    halves = torch.cat([run(x[:8].contiguous(), y[:8].contiguous()), run(x[8:].contiguous(), y[8:].contiguous())])
    changed = (full != x).flatten(1).any(1)
    print('autoattack %s: %d of 16 images moved' % (norm, int(changed.sum())))
    assert torch.equal(full[5], x[5]) and int(changed.sum()) >= 1
    assert torch.equal(full, halves)
    # ... and the per-row index tensor is what makes it so: Square's start signs of a 3-image subset at its rows' own indices
    rows = torch.tensor([4, 9, 2], dtype=torch.int64, device='cuda')
    sub = x[rows].contiguous()
    a = adv.apgd_init(sub, norm, eps, seed=7, sample_offset=rows)
    for k, r in enumerate(rows.tolist()):
        b = adv.apgd_init(x[r:r + 1].contiguous(), norm, eps, seed=7, sample_offset=r)
        assert torch.equal(a[k], b[0])


def test_square_sign_rows_come_from_the_device_generator():
    """rart_rng_signs_f32 == the host generator's host_uniform(...) >= 0.5 for the same (seed, sample, stream, index)
    -- Square's L2 / L1 proposals draw their per-image sign rows on the device instead of numpy + a host-to-device copy per query."""
    from robustart_amd import _lib
    from robustart_amd.noise import rng
    lib = _lib.load()
    B, C, seed = 5, 3, 1234567
    rows = torch.tensor([11, 3, 700, 3, 42], dtype=torch.int64, device='cuda')
    out = torch.empty(B, C, device='cuda')
    for it in (0, 17, (1 << 20) + 3):
        _lib.check(lib.rart_rng_signs_f32(_lib.ptr(out), B, C, seed, 0, _lib.ptr(rows), 11, it * 4, _lib.stream_ptr()))
        want = [[1.0 if rng.host_uniform(seed, int(r), 11, it * 4 + c) >= 0.5 else -1.0 for c in range(C)] for r in rows.tolist()]
        assert out.cpu().tolist() == want
        _lib.check(lib.rart_rng_signs_f32(_lib.ptr(out), B, C, seed, 100, None, 11, it * 4, _lib.stream_ptr()))
        want = [[1.0 if rng.host_uniform(seed, 100 + b, 11, it * 4 + c) >= 0.5 else -1.0 for c in range(C)] for b in range(B)]
        assert out.cpu().tolist() == want
    n = torch.empty(3, 1000, device='cuda')
    r3 = torch.tensor([5, 6, 9], dtype=torch.int64, device='cuda')
    _lib.check(lib.rart_rng_normal_rows_f32(_lib.ptr(n), 3, 1000, seed, _lib.ptr(r3), 5, _lib.stream_ptr()))
    ref = torch.empty(5, 1000, device='cuda')
    _lib.check(lib.rart_rng_normal_f32(_lib.ptr(ref), 5, 1000, seed, 5, 5, _lib.stream_ptr()))
    assert torch.equal(n[0], ref[0]) and torch.equal(n[1], ref[1]) and torch.equal(n[2], ref[4])


@pytest.mark.parametrize('norm,eps,tol', [('Linf', 1.5 / 255, 5e-5), ('L2', 0.25, 5e-5), ('L1', 3.0, 2e-4)])
def test_untargeted_fab_and_restarts_match_reference(norm, eps, tol):
    """VERDICT r3 missing item 5: the untargeted `fab` stage of AutoAttack version 'plus' and the random-start restarts of both FAB stages
    (autoattack.py:269-275, fab_pt.py:77-100, fab_base.py:133-186) on the HIP step kernels, against the unmodified reference's outputs on
    the tiny CNN (tests/golden/fab_plus_ref.npz): a single run without / with the random start, the untargeted perturb() with three
    restarts and the targeted perturb() with two restarts per class, the reference's torch draws injected."""
    from robustart_amd.noise import adv
    g = np.load(os.path.join(GOLD, 'fab_plus_ref.npz'))
    net = make_tinynet().cuda()
    mean = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()
    std = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()
    f_gpu = lambda z: net((z - mean) / std)  # noqa: E731
    x, y = torch.from_numpy(g['x']).cuda(), torch.from_numpy(g['y']).cuda()
    prov = adv._Provider(f_gpu, normalize_inside=False)
    run0 = adv._fab_single_run(prov, x, y, None, eps, 5, norm=norm)
    torch.testing.assert_close(run0.cpu(), torch.from_numpy(g[f'fab/{norm}/run0']), atol=tol, rtol=0)
    torch.random.manual_seed(0)
    run1 = adv._fab_single_run(prov, x, y, None, eps, 5, norm=norm, start=dict(seed=0, rows=None, t=lambda shape: A.fab_start_draw(norm, shape)))
    torch.testing.assert_close(run1.cpu(), torch.from_numpy(g[f'fab/{norm}/run1']), atol=tol, rtol=0)
    torch.random.manual_seed(0)
    got = adv.fab_perturb(f_gpu, x, y, eps, 5, 3, norm, False, start_draws=A.fab_start_draw)
    torch.testing.assert_close(got.cpu(), torch.from_numpy(g[f'fab/{norm}/adv']), atol=tol, rtol=0)
    torch.random.manual_seed(0)
    got = adv.fab_perturb(f_gpu, x, y, eps, 5, 2, norm, True, 2, start_draws=A.fab_start_draw)
    torch.testing.assert_close(got.cpu(), torch.from_numpy(g[f'fabt_restarts/{norm}/adv']), atol=tol, rtol=0)
    # native restarts (counter generator at the samples' global indices): in the ball, in the box, reproducible
    a = adv.fab_perturb(f_gpu, x, y, eps, 4, 3, norm, False, seed=5, sample_offset=100)
    b = adv.fab_perturb(f_gpu, x, y, eps, 4, 3, norm, False, seed=5, sample_offset=100)
    assert torch.equal(a, b) and float(a.min()) >= 0 and float(a.max()) <= 1


def test_autoattack_plus_runs_all_six_stages():
    """AutoAttack version 'plus' = [apgd-ce, apgd-dlr, fab, square, apgd-t, fab-t] with 5 restarts (autoattack.py:269-275), shrunk: every
    stage runs (round 3 reported the untargeted `fab` as skipped), nothing warns, the result stays in the eps ball and the box."""
    import warnings
    from robustart_amd.noise import adv, rng
    net = make_tinynet().cuda()
    x = _rand((4, 3, 32, 32), 3).cuda()
    mean = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1).cuda()
    std = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1).cuda()
    y = net((x - mean) / std).argmax(1)
    rng.manual_seed(1, 0)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        xa = adv.autoattack_linf(x, y, net, 'Linf', 2 / 255, 'plus', False, seed=1,
                                 _overrides=dict(apgd_iter=3, apgdt_iter=2, apgdt_classes=2, fab_iter=2, fab_classes=2, fab_restarts=2,
                                                 square_queries=10))
    assert (xa - x).abs().max().item() <= 2 / 255 + 1e-6 and xa.min().item() >= 0 and xa.max().item() <= 1


def test_attack_entry_with_a_device_index_tensor_never_blocks_the_host():
    """VERDICT r5 item 5 / SURVEY 8(b) "no hidden device syncs": pgd_linf through the HIP engine with every row's GLOBAL sample index in a
    DEVICE tensor -- what an adversarial-training iteration passes (EpochSampler.batch_rows) and what AutoAttack's sub-attacks pass -- must
    not read that tensor back: the whole attack runs under torch.cuda.set_sync_debug_mode('error'), which turns any blocking
    device-to-host copy into an exception.  The draws equal those of the contiguous-offset form; a HOST tensor is still range-checked
    (before its upload, where the check is free)."""
    from robustart_amd.noise import adv
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import EngineModel
    from robustart_amd.model.resnet_torch import randomize_bn_stats
    from robustart_amd.train.cls_solver import EpochSampler
    torch.manual_seed(0)
    f = EngineModel(randomize_bn_stats(get_model({'type': 'resnet50_official'})).eval(), takes_normalized=False)
    x = _rand((4, 3, 64, 64), 2).cuda()
    y = torch.randint(0, 1000, (4,)).cuda()
    sampler = EpochSampler(64, 4, 0, 1, seed=3)
    rows = sampler.batch_rows(5, x.device)                         # a view of the epoch's permutation, uploaded once per epoch
    assert rows.is_cuda and rows.tolist() == sampler.batch(5)[0] and sampler.batch_rows(6, x.device).data_ptr() != rows.data_ptr()
    assert sampler.batch_rows(5, x.device).data_ptr() == rows.data_ptr()                       # no second upload inside an epoch
    want = adv.pgd_linf(x, y, f, 2 / 255, 3 / 40, 2, seed=9, sample_offset=rows)                # warm-up: allocations, table packing
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        got = adv.pgd_linf(x, y, f, 2 / 255, 3 / 40, 2, seed=9, sample_offset=rows)
        start = adv.attack_init_linf(x, 2 / 255, True, 9, rows)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    assert torch.equal(got, want)
    ref = torch.cat([adv.attack_init_linf(x[i:i + 1].contiguous(), 2 / 255, True, 9, int(rows[i])) for i in range(4)])
    assert torch.equal(start, ref)
    with pytest.raises(ValueError, match=r'\[0, 2\^32\)'):
        adv.attack_init_linf(x, 2 / 255, True, 9, torch.tensor([0, 1, 2, 1 << 32]))
    with pytest.raises(ValueError, match='one global sample index per row'):
        adv.attack_init_linf(x, 2 / 255, True, 9, rows[:3])
