"""Generate the committed golden fixtures by running UNMODIFIED reference code
(/root/reference, imported with stubs for absent wheels -- see _ref_import.py).

Run in the build container only:  python tests/golden/make_golden.py
Outputs (data only: inputs are regenerated from seeds by tests/_inputs.py):
  tests/golden/corruptions_ref.npz   sha256 + 64x64 crop of reference outputs, 8 corruptions x 5 severities
  tests/golden/attacks_ref.npz       reference APGD / APGD-T / MIM outputs on tests/_tinynet.TinyNet
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference_noise  # noqa: E402

ref_noise = import_reference_noise()
from PIL import Image  # noqa: E402
from RobustART.noise.utils.imagenet_c import corrupt as ref_corrupt  # noqa: E402
from RobustART.noise.utils.adv.Attacks.autoattack.autopgd_base import APGDAttack, APGDAttack_targeted  # noqa: E402
from RobustART.noise.utils.adv.Attacks.imfgsm_attack import _mim_whitebox  # noqa: E402
from RobustART.noise.utils.adv.Attacks.autoattack.square import SquareAttack  # noqa: E402
from RobustART.noise.utils.adv.Attacks.autoattack.fab_pt import FABAttack_PT  # noqa: E402
from RobustART.noise.utils.adv.Attacks.autoattack.fab_projections import projection_linf  # noqa: E402

from _inputs import RUNNABLE, make_image, case_seed  # noqa: E402
from _tinynet import make_tinynet, make_batch  # noqa: E402
from oracle.attacks_ref import normalize  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_corruptions():
    out = {}
    for name in RUNNABLE:
        for sev in range(1, 6):
            x = make_image(sev)
            np.random.seed(case_seed(name, sev))
            y = np.asarray(ref_corrupt(Image.fromarray(x), severity=sev, corruption_name=name))
            out[f'{name}/{sev}/sha'] = np.array(sha(y))
            out[f'{name}/{sev}/crop'] = y[80:144, 80:144].copy()
    # the AddNoise('imagenet-c') batch path: in-place, per-image loop, one random stream
    A = ref_noise.AddNoise('imagenet-c')
    A.set_config(corruption_name='gaussian_noise', severity=3)
    batch = np.stack([make_image(i) for i in (1, 2, 3)])
    np.random.seed(4242)
    ret = A.add_noise(batch)
    assert ret is batch
    out['batch/gaussian_noise/3/sha'] = np.array(sha(ret))
    out['batch/gaussian_noise/3/crop'] = ret[:, 80:112, 80:112].copy()
    A.set_config(corruption_name=None, corruption_number=1, severity=2)   # shot_noise by index
    batch = np.stack([make_image(i) for i in (4, 5)])
    np.random.seed(4243)
    ret = A.add_noise(batch)
    out['batch/number1/2/sha'] = np.array(sha(ret))
    np.savez_compressed(os.path.join(HERE, 'corruptions_ref.npz'), **out)
    print('corruptions_ref.npz', len(out), 'entries')


def gen_attacks():
    net = make_tinynet()
    model_fn = lambda x: net(normalize(x))  # noqa: E731
    x = make_batch()
    y = model_fn(x).max(1)[1]          # every sample starts "correct"
    out = {f'net/{k}': v.numpy() for k, v in net.state_dict().items()}
    out['x'] = x.numpy()
    out['y'] = y.numpy()
    for norm, eps in (('Linf', 8 / 255), ('L2', 0.5)):
        for loss in ('ce', 'dlr'):
            att = APGDAttack(model_fn, n_iter=10, norm=norm, n_restarts=1, eps=eps, seed=0, loss=loss,
                             device='cpu')
            adv = att.perturb(x.clone(), y.clone())
            out[f'apgd/{norm}/{loss}/adv'] = adv.detach().numpy()
            # single run internals (best-loss point and flags) for a tighter pin
            att.init_hyperparam(x)
            torch.random.manual_seed(0)
            xb, acc, lb, xba = att.attack_single_run(x.clone(), y.clone())
            out[f'apgd/{norm}/{loss}/x_best'] = xb.detach().numpy()
            out[f'apgd/{norm}/{loss}/acc'] = acc.numpy()
            out[f'apgd/{norm}/{loss}/loss_best'] = lb.detach().numpy()
            out[f'apgd/{norm}/{loss}/x_best_adv'] = xba.detach().numpy()
    att = APGDAttack_targeted(model_fn, n_iter=8, norm='Linf', n_restarts=1, eps=4 / 255, seed=0,
                              n_target_classes=3, device='cpu')
    out['apgdt/Linf/adv'] = att.perturb(x.clone(), y.clone()).detach().numpy()
    torch.manual_seed(11)
    adv = _mim_whitebox(net, x.clone(), y.clone(), epsilon=8 / 255, num_steps=5, step_size=0.002,
                        decay_factor=1.0)
    out['mim/adv'] = adv.detach().numpy()
    sq = SquareAttack(model_fn, p_init=.8, n_queries=40, eps=8 / 255, norm='Linf', n_restarts=1, seed=0,
                      resc_schedule=False, device='cpu')
    out['square/Linf/adv'] = sq.perturb(x.clone(), y.clone()).detach().numpy()
    fab = FABAttack_PT(model_fn, n_restarts=1, n_iter=6, eps=8 / 255, seed=0, norm='Linf', verbose=False, device='cpu',
                       targeted=True, n_target_classes=3)
    out['fabt/Linf/adv'] = fab.perturb(x.clone(), y.clone()).detach().numpy()
    gp = torch.Generator().manual_seed(3)
    pt, pw = torch.rand(6, 500, generator=gp), torch.randn(6, 500, generator=gp)
    pw[:, ::7] = 0
    pb = torch.tensor([0.1, 5.0, -3.0, 40.0, -60.0, 0.0]) + (pw * pt).sum(1) * torch.tensor([1, 1, 1, 0, 0, 1.])
    out['fabproj/t'], out['fabproj/w'], out['fabproj/b'] = pt.numpy(), pw.numpy(), pb.numpy()
    out['fabproj/d'] = projection_linf(pt, pw, pb).numpy()
    np.savez_compressed(os.path.join(HERE, 'attacks_ref.npz'), **out)
    print('attacks_ref.npz', len(out), 'entries')


AA_CASES = {   # name -> (eps, plan or None (= the 'standard' order), apgd n_iter, apgd-t n_iter, apgd-t classes, fab n_iter, fab classes, square queries[, norm])
    'standard': (1 / 255, None, 2, 2, 2, 10, 3, 60),
    'reordered': (1 / 255, ['square', 'fab-t', 'apgd-t', 'apgd-ce'], 4, 4, 2, 6, 3, 40),
    'standard_L2': (0.12, None, 2, 2, 2, 6, 3, 40, 'L2'),
    'reordered_L2': (0.12, ['square', 'fab-t', 'apgd-t', 'apgd-ce'], 3, 3, 2, 5, 2, 30, 'L2'),
    'rand': (1 / 255, None, 4, 0, 0, 0, 0, 0, 'Linf', 'rand', 3),       # version 'rand': apgd-ce + apgd-dlr with eot_iter passes (20 -> 3)
}


def aa_inputs():
    net = make_tinynet()
    x = make_batch(n=16, seed=21)
    y = net(normalize(x)).max(1)[1]
    y[3] = (y[3] + 1) % 10                 # one sample is misclassified before any attack runs
    return net, x, y


def gen_autoattack():
    """AutoAttack(model, 'Linf', eps, version='standard', seed=0).run_standard_evaluation of the UNMODIFIED reference
    (autoattack.py:90-211) on the tiny CNN, iteration counts shrunk through the attribute overrides the reference itself
    uses in set_version (autoattack.py:253-267).  `model` takes normalised input (NormalizeModel wraps it; its .cuda()
    calls are identity here, _ref_import.py).  Saved next to attacks_ref.npz as autoattack_ref.npz."""
    from RobustART.noise.utils.adv.Attacks.autoattack.autoattack import AutoAttack
    net, x, y = aa_inputs()
    out = {'x': x.numpy(), 'y': y.numpy()}
    for name, case in AA_CASES.items():
        eps, plan, ai, ti, tc, fi, fc, sq = case[:8]
        version = case[9] if len(case) > 9 else 'standard'
        aa = AutoAttack(net, norm=case[8] if len(case) > 8 else 'Linf', eps=eps, seed=0, verbose=False, version=version, device='cpu')
        if version == 'rand':
            aa.apgd.eot_iter = case[10]
        aa.apgd.n_iter = ai
        aa.apgd_targeted.n_iter, aa.apgd_targeted.n_target_classes = ti, tc
        aa.fab.n_iter, aa.fab.n_target_classes = fi, fc
        aa.square.n_queries = sq
        if plan is not None:
            aa.attacks_to_run = list(plan)
        adv = aa.run_standard_evaluation(x.clone(), y.clone(), bs=len(x))
        out[f'{name}/adv'] = adv.detach().numpy()
        out[f'{name}/robust'] = (net(normalize(adv)).max(1)[1] == y).numpy()
        # the same run one attack at a time (run_standard_evaluation_individual, autoattack.py:228-249)
        aa.attacks_to_run = list(plan) if plan is not None else (['apgd-ce', 'apgd-dlr'] if version == 'rand' else ['apgd-ce', 'apgd-t', 'fab-t', 'square'])
        indiv = aa.run_standard_evaluation_individual(x.clone(), y.clone(), bs=len(x))
        for k, v in indiv.items():
            out[f'{name}/individual/{k}'] = v.detach().numpy()
        print(name, 'robust after the ensemble:', int(out[f'{name}/robust'].sum()), 'of', len(x), '| changed rows',
              ((adv - x).abs().flatten(1).max(1)[0] > 0).nonzero().flatten().tolist())
    np.savez_compressed(os.path.join(HERE, 'autoattack_ref.npz'), **out)
    print('autoattack_ref.npz', len(out), 'entries')


def gen_apgd_l1():
    """L1_projection itself on random rows (some already inside the ball / box, some far outside) and APGDAttack with
    norm='L1' (plain and with the larger-eps schedule of AutoAttack's L1 'standard' version) -> apgd_l1_ref.npz."""
    from RobustART.noise.utils.adv.Attacks.autoattack.autopgd_base import L1_projection
    out = {}
    g = torch.Generator().manual_seed(8)
    px = torch.rand(8, 3, 10, 10, generator=g)
    py = torch.randn(8, 3, 10, 10, generator=g) * torch.tensor([0.001, 0.01, 0.05, 0.2, 0.5, 1.0, 2.0, 0.0]).view(8, 1, 1, 1)
    py[6, :, :5] = 0
    out['proj/x'], out['proj/y'] = px.numpy(), py.numpy()
    for e in (0.5, 4.0, 12.0):
        out[f'proj/delta/{e}'] = L1_projection(px, py, e).numpy()
    net = make_tinynet()
    model_fn = lambda z: net(normalize(z))  # noqa: E731
    x = make_batch(n=6, seed=33)
    y = model_fn(x).max(1)[1]
    out['x'], out['y'] = x.numpy(), y.numpy()
    for loss in ('ce', 'dlr'):
        att = APGDAttack(model_fn, n_iter=25, norm='L1', n_restarts=1, eps=3.0, seed=0, loss=loss, device='cpu')
        att.init_hyperparam(x)
        torch.random.manual_seed(0)
        xb, acc, lb, xba = att.attack_single_run(x.clone(), y.clone())
        out[f'single/{loss}/x_best'], out[f'single/{loss}/acc'] = xb.detach().numpy(), acc.numpy()
        out[f'single/{loss}/loss_best'], out[f'single/{loss}/x_best_adv'] = lb.detach().numpy(), xba.detach().numpy()
        out[f'perturb/{loss}/adv'] = att.perturb(x.clone(), y.clone()).detach().numpy()
    att = APGDAttack(model_fn, n_iter=20, norm='L1', n_restarts=2, eps=2.0, seed=0, loss='ce', device='cpu', use_largereps=True)
    out['largereps/ce/adv'] = att.perturb(x.clone(), y.clone()).detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'apgd_l1_ref.npz'), **out)
    print('apgd_l1_ref.npz', len(out), 'entries; robust after the plain ce run:',
          int((model_fn(torch.from_numpy(out['perturb/ce/adv'])).max(1)[1] == y).sum()), 'of', len(x))


def gen_fab_l2_l1():
    """projection_l2 / projection_l1 themselves (fab_projections.py:62-166) on random rows -- hyperplanes near, far and out of reach,
    zero and tiny gradient entries, points on the box faces -- and FABAttack_PT, targeted, norm='L2' / 'L1', on the tiny CNN
    -> fab_l2_l1_ref.npz."""
    from RobustART.noise.utils.adv.Attacks.autoattack.fab_projections import projection_l2, projection_l1
    out = {}
    gp = torch.Generator().manual_seed(5)
    pt, pw = torch.rand(8, 600, generator=gp), torch.randn(8, 600, generator=gp)
    pw[:, ::7] = 0
    pw[:, 3::11] *= 1e-9                    # below the |w| > 1e-8 gate
    pt[:, 5::13] = 0.0                      # on a face of the box
    pt[:, 6::17] = 1.0
    pw[6] *= 1e-3
    shift = torch.tensor([0.1, 5.0, -3.0, 400.0, -600.0, 0.0, 0.01, -40.0])
    pb = (pw * pt).sum(1) + shift
    out['proj/t'], out['proj/w'], out['proj/b'] = pt.numpy(), pw.numpy(), pb.numpy()
    out['proj/d_l2'] = projection_l2(pt.clone(), pw.clone(), pb.clone()).numpy()
    out['proj/d_l1'] = projection_l1(pt.clone(), pw.clone(), pb.clone()).numpy()
    net = make_tinynet()
    model_fn = lambda z: net(normalize(z))  # noqa: E731
    x = make_batch()
    y = model_fn(x).max(1)[1]
    out['x'], out['y'] = x.numpy(), y.numpy()
    for norm, eps in (('L2', 1.0), ('L1', 12.0)):
        fab = FABAttack_PT(model_fn, n_restarts=1, n_iter=6, eps=eps, seed=0, norm=norm, verbose=False, device='cpu',
                           targeted=True, n_target_classes=3)
        adv = fab.perturb(x.clone(), y.clone()).detach()
        out[f'fabt/{norm}/adv'] = adv.numpy()
        print('fab-t', norm, 'robust', int((model_fn(adv).max(1)[1] == y).sum()), 'of', len(x))
    np.savez_compressed(os.path.join(HERE, 'fab_l2_l1_ref.npz'), **out)
    print('fab_l2_l1_ref.npz', len(out), 'entries')


def gen_square_lp():
    """SquareAttack with norm='L2' / 'L1' (square.py:296-530) on the tiny CNN -> square_lp_ref.npz."""
    net = make_tinynet()
    model_fn = lambda z: net(normalize(z))  # noqa: E731
    x = make_batch(n=6, seed=41)
    y = model_fn(x).max(1)[1]
    out = {'x': x.numpy(), 'y': y.numpy()}
    for norm, eps, nq in (('L2', 0.5, 60), ('L2', 2.0, 25), ('L1', 12.0, 60), ('L1', 40.0, 25)):
        sq = SquareAttack(model_fn, p_init=.8, n_queries=nq, eps=eps, norm=norm, n_restarts=1, seed=0, resc_schedule=False,
                          device='cpu')
        adv = sq.perturb(x.clone(), y.clone()).detach()
        out[f'square/{norm}/{eps}/adv'] = adv.numpy()
        # the single run itself: the best point of EVERY image (perturb() only returns the fooled ones), same seed as perturb()
        sq.init_hyperparam(x)
        torch.random.manual_seed(0)
        _, x_best = sq.attack_single_run(x.clone(), y.clone())
        out[f'square/{norm}/{eps}/x_best'] = x_best.detach().numpy()
        print('square', norm, eps, 'robust', int((model_fn(adv).max(1)[1] == y).sum()), 'of', len(x))
    np.savez_compressed(os.path.join(HERE, 'square_lp_ref.npz'), **out)
    print('square_lp_ref.npz', len(out), 'entries')


def gen_fab_plus():
    """FABAttack_PT as AutoAttack version 'plus' configures it (autoattack.py:269-275: fab.n_restarts = 5 for BOTH the untargeted `fab` stage
    and `fab-t`): the UNTARGETED attack (full Jacobian, fab_pt.py:77-100; the closest linearised class boundary, fab_base.py:168-186) and
    random-start restarts (fab_base.py:133-166), shrunk to 2-3 restarts / 5 iterations on the 10-class tiny CNN, Linf / L2 / L1
    -> fab_plus_ref.npz."""
    net = make_tinynet()
    model_fn = lambda z: net(normalize(z))  # noqa: E731
    x = make_batch()
    y = model_fn(x).max(1)[1]
    out = {'x': x.numpy(), 'y': y.numpy()}
    for norm, eps in (('Linf', 1.5 / 255), ('L2', 0.25), ('L1', 3.0)):        # radii at which some images survive the first restart
        fab = FABAttack_PT(model_fn, n_restarts=3, n_iter=5, eps=eps, seed=0, norm=norm, verbose=False, device='cpu', targeted=False)
        adv = fab.perturb(x.clone(), y.clone()).detach()
        out[f'fab/{norm}/adv'] = adv.numpy()
        print('fab (untargeted, 3 restarts)', norm, 'robust', int((model_fn(adv).max(1)[1] == y).sum()), 'of', len(x))
        # one run without and one with the random start, every image's result (perturb() keeps only the fooled ones)
        torch.random.manual_seed(0)
        out[f'fab/{norm}/run0'] = fab.attack_single_run(x.clone(), y.clone(), use_rand_start=False, is_targeted=False).detach().numpy()
        torch.random.manual_seed(0)
        out[f'fab/{norm}/run1'] = fab.attack_single_run(x.clone(), y.clone(), use_rand_start=True, is_targeted=False).detach().numpy()
        fabt = FABAttack_PT(model_fn, n_restarts=2, n_iter=5, eps=eps, seed=0, norm=norm, verbose=False, device='cpu', targeted=True,
                            n_target_classes=2)
        advt = fabt.perturb(x.clone(), y.clone()).detach()
        out[f'fabt_restarts/{norm}/adv'] = advt.numpy()
        print('fab-t (2 restarts x 2 classes)', norm, 'robust', int((model_fn(advt).max(1)[1] == y).sum()), 'of', len(x))
    np.savez_compressed(os.path.join(HERE, 'fab_plus_ref.npz'), **out)
    print('fab_plus_ref.npz', len(out), 'entries')


if __name__ == '__main__':
    if 'fab_plus' in sys.argv[1:]:
        gen_fab_plus()
    elif 'square_lp' in sys.argv[1:]:
        gen_square_lp()
    elif 'fab_l2_l1' in sys.argv[1:]:
        gen_fab_l2_l1()
    elif 'apgd_l1' in sys.argv[1:]:
        gen_apgd_l1()
    elif 'autoattack' in sys.argv[1:]:
        gen_autoattack()
    else:
        gen_corruptions()
        gen_attacks()
        gen_autoattack()
        gen_apgd_l1()
        gen_fab_l2_l1()
        gen_square_lp()
        gen_fab_plus()
