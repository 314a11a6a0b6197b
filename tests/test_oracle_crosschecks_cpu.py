"""Independent cross-checks of oracle restatements whose reference dependency is absent from this image (VERDICT r1:
"where a present library computes the same function, a cheap cross-check is still missing")."""
import colorsys

import numpy as np
import torch

from oracle import attacks_ref as A
from oracle import corruptions_np as O


def test_rgb_hsv_restatement_agrees_with_colorsys():
    """oracle rgb2hsv / hsv2rgb (skimage.color semantics restated; brightness and saturate, corruptions.py:262-275,
    409-424) against the standard library's colorsys on random and edge-case pixels."""
    rs = np.random.RandomState(0)
    px = np.concatenate([rs.rand(4000, 3), np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0, 1, 0], [0, 0, 1],
                                                      [1, 1, 0], [0.2, 0.2, 0.7], [0.7, 0.2, 0.2]], dtype=np.float64)])
    hsv = O.rgb2hsv(px.reshape(-1, 1, 3)).reshape(-1, 3)
    want = np.array([colorsys.rgb_to_hsv(*p) for p in px])
    np.testing.assert_allclose(hsv, want, atol=1e-12)
    back = O.hsv2rgb(hsv.reshape(-1, 1, 3)).reshape(-1, 3)
    want_back = np.array([colorsys.hsv_to_rgb(*h) for h in hsv])
    np.testing.assert_allclose(back, want_back, atol=1e-12)
    np.testing.assert_allclose(back, px, atol=1e-12)


def _duchi_l1_ball(v, z):
    """Euclidean projection of v onto the L1 ball of radius z (Duchi et al. 2008, sort-based)."""
    if np.abs(v).sum() <= z:
        return v.copy()
    u = np.sort(np.abs(v))[::-1]
    css = np.cumsum(u)
    rho = np.nonzero(u * np.arange(1, len(u) + 1) > (css - z))[0][-1]
    theta = (css[rho] - z) / (rho + 1.0)
    return np.sign(v) * np.maximum(np.abs(v) - theta, 0)


def test_l1_projection_agrees_with_duchi_when_the_box_is_inactive():
    """The reference's L1_projection (restated in oracle.attacks_ref.l1_projection and pinned to it by golden fixtures) is
    the exact projection onto ball-intersect-box; far from the box faces it must equal the classical L1-ball projection."""
    rs = np.random.RandomState(1)
    x = np.full((6, 400), 0.5, dtype=np.float32)
    y = (rs.randn(6, 400) * np.array([0.001, 0.01, 0.02, 0.05, 0.05, 0.1]).reshape(6, 1)).astype(np.float32)
    for eps in (0.5, 2.0, 5.0):
        d = A.l1_projection(torch.from_numpy(x).view(6, 1, 20, 20), torch.from_numpy(y).view(6, 1, 20, 20), eps).view(6, -1).numpy()
        for r in range(6):
            want = _duchi_l1_ball(y[r].astype(np.float64), eps)
            np.testing.assert_allclose(y[r] + d[r], want, atol=2e-6)


def test_art_l1_scaling_projection_stays_in_the_ball_but_is_not_the_exact_projection():
    """F11 (ART PGD norm=1, unpinned): the step's projection is ART's SCALING onto the L1 ball; it satisfies the constraint
    and differs from the exact (Duchi) projection -- documenting that the oracle restates ART's choice, not the optimum."""
    rs = np.random.RandomState(2)
    x0 = rs.rand(3, 50).astype(np.float32)
    grad = rs.randn(3, 50).astype(np.float32)
    se = (rs.exponential(size=(3, 50)) * rs.choice([-1.0, 1.0], size=(3, 50))).astype(np.float32)
    x1 = A.pgd_l1_art(lambda xa, yy: grad, x0, np.zeros(3, dtype=np.int64), 2.0, 5.0, 1, se,
                      np.array([0.1, 0.5, 1.0], dtype=np.float32))
    d = x1 - x0
    assert (np.abs(d).sum(1) <= 2.0 * (1 + 1e-5)).all() and x1.min() >= 0 and x1.max() <= 1
