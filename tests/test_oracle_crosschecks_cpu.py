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


def test_spatter_water_stages_restatements():
    """OpenCV stages of spatter's water branch (unpinned: cv2 absent): the scan formulation of the 5x5 chamfer distance
    equals the literal two-pass raster loops of distransform.cpp; the chamfer distance tracks scipy's exact Euclidean
    transform within the mask's known ~2 % error; Canny of a filled square is its closed one-pixel outline; equalizeHist
    maps the extreme populated bins to 0 / 255; the whole branch runs for severities 1-3."""
    import scipy.ndimage as ndi
    rs = np.random.RandomState(5)

    def dt_loops(src):
        h, w = src.shape
        a, b, c = O.CV_DIST_A, O.CV_DIST_B, O.CV_DIST_C
        T = np.full((h + 4, w + 4), O.CV_DIST_INIT, dtype=np.int64)
        for i in range(h):
            for j in range(w):
                I, J = i + 2, j + 2
                T[I, J] = 0 if src[i, j] == 0 else min(T[I - 2, J - 1] + c, T[I - 2, J + 1] + c, T[I - 1, J - 2] + c,
                                                      T[I - 1, J - 1] + b, T[I - 1, J] + a, T[I - 1, J + 1] + b,
                                                      T[I - 1, J + 2] + c, T[I, J - 1] + a)
        for i in range(h - 1, -1, -1):
            for j in range(w - 1, -1, -1):
                I, J = i + 2, j + 2
                if T[I, J] > a:
                    T[I, J] = min(T[I, J], T[I + 2, J + 1] + c, T[I + 2, J - 1] + c, T[I + 1, J + 2] + c, T[I + 1, J + 1] + b,
                                  T[I + 1, J] + a, T[I + 1, J - 1] + b, T[I + 1, J - 2] + c, T[I, J + 1] + a)
        return np.minimum(T[2:h + 2, 2:w + 2], O.CV_DIST_INIT)
    src = (rs.rand(48, 41) > 0.02).astype(np.uint8) * 255
    fixed = O.cv_distance_transform_l2_5(src)
    assert np.array_equal(fixed, dt_loops(src))
    edt = ndi.distance_transform_edt(src)
    cham = fixed / 65536.0
    assert (np.abs(cham - edt) <= 0.03 * edt + 1e-9).all()
    img = np.zeros((40, 40), np.uint8)
    img[10:30, 12:28] = 200
    e = O.cv_canny_u8(img, 50, 150)
    ys, xs = np.nonzero(e)
    assert e.max() == 255 and ys.min() in (9, 10) and ys.max() in (29, 30) and (e[15:25, 16:24] == 0).all()
    assert ndi.label(e, structure=np.ones((3, 3)))[1] == 1              # one closed contour
    q = rs.randint(3, 9, (30, 30)).astype(np.uint8)
    eq = O.cv_equalize_hist_u8(q)
    assert eq[q == q.min()].max() == 0 and eq[q == q.max()].min() == 255
    x = rs.randint(0, 256, (224, 224, 3)).astype(np.uint8)
    for sev in (1, 2, 3):
        y = O.corrupt('spatter', x, sev, O.draw('spatter', x, sev, np.random.RandomState(sev)))
        assert y.dtype == np.uint8 and 0.005 < (y != x).mean() < 0.5
