"""GPU parity: ImageNet-S resize operators (rart_pil_resize_u8) vs the oracle (itself bit-exact vs Pillow)."""
import io

import numpy as np
import pytest
import torch

from _inputs import make_image
from oracle import resize_np as R

pytestmark = pytest.mark.gpu
FID = {'nearest': 0, 'bilinear': 1, 'bicubic': 2, 'box': 3, 'hamming': 4, 'lanczos': 5}


@pytest.mark.parametrize('name', R.FILTERS)
def test_resize_bit_exact(name):
    from robustart_amd.noise import imagenet_s as S
    for (h, w, oh, ow) in [(333, 500, 256, 256), (100, 80, 256, 256), (500, 333, 117, 301), (224, 224, 224, 224)]:
        batch = np.stack([make_image(5, h, w), make_image(6, h, w)])
        got = S.pil_resize(torch.from_numpy(batch).cuda(), (oh, ow), FID[name]).cpu().numpy()
        for i in range(2):
            np.testing.assert_array_equal(got[i], R.pil_resize_u8(batch[i], oh, ow, name), err_msg=str((name, h, w, oh, ow)))
        # fused crop == crop of the full result
        cy, cx, ch, cw = oh // 7, ow // 5, oh // 2, ow // 2
        c = S.pil_resize(torch.from_numpy(batch).cuda(), (oh, ow), FID[name], crop=(cy, cx, ch, cw)).cpu().numpy()
        np.testing.assert_array_equal(c, got[:, cy:cy + ch, cx:cx + cw])


def test_addnoise_imagenet_s_val_path(tmp_path):
    """AddNoise('imagenet-s') on a file path, every pil-* operator, vs the oracle and vs Pillow itself."""
    from PIL import Image
    from robustart_amd.noise import AddNoise
    x = make_image(9, 375, 500)
    path = str(tmp_path / 'img.png')
    Image.fromarray(x).save(path)
    pil_const = {'pil-bilinear': Image.BILINEAR, 'pil-nearest': Image.NEAREST, 'pil-box': Image.BOX,
                 'pil-hamming': Image.HAMMING, 'pil-cubic': Image.BICUBIC, 'pil-lanczos': Image.LANCZOS}
    for rt, const in pil_const.items():
        a = AddNoise('imagenet-s')
        a.set_config(resize_type=rt)
        out = a.add_noise(path)
        assert out.shape == (224, 224, 3) and out.dtype == np.uint8
        np.testing.assert_array_equal(out, R.imagenet_s_val(x, rt))
        ref = np.asarray(Image.fromarray(x).resize((256, 256), const).crop((16, 16, 240, 240)))
        np.testing.assert_array_equal(out, ref)
    a = AddNoise('imagenet-s')
    a.set_config(resize_type='pil-bilinear', decoder_type='ffmpeg')
    with pytest.raises(NotImplementedError):
        a.add_noise(path)
    a.set_config(decoder_type='pil', transform_type='train')
    assert a.add_noise(path).shape == (224, 224, 3)


@pytest.mark.gpu
@pytest.mark.parametrize('hw,dst', [((375, 500), (256, 256)), ((512, 512), (256, 256)), ((300, 768), (256, 256)),
                                    ((180, 200), (256, 256)), ((200, 333), (100, 111)), ((64, 64), (224, 224))])
def test_opencv_resize_operators_match_the_oracle(hw, dst):
    """The five opencv-* operators (parity unpinned: cv2 is absent; oracle/resize_cv_np.py restates resize.cpp):
    HIP == oracle bit for bit, for down-, up- and mixed scaling, the exact-2x LINEAR->AREA switch, integer AREA."""
    import torch
    from oracle import resize_cv_np as CV
    from robustart_amd.noise import imagenet_s as S
    x = make_image(hw[0] * 7 + hw[1], hw[0], hw[1])
    dev = torch.from_numpy(x[None]).cuda()
    for name, interp in S.CV_MODES.items():
        want = CV.resize(x, (dst[1], dst[0]), interp)
        got = S.cv_resize(dev, dst, interp)[0].cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg='%s %s->%s' % (name, hw, dst))
        # fused crop == crop of the full result
        cy, cx, ch, cw = dst[0] // 8, dst[1] // 7, dst[0] // 2, dst[1] // 3
        part = S.cv_resize(dev, dst, interp, crop=(cy, cx, ch, cw))[0].cpu().numpy()
        np.testing.assert_array_equal(part, want[cy:cy + ch, cx:cx + cw])


@pytest.mark.gpu
def test_opencv_resize_properties_and_plugin(tmp_path):
    import torch
    from PIL import Image
    from oracle import resize_cv_np as CV
    from robustart_amd.noise import AddNoise, imagenet_s as S
    # constants are preserved by every operator; identity size is the identity for the interpolating ones
    flat = np.full((90, 130, 3), 77, np.uint8)
    dev = torch.from_numpy(flat[None]).cuda()
    for interp in range(5):
        assert int(S.cv_resize(dev, (256, 256), interp).min()) == 77 == int(S.cv_resize(dev, (256, 256), interp).max())
    x = make_image(3, 120, 160)
    dev = torch.from_numpy(x[None]).cuda()
    for interp in range(5):
        np.testing.assert_array_equal(S.cv_resize(dev, (120, 160), interp)[0].cpu().numpy(), x)
    # exact 2x2 decimation: LINEAR and AREA coincide ((a+b+c+d+2)>>2)
    a = S.cv_resize(dev, (60, 80), 1)[0].cpu().numpy()
    b = S.cv_resize(dev, (60, 80), 3)[0].cpu().numpy()
    ref = (x.astype(np.int64).reshape(60, 2, 80, 2, 3).sum((1, 3)) + 2) >> 2
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, ref.astype(np.uint8))
    # plugin path: AddNoise('imagenet-s') with an opencv-* operator == the oracle's val transform
    big = make_image(9, 375, 500)
    path = str(tmp_path / 'img.png')
    Image.fromarray(big).save(path)
    for rt, interp in S.CV_MODES.items():
        an = AddNoise('imagenet-s')
        an.set_config(resize_type=rt)
        out = an.add_noise(path)
        assert out.shape == (224, 224, 3) and out.dtype == np.uint8
        np.testing.assert_array_equal(out, CV.imagenet_s_val(big, interp))
    an.set_config(resize_type='opencv-area', transform_type='train')
    assert an.add_noise(path).shape == (224, 224, 3)
