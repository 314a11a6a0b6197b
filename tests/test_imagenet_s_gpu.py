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
    a.set_config(resize_type='opencv-bilinear')
    with pytest.raises(NotImplementedError):
        a.add_noise(path)
    a.set_config(resize_type='pil-bilinear', decoder_type='ffmpeg')
    with pytest.raises(NotImplementedError):
        a.add_noise(path)
    a.set_config(decoder_type='pil', transform_type='train')
    assert a.add_noise(path).shape == (224, 224, 3)
