"""Noise registry: mirrors RobustART/noise/utils/add_noise_utils.py:7-50 (same names, same order,
same default values, including the f_model / model key asymmetry)."""
from PIL import Image
import numpy as np

from .imagenet_c import corrupt
from .adv import pgd_l1, pgd_l2, pgd_linf, autoattack_linf, mim_linf, fgsm

noise_list = ['imagenet-s', 'imagenet-c', 'pgd_linf', 'pgd_l2', 'fgsm', 'autoattack_linf', 'mim_linf', 'pgd_l1']

default_config = {
    'imagenet-s': {'decoder_type': 'pil', 'resize_type': 'pil-bilinear', 'transform_type': 'val'},
    'imagenet-c': {'severity': 1, 'corruption_name': None, 'corruption_number': -1},
    'pgd_linf': {'f_model': None, 'eps': 8 / 255, 'rel_stepsize': 3 / 40, 'steps': 20},
    'pgd_l2': {'f_model': None, 'eps': 8.0, 'rel_stepsize': 3 / 40, 'steps': 20},
    'fgsm': {'f_model': None, 'eps': 8 / 255},
    'autoattack_linf': {'model': None, 'norm': 'Linf', 'eps': 8 / 255, 'version': 'standard', 'verbose': False},
    'mim_linf': {'model': None, 'eps': 8 / 255, 'num_steps': 20, 'step_size': 0.002, 'decay_factor': 1.0},
    'pgd_l1': {'model': None, 'eps': 1600.0, 'input_size': 224, 'eps_step': 120, 'max_iter': 20, 'batch_size': 16},
}


def add_noise_for_imagenet_c(image, severity=1, corruption_name=None, corruption_number=-1):
    """add_noise_utils.py:22-31.  A file path is opened with PIL (the documented README usage,
    which the reference's inverted assert makes unreachable); batches are corrupted in place."""
    if isinstance(image, str):
        img = np.asarray(Image.open(image, 'r').convert('RGB'))
        return corrupt(img, severity=severity, corruption_name=corruption_name,
                       corruption_number=corruption_number)
    return corrupt(image, severity=severity, corruption_name=corruption_name,
                   corruption_number=corruption_number)


def add_noise_for_imagenet_s(image, decoder_type='pil', resize_type='pil-bilinear', transform_type='val'):
    """add_noise_utils.py:34-38 -> ImageTransfer(file_path=image, ..., return_online=True).getimage()."""
    assert isinstance(image, str), "Input of imagenet-S can only be file path"
    from .imagenet_s import image_transfer
    return image_transfer(image, decoder_type=decoder_type, resize_type=resize_type, transform_type=transform_type)


function_dict = {
    'imagenet-s': add_noise_for_imagenet_s,
    'imagenet-c': add_noise_for_imagenet_c,
    'pgd_l1': pgd_l1,
    'pgd_linf': pgd_linf,
    'pgd_l2': pgd_l2,
    'fgsm': fgsm,
    'autoattack_linf': autoattack_linf,
    'mim_linf': mim_linf,
}
