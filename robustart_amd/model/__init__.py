"""Model zoo entry point -- mirrors RobustART/model/__init__.py:1 (`get_model`).

Only the two architectures BASELINE.json names are in scope:
`resnet50_official` (forward + backward-to-input HIP engine: engine.py) and `vit_base` / `vit_b16_224`
(forward HIP engine: vit_engine.py)."""
from .resnet_torch import resnet50
from .vit_torch import vit_base

_REGISTRY = {'resnet50_official': resnet50, 'resnet50': resnet50, 'vit_base': vit_base, 'vit_b16_224': vit_base,
             'vit_base_patch16_224': vit_base}


def get_model(config):
    """config: mapping with `type` and optional `kwargs` (the YAML `model:` block,
    exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml:1-6)."""
    mtype = config['type'] if isinstance(config, dict) else config.type
    kwargs = dict((config.get('kwargs') if isinstance(config, dict) else getattr(config, 'kwargs', None)) or {})
    kwargs.pop('bn', None)          # {use_sync_bn: False}: BN statistics are local
    if mtype not in _REGISTRY:
        raise NotImplementedError('model type %r is outside the hot-path scope (ResNet-50 / ViT-B/16 only)' % mtype)
    return _REGISTRY[mtype](**kwargs)
