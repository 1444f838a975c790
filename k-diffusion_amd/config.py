"""Config loading and model factories for the sampling path.

Mirrors the call surface of k_diffusion/config.py: ``load_config`` (:23-146, JSON file, dict, or the
``config`` entry of a safetensors checkpoint's metadata), ``make_model`` (:149-213, the
image_transformer_v2 branch) and ``make_denoiser_wrapper`` (:216-231).  The merged dict has the
same keys and defaults as the reference's, so configs and checkpoints are interchangeable.  The
U-Net / transformer-v1 model families and ``make_sample_density`` (training) are out of scope.
"""
import json
from functools import partial
from pathlib import Path

from . import layers, models, utils

# section -> key -> default, per model type; applied under the user's config (user wins)
_COMMON = {
    'model': dict(sigma_data=1., dropout_rate=0., augment_prob=0., loss_config='karras', loss_weighting='karras', loss_scales=1),
    'dataset': dict(type='imagefolder', num_classes=0, cond_dropout_rate=0.1),
    'optimizer': dict(type='adamw', lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=1e-4),
    'lr_sched': dict(type='constant', warmup=0.),
    'ema_sched': dict(type='inverse', power=0.6667, max_value=0.9999),
}
_BY_TYPE = {
    'image_v1': {
        'model': dict(patch_size=1, augment_wrapper=True, mapping_cond_dim=0, unet_cond_dim=0, cross_cond_dim=0,
                      cross_attn_depths=None, skip_stages=0, has_variance=False),
        'optimizer': dict(type='adamw', lr=1e-4, betas=[0.95, 0.999], eps=1e-6, weight_decay=1e-3),
    },
    'image_transformer_v1': {
        'model': dict(d_ff=0, augment_wrapper=False, skip_stages=0, has_variance=False),
        'optimizer': dict(type='adamw', lr=5e-4, betas=[0.9, 0.99], eps=1e-8, weight_decay=1e-4),
    },
    'image_transformer_v2': {
        'model': dict(mapping_width=256, mapping_depth=2, mapping_d_ff=None, mapping_cond_dim=0, mapping_dropout_rate=0.,
                      d_ffs=None, self_attns=None, dropout_rate=None, augment_wrapper=False, skip_stages=0, has_variance=False),
        'optimizer': dict(type='adamw', lr=5e-4, betas=[0.9, 0.99], eps=1e-8, weight_decay=1e-4),
    },
}


def _overlay(base, head):
    """jsonmerge's default strategy: objects merge recursively, anything else is overwritten."""
    if not (isinstance(base, dict) and isinstance(head, dict)):
        return head
    out = dict(base)
    for k, v in head.items():
        out[k] = _overlay(out[k], v) if k in out else v
    return out


def _round_to_power_of_two(x, tol):
    import math
    cands = [round(x / 2 ** i) * 2 ** i for i in range(math.ceil(math.log2(x)))]
    for c in reversed(cands):
        if abs((c - x) / x) <= tol:
            return c
    return cands[0]


def load_config(path_or_dict):
    if isinstance(path_or_dict, dict):
        config = path_or_dict
    else:
        file = Path(path_or_dict)
        if file.suffix == '.safetensors':
            config = json.loads(utils.get_safetensors_metadata(file)['config'])
        else:
            config = json.loads(file.read_text())
    kind = config['model']['type']
    if kind in _BY_TYPE:
        config = _overlay(_BY_TYPE[kind], config)
    m = config['model']
    if kind == 'image_transformer_v1' and not m['d_ff']:
        m['d_ff'] = _round_to_power_of_two(m['width'] * 8 / 3, tol=0.05)
    if kind == 'image_transformer_v2':
        n = len(m['widths'])
        if not m['mapping_d_ff']:
            m['mapping_d_ff'] = m['mapping_width'] * 3
        if not m['d_ffs']:
            m['d_ffs'] = [w * 3 for w in m['widths']]
        if not m['self_attns']:
            local = {"type": "neighborhood", "d_head": 64, "kernel_size": 7}
            m['self_attns'] = [dict(local) for _ in range(n - 1)] + [{"type": "global", "d_head": 64}]
        if m['dropout_rate'] is None:
            m['dropout_rate'] = [0.0] * n
        elif isinstance(m['dropout_rate'], float):
            m['dropout_rate'] = [m['dropout_rate']] * n
    return _overlay(_COMMON, config)


def _attention_spec(sa):
    v2 = models.image_transformer_v2
    kind = sa['type']
    if kind == 'global':
        return v2.GlobalAttentionSpec(sa.get('d_head', 64))
    if kind == 'neighborhood':
        return v2.NeighborhoodAttentionSpec(sa.get('d_head', 64), sa.get('kernel_size', 7))
    if kind == 'shifted-window':
        return v2.ShiftedWindowAttentionSpec(sa.get('d_head', 64), sa['window_size'])
    if kind == 'none':
        return v2.NoAttentionSpec()
    raise ValueError(f'unsupported self attention type {kind}')


def make_model(config):
    num_classes = config['dataset']['num_classes']
    m = config['model']
    if m['type'] != 'image_transformer_v2':
        raise ValueError(f'unsupported model type {m["type"]}: only image_transformer_v2 is on the MI355X sampling hot path')
    v2 = models.image_transformer_v2
    per_level = (m['depths'], m['widths'], m['d_ffs'], m['self_attns'], m['dropout_rate'])
    assert all(len(p) == len(m['widths']) for p in per_level)
    levels = [v2.LevelSpec(depth, width, d_ff, _attention_spec(sa), dropout) for depth, width, d_ff, sa, dropout in zip(*per_level)]
    mapping = v2.MappingSpec(m['mapping_depth'], m['mapping_width'], m['mapping_d_ff'], m['mapping_dropout_rate'])
    return v2.ImageTransformerDenoiserModelV2(
        levels=levels, mapping=mapping, in_channels=m['input_channels'], out_channels=m['input_channels'],
        patch_size=m['patch_size'], num_classes=num_classes + 1 if num_classes else 0, mapping_cond_dim=m['mapping_cond_dim'])


def make_denoiser_wrapper(config):
    m = config['model']
    sigma_data, has_variance = m.get('sigma_data', 1.), m.get('has_variance', False)
    loss_config = m.get('loss_config', 'karras')
    if loss_config == 'karras':
        if has_variance:
            return partial(layers.DenoiserWithVariance, sigma_data=sigma_data, weighting=m.get('loss_weighting', 'karras'))
        return partial(layers.Denoiser, sigma_data=sigma_data, weighting=m.get('loss_weighting', 'karras'), scales=m.get('loss_scales', 1))
    if loss_config == 'simple':
        if has_variance:
            raise ValueError('Simple loss config does not support a variance output')
        return partial(layers.SimpleLossDenoiser, sigma_data=sigma_data)
    raise ValueError('Unknown loss config type')
