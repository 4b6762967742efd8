"""Config loading with the reference's YAML schema (bem/utils_exp.py:256-259) and the model factory
(dlpm/dlpm_experiment.py:68-84)."""
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
IMAGE_DATASETS = ('mnist', 'cifar10', 'cifar10_lt', 'celeba', 'celebahq', 'lsun', 'tinyimagenet', 'fashion_mnist')


def is_image_dataset(name):
    return name.lower() in IMAGE_DATASETS


def load_config(name_or_path):
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(_HERE, 'configs', name_or_path if name_or_path.endswith('.yml') else name_or_path + '.yml')
    with open(path) as f:
        return yaml.safe_load(f)


def init_model_by_parameter(p):
    from .mlp import MLPModel
    from .unet import unet_from_config
    if not is_image_dataset(p['data']['dataset']):
        return MLPModel(p)
    if p['model']['model_type'] != 'ddpm':
        raise ValueError('model type {} not recognized (only the improved-DDPM UNet is on the DLPM path)'.format(
            p['model']['model_type']))
    return unet_from_config(p)


def sample_shape(p):
    """Per-sample shape, as the reference's datasets produce it (bem/datasets/__init__.py:141-149)."""
    if is_image_dataset(p['data']['dataset']):
        return [p['data']['channels'], p['data']['image_size'], p['data']['image_size']]
    return [1, p['data']['nfeatures']]
