"""lvdm/basics.py names (reference: third_party/ViewCrafter/lvdm/basics.py:14-86)."""
import torch.nn as nn

from lvdm_amd.model import instantiate_from_config  # noqa: F401  (imported from here by reference-side code)
from lvdm_amd.unet import GroupNorm32 as GroupNormSpecific, zero_module  # noqa: F401


def disabled_train(self, mode=True):
    """Overwrite model.train with this function to make sure train/eval mode does not change anymore."""
    return self


def scale_module(module, scale):
    for p in module.parameters():
        p.detach().mul_(scale)
    return module


def conv_nd(dims, *args, **kwargs):
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dims](*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def avg_pool_nd(dims, *args, **kwargs):
    return {1: nn.AvgPool1d, 2: nn.AvgPool2d, 3: nn.AvgPool3d}[dims](*args, **kwargs)


def nonlinearity(type='silu'):
    return {"silu": nn.SiLU, "leaky_relu": nn.LeakyReLU}[type]()


def normalization(channels, num_groups=32):
    return GroupNormSpecific(num_groups, channels)
