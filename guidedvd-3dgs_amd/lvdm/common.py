"""lvdm/common.py names used on the sampling path (reference: third_party/ViewCrafter/lvdm/common.py:25-52)."""
import math

import torch


def extract_into_tensor(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def noise_like(shape, device, repeat=False):
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) and not isinstance(d, torch.nn.Module) else d


def exists(val):
    return val is not None


def identity(*args, **kwargs):
    return torch.nn.Identity()


def uniq(arr):
    return {el: True for el in arr}.keys()


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def max_neg_value(t):
    return -torch.finfo(t.dtype).max


def init_(tensor):
    std = 1 / math.sqrt(tensor.shape[-1])
    tensor.uniform_(-std, std)
    return tensor


def checkpoint(func, inputs, params, flag):
    """Activation checkpointing hook of the reference (common.py:81-94); `params` is unused by torch's implementation."""
    if flag:
        from torch.utils.checkpoint import checkpoint as _ckpt
        return _ckpt(func, *inputs, use_reentrant=False)
    return func(*inputs)
