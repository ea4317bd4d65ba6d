"""GradCell (lvdm_amd/ops.py): the hand-over of a residual-branch gradient to the normalisation node that shares its input, out of
autograd (~170 forks per differentiable U-Net evaluation).  Advisor finding, round 4: nothing named it in a test.  Here:
  * the put / take protocol as a unit, including the taker running first and a second pass over a retained graph (CPU);
  * on the device, one differentiable U-Net evaluation and one VAE decode with the cells ON against OFF (every fan-in sum left to
    autograd): d/dx must agree to fp16 rounding; a second backward over a retained graph; activation checkpointing on."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from fill_by_name import fill_by_name
from lvdm_amd import ops


def test_put_take_protocol(monkeypatch):
    monkeypatch.setattr(ops, "GRAD_CELLS", True)
    g1, g2 = torch.ones(3), 2 * torch.ones(3)
    c = ops.GradCell()
    assert c.put(g1) is False and c.g is None            # unarmed: the putter keeps its gradient (autograd sums as usual)
    c = ops.GradCell().arm()
    assert c.put(g1) is True and c.put(g2) is True       # two putters: summed in the cell
    assert torch.equal(c.take(), 3 * torch.ones(3)) and c.g is None
    assert c.put(g1) is False                            # second backward over a retained graph: already taken -> back to autograd
    assert c.take() is None
    c = ops.GradCell().arm()                             # the taker runs FIRST (the engine ordered the norm node before the residual node)
    assert c.take() is None
    assert c.put(g1) is False and c.g is None            # ... then the putter must not strand its gradient in the cell
    c = ops.GradCell().arm()                             # reshape to the taker's layout
    c.put(torch.arange(6.0).reshape(2, 3))
    assert c.take(like=torch.empty(3, 2)).shape == (3, 2)
    monkeypatch.setattr(ops, "GRAD_CELLS", False)
    c = ops.GradCell().arm()
    assert c.armed is False and c.put(g1) is False       # GVD_GRAD_CELLS=0: every cell stays unarmed


DEV = "cuda:0"
_UNET = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[1, 2], num_res_blocks=1,
             channel_mult=[1, 2], dropout=0.0, num_head_channels=64, transformer_depth=1, context_dim=64, use_linear=True,
             use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
             use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
             image_cross_attention=True, default_fs=10, fs_condition=True)


def _unet_grad(monkeypatch, cells, use_checkpoint=False, twice=False):
    from lvdm_amd.model import DiffusionWrapper
    from lvdm_amd.unet import UNetModel
    monkeypatch.setattr(ops, "GRAD_CELLS", cells)
    unet = fill_by_name(UNetModel(**dict(_UNET, use_checkpoint=use_checkpoint)), std=0.08).half().eval().to(DEV).to_token_major()
    unet.requires_grad_(False)
    w = DiffusionWrapper(unet)
    g = torch.Generator(device=DEV).manual_seed(3)
    mk = lambda *s: torch.randn(*s, device=DEV, generator=g)
    T, H, W = 5, 16, 24
    x = mk(1, 4, T, H, W).requires_grad_(True)
    c = {"c_crossattn": [mk(1, 93, 64).half()], "c_concat": [(mk(1, 4, T, H, W) * 0.2).half()]}
    t, fs, probe = torch.tensor([500], device=DEV), torch.tensor([10], device=DEV), mk(1, 4, T, H, W)
    e = w(x.half(), t, **c, fs=fs)
    loss = (e.float() * probe).sum()
    (g1,) = torch.autograd.grad(loss, x, retain_graph=twice)
    g2 = torch.autograd.grad(loss, x)[0] if twice else None
    return e.detach().float(), g1.float(), None if g2 is None else g2.float()


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.gpu
def test_unet_input_gradient_with_cells_matches_autograd_fan_in(monkeypatch):
    e0, g0, _ = _unet_grad(monkeypatch, False)
    e1, g1, _ = _unet_grad(monkeypatch, True)
    assert torch.equal(e0, e1)                               # the forward does not depend on the cells
    assert _rel(g1, g0) < 4e-3, _rel(g1, g0)                 # fp32 sum, one rounding, against autograd's fp16 adds


@pytest.mark.gpu
def test_second_backward_over_a_retained_graph_loses_no_gradient(monkeypatch):
    """The cells are consumed by the first backward; the second one must fall back to autograd's own fan-in sums."""
    _, g0, _ = _unet_grad(monkeypatch, False)
    _, g1, g2 = _unet_grad(monkeypatch, True, twice=True)
    assert _rel(g1, g0) < 4e-3 and _rel(g2, g0) < 4e-3, (_rel(g1, g0), _rel(g2, g0))


@pytest.mark.gpu
def test_activation_checkpointing_with_cells(monkeypatch):
    """use_checkpoint=True re-runs the block forwards inside the backward: cells armed in the recomputation, not in the first pass."""
    _, g0, _ = _unet_grad(monkeypatch, False)
    _, g1, _ = _unet_grad(monkeypatch, True, use_checkpoint=True)
    assert _rel(g1, g0) < 4e-3, _rel(g1, g0)


@pytest.mark.gpu
def test_vae_decode_input_gradient_with_cells_matches_autograd_fan_in(monkeypatch):
    from lvdm_amd.vae import AutoencoderKLDecoder
    cfg = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4],
               num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    vae = fill_by_name(AutoencoderKLDecoder(cfg), std=0.05).half().eval().to(DEV).requires_grad_(False).to_token_major()
    z = torch.randn(1, 4, 12, 20, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)).half()
    out = []
    for cells in (False, True):
        monkeypatch.setattr(ops, "GRAD_CELLS", cells)
        zz = z.clone().requires_grad_(True)
        y = vae.decode(zz)
        (gz,) = torch.autograd.grad((y.float() ** 2).sum(), zz)
        out.append((y.float(), gz.float()))
    assert torch.equal(out[0][0], out[1][0])
    assert _rel(out[1][1], out[0][1]) < 4e-3, _rel(out[1][1], out[0][1])
