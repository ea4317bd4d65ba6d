"""Image losses around the rasterizer.  Goldens come from the reference's own utils/loss_utils.py
(tests/golden/make_golden_ssim.py).  Tolerances: fp32 sums of 121 products in a different order than the reference's
convolution -> 2e-6 on the value (mean of O(1) numbers), 1e-4 relative to the largest entry on the gradient."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

G = np.load(os.path.join(ROOT, "tests", "golden", "ssim_ref.npz"))
CASES = ["a", "b", "c", "d"]


@pytest.mark.parametrize("tag", CASES)
def test_ssim_oracle_reproduces_the_reference_goldens(tag):
    from oracle import ssim_oracle as so
    v, g = so.ssim(G[f"{tag}_x"], G[f"{tag}_y"])
    assert abs(float(v) - float(G[f"{tag}_ssim"])) < 2e-6
    ref = G[f"{tag}_grad"]
    assert np.abs(g - ref).max() < 1e-4 * np.abs(ref).max()
    if f"{tag}_ssim_per_batch" in G:
        vb, _ = so.ssim(G[f"{tag}_x"], G[f"{tag}_y"], size_average=False)
        np.testing.assert_allclose(vb, G[f"{tag}_ssim_per_batch"], atol=2e-6)
    if f"{tag}_mask" in G:
        m = G[f"{tag}_mask"]
        vm, gm = so.ssim(G[f"{tag}_x"] * m + (1 - m), G[f"{tag}_y"] * m + (1 - m))
        assert abs(float(vm) - float(G[f"{tag}_ssim_masked"])) < 2e-6
        refm = G[f"{tag}_grad_masked"]
        assert np.abs(gm * m - refm).max() < 1e-4 * np.abs(refm).max()


def test_loss_library_exports_and_host_helpers():
    import __graft_entry__ as g
    g.build_loss()
    import fused_loss as fl
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gvd_loss.h")).read(), flags=re.S)
    names = set(re.findall(r"\b(gvd_[a-z_0-9]+)\s*\(", hdr))
    assert {"gvd_ssim_forward", "gvd_ssim_backward", "gvd_ssim_partial_count", "gvd_loss_last_error",
            "gvd_photometric_forward", "gvd_photometric_backward"} <= names
    L = fl.lib()
    for n in sorted(names):
        assert hasattr(L, n), f"libgvd_loss.so does not export {n}"
    assert L.gvd_ssim_partial_count(3, 480, 640) == 3 * 30 * 40 and L.gvd_ssim_partial_count(1, 17, 16) == 2
    from oracle import ssim_oracle as so
    w1 = fl.gaussian(11, 1.5).numpy()
    np.testing.assert_allclose(np.outer(w1, w1), so.window(), rtol=1e-6)   # fp32 normalisation sums in a different order
    x, y = torch.tensor(G["a_x"]), torch.tensor(G["a_y"])
    assert abs(float(fl.l1_loss(x, y)) - float(G["a_l1"])) < 1e-7
    assert abs(float(fl.l1_loss_mask(x, y, torch.tensor(G["a_mask"]))) - float(G["a_l1_masked"])) < 1e-7
    with pytest.raises(RuntimeError, match="ROCm device"):
        fl.ssim(x, y)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_fused_ssim_matches_the_reference_goldens_on_device(tag):
    import fused_loss as fl
    x = torch.tensor(G[f"{tag}_x"], device="cuda:0").requires_grad_(True)
    y = torch.tensor(G[f"{tag}_y"], device="cuda:0")
    v = fl.ssim(x, y)
    (g,) = torch.autograd.grad(v, x)
    assert abs(float(v) - float(G[f"{tag}_ssim"])) < 2e-6
    ref = G[f"{tag}_grad"]
    assert np.abs(g.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
    if f"{tag}_ssim_per_batch" in G:
        vb = fl.ssim(x.detach(), y, size_average=False)
        np.testing.assert_allclose(vb.cpu().numpy(), G[f"{tag}_ssim_per_batch"], atol=2e-6)
    if f"{tag}_mask" in G:
        m = torch.tensor(G[f"{tag}_mask"], device="cuda:0")
        vm = fl.ssim(x, y, m)
        (gm,) = torch.autograd.grad(vm, x)
        assert abs(float(vm) - float(G[f"{tag}_ssim_masked"])) < 2e-6
        refm = G[f"{tag}_grad_masked"]
        assert np.abs(gm.cpu().numpy() - refm).max() < 1e-4 * np.abs(refm).max()


@pytest.mark.gpu
def test_fused_ssim_full_size_against_the_oracle_and_properties():
    """640x480 (the C2 render size): value and gradient against the float64 oracle; ssim(x, x) == 1 with zero gradient;
    symmetric in its arguments; reproducible bit for bit run to run."""
    import fused_loss as fl
    from oracle import ssim_oracle as so
    g = torch.Generator(device="cuda:0").manual_seed(4)
    x = torch.rand(3, 480, 640, device="cuda:0", generator=g).requires_grad_(True)
    y = (x.detach() + 0.1 * torch.randn(3, 480, 640, device="cuda:0", generator=g)).clamp(0, 1)
    v = fl.ssim(x, y)
    (gx,) = torch.autograd.grad(v, x)
    vo, go = so.ssim(x.detach().cpu().numpy(), y.cpu().numpy())
    assert abs(float(v) - float(vo)) < 2e-6
    assert np.abs(gx.cpu().numpy() - go).max() < 1e-4 * np.abs(go).max()
    v2 = fl.ssim(x, y)
    assert torch.equal(v, v2) and torch.equal(gx, torch.autograd.grad(v2, x)[0])
    assert abs(float(fl.ssim(y, x.detach())) - float(v)) < 1e-6
    xx = y.clone().requires_grad_(True)
    one = fl.ssim(xx, y)
    assert abs(float(one) - 1.0) < 1e-6 and float(torch.autograd.grad(one, xx)[0].abs().max()) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "d"])
def test_fused_photometric_loss_equals_its_two_term_form_and_the_goldens(tag):
    """loss = 0.8 L1 + 0.2 (1 - ssim) (train_guidedvd.py:339-340) as one op: value from the reference goldens of both
    terms; gradient = 0.8 sign(x - y)/N - 0.2 * golden ssim gradient (tolerance as for ssim alone)."""
    import fused_loss as fl
    x = torch.tensor(G[f"{tag}_x"], device="cuda:0").requires_grad_(True)
    y = torch.tensor(G[f"{tag}_y"], device="cuda:0")
    loss, stats = fl.photometric_loss(x, y, 0.2)
    (g,) = torch.autograd.grad(loss * 3.0, x)   # non-trivial upstream factor
    ref = 0.8 * float(G[f"{tag}_l1"]) + 0.2 * (1.0 - float(G[f"{tag}_ssim"]))
    assert abs(float(loss) - ref) < 3e-6
    assert abs(float(stats[1]) - float(G[f"{tag}_l1"])) < 2e-6 and abs(float(stats[2]) - float(G[f"{tag}_ssim"])) < 2e-6
    gref = 3.0 * (0.8 * np.sign(G[f"{tag}_x"] - G[f"{tag}_y"]) / G[f"{tag}_x"].size - 0.2 * G[f"{tag}_grad"])
    assert g.shape == x.shape and np.abs(g.cpu().numpy() - gref).max() < 1e-4 * np.abs(gref).max()
    assert not stats.requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(8)))
def test_fused_ssim_randomized_sizes_against_the_oracle(seed):
    """Sizes around the 16-pixel tile and the 5-pixel window radius (1x1 ... ragged), 1-4 planes, batch or not."""
    import fused_loss as fl
    from oracle import ssim_oracle as so
    rng = np.random.default_rng(900 + seed)
    H, W = int(rng.choice([1, 2, 5, 6, 11, 15, 16, 17, 31, 33, 50])), int(rng.choice([1, 3, 5, 10, 16, 17, 32, 47, 64, 65]))
    C = int(rng.integers(1, 5))
    shp = (C, H, W) if rng.integers(0, 2) else (int(rng.integers(1, 3)), C, H, W)
    x = rng.random(shp).astype(np.float32)
    y = np.clip(x + 0.2 * rng.normal(size=shp), 0, 1).astype(np.float32)
    tx = torch.tensor(x, device="cuda:0").requires_grad_(True)
    v = fl.ssim(tx, torch.tensor(y, device="cuda:0"))
    (g,) = torch.autograd.grad(v, tx)
    vo, go = so.ssim(x, y)
    assert abs(float(v) - float(vo)) < 3e-6, (shp, float(v), float(vo))
    assert np.abs(g.cpu().numpy() - go).max() < 1e-4 * max(np.abs(go).max(), 1e-12), shp


def _loss_guidance_case(device):
    from lvdm_amd.guidance import LossGuidance
    lg = LossGuidance(ddim_steps=50, recur_steps=1, ssim_guidance=True, device=str(device))
    lg.set_hw(24, 40)
    lg.set_guidance_images(torch.tensor(G["lg_G"], device=device))
    lg.set_guidance_masks(torch.tensor(G["lg_mask"], device=device))
    x = torch.tensor(G["lg_x"], device=device).requires_grad_(True)
    loss_dict, numel = lg(x, 10, 0, 1)
    (gx,) = torch.autograd.grad(loss_dict["recon"], x)
    assert abs(float(numel) - float(G["lg_numel"])) < 0.5
    assert abs(float(loss_dict["recon"]) - float(G["lg_loss"])) < 2e-5 * abs(float(G["lg_loss"]))
    ref = G["lg_grad"]
    assert np.abs(gx.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()


def test_loss_guidance_with_ssim_term_matches_the_reference_form_cpu():
    """LossGuidance(ssim_guidance=True) (viewcrafter_wrapper.py:145-155) against the value / gradient the reference's
    ssim_noavg gives (golden), explicit-math path."""
    from lvdm_amd import ops
    ops.use_reference_math(True)
    try:
        _loss_guidance_case(torch.device("cpu"))
    finally:
        ops.use_reference_math(False)


@pytest.mark.gpu
def test_loss_guidance_with_ssim_term_matches_the_reference_form_on_device():
    _loss_guidance_case(torch.device("cuda:0"))
