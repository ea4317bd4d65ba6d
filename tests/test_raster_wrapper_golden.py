"""The PYTHON half of the rasterizer against the reference's own wrapper: tests/golden/make_golden_raster_wrapper.py drove
submodules/diff-gaussian-rasterization-confidence/diff_gaussian_rasterization/__init__.py with the deterministic native stand-in of
tests/raster_fake_backend.py; here this repository's diff_gaussian_rasterization (its Python autograd operator, the carrier used for debug dumps, the
capacity mode and whenever the compiled operator is absent) is driven with the same stand-in.  Equal: the 13 settings fields, every argument the native
entry points receive, slot by slot, the outputs, the gradients that reach the inputs -- with the confidence applied by the native side here
(kernel contract) and in Python there -- the markVisible call and the five exception messages.  CPU only; nothing native runs."""
import json
import os

import numpy as np
import pytest
import torch

import raster_fake_backend as fb

HERE = os.path.dirname(os.path.abspath(__file__))
R = np.load(os.path.join(HERE, "golden", "raster_wrapper_ref.npz"), allow_pickle=False)


@pytest.fixture()
def wrapped(monkeypatch):
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C
    backend = fb.FakeBackend()
    monkeypatch.setattr(_C, "rasterize_gaussians", backend.rasterize_gaussians)
    monkeypatch.setattr(_C, "rasterize_gaussians_backward", backend.rasterize_gaussians_backward)
    monkeypatch.setattr(_C, "mark_visible", backend.mark_visible)
    monkeypatch.setattr(_C, "ext", lambda: None)            # the Python carrier (the compiled operator has its own GPU test against it)
    monkeypatch.setattr(_C, "_CAPACITY", 0)
    return dgr, backend


def _same_calls(ours, ref_json):
    ref = json.loads(str(ref_json))
    assert len(ours) == len(ref)
    for (kind_o, args_o), (kind_r, args_r) in zip(ours, ref):
        assert kind_o == kind_r and len(args_o) == len(args_r), (kind_o, len(args_o), len(args_r))
        for i, (a, b) in enumerate(zip(args_o, args_r)):
            a = json.loads(json.dumps(a))                   # tuples -> lists, like the stored form
            if a[0] == "T" and b[0] == "T":
                assert a[1] == b[1] and a[2] == b[2] and abs(a[3] - b[3]) <= 1e-5 * max(1.0, abs(b[3])), (kind_o, i, a, b)
            else:
                assert a == b, (kind_o, i, a, b)


def test_settings_tuple_is_the_references():
    import diff_gaussian_rasterization as dgr
    assert list(dgr.GaussianRasterizationSettings._fields) == list(R["settings_fields"])


@pytest.mark.parametrize("case", ["sh", "pre"])
def test_operator_slots_outputs_and_gradients_match_the_references_wrapper(wrapped, case):
    dgr, backend = wrapped
    sc = fb.scene()
    outs, grads = fb.drive(dgr, backend, sc, case)
    _same_calls(backend.calls, R[f"{case}_calls"])
    for k, v in outs.items():
        assert np.array_equal(v.numpy(), R[f"{case}_out_{k}"]), k
    assert {f"{case}_grad_{k}" for k in grads} == {k for k in R.files if k.startswith(f"{case}_grad_")}
    for k, v in grads.items():
        np.testing.assert_allclose(v.numpy(), R[f"{case}_grad_{k}"], rtol=1e-6, atol=1e-7, err_msg=k)
    # the screen-space gradient is the one the confidence does not touch -- in both packages
    assert float(np.abs(R[f"{case}_grad_means2D"]).max()) > 0


def test_mark_visible_and_the_exception_messages(wrapped):
    dgr, backend = wrapped
    sc = fb.scene()
    S = dgr.GaussianRasterizationSettings(6, 8, 0.7, 0.6, sc["bg"], 1.25, sc["view"], sc["proj"], 3, sc["campos"], False, False, sc["confidence"])
    vis = dgr.GaussianRasterizer(S).markVisible(sc["means3D"])
    assert np.array_equal(vis.numpy(), R["vis"])
    _same_calls(backend.calls, R["vis_calls"])
    msgs = []
    for kw in (dict(), dict(shs=sc["shs"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"]), dict(shs=sc["shs"]),
               dict(shs=sc["shs"], scales=sc["scales"]), dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=sc["cov3D"])):
        with pytest.raises(Exception) as e:
            dgr.GaussianRasterizer(S)(means3D=sc["means3D"], means2D=sc["means2D"], opacities=sc["opacities"], **kw)
        msgs.append(f"{type(e.value).__name__}: {e.value}")
    assert msgs == list(R["error_messages"])
