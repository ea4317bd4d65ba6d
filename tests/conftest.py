import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "guidedvd-3dgs_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# The parity tests feed fp32 device tensors through the package ON PURPOSE (fp32 legs of the goldens, the Resampler / CLIP towers in their
# checkpoint dtype); the product default raises there (lvdm_amd.allow_torch_fallback).  The suite opts in for itself and every process it
# spawns; tests/test_diffusion_gpu.py::test_fp32_device_tensors_raise_unless_the_caller_opts_in holds the default.
os.environ.setdefault("GVD_TORCH_FALLBACK", "warn")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import raster_oracle
    raster_oracle.lib()
    return raster_oracle
