"""lvdm.models.samplers.ddim (reference: lvdm/models/samplers/ddim.py:10-280)."""
from lvdm_amd.samplers import DDIMSampler  # noqa: F401
