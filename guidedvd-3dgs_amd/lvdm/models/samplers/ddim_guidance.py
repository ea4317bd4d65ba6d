"""lvdm.models.samplers.ddim_guidance (reference: lvdm/models/samplers/ddim_guidance.py:12-363)."""
from lvdm_amd.samplers import DDIMSamplerGuidance  # noqa: F401
