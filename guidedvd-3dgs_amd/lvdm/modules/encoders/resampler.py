"""lvdm.modules.encoders.resampler (reference: lvdm/modules/encoders/resampler.py:9-144)."""
from lvdm_amd.resampler import ImageProjModel, Resampler  # noqa: F401
