"""lvdm.modules.encoders.condition (reference: lvdm/modules/encoders/condition.py:174-372): the frozen OpenCLIP ViT-H/14
text and image towers that produce the once-per-video conditioning."""
from lvdm_amd.clip import FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedderV2  # noqa: F401
