"""lvdm.modules.attention (reference: lvdm/modules/attention.py:42-442)."""
from lvdm_amd.unet import (BasicTransformerBlock, CrossAttention, FeedForward, GEGLU, SpatialTransformer,  # noqa: F401
                           TemporalTransformer)
