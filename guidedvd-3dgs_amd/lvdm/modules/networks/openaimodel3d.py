"""lvdm.modules.networks.openaimodel3d (reference: lvdm/modules/networks/openaimodel3d.py:30-603)."""
from lvdm_amd.unet import (Downsample, ResBlock, TemporalConvBlock, TimestepEmbedSequential, UNetModel,  # noqa: F401
                           Upsample)
