"""lvdm.modules.networks.ae_modules (reference: lvdm/modules/networks/ae_modules.py:26-578)."""
from lvdm_amd.vae import AttnBlock, Decoder, Downsample, Encoder, ResnetBlock, Upsample  # noqa: F401
