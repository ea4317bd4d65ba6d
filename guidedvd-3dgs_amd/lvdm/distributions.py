"""lvdm/distributions.py (reference :24-41): the posterior the VAE encoder returns."""
from lvdm_amd.vae import DiagonalGaussianDistribution  # noqa: F401
