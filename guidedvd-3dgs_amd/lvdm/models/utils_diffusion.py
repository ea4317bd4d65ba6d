"""lvdm/models/utils_diffusion.py (reference :8-158): schedule helpers, bit-exact tables (tests/test_diffusion_cpu.py)."""
from lvdm_amd.schedule import (make_beta_schedule, make_ddim_sampling_parameters, make_ddim_timesteps,  # noqa: F401
                               rescale_noise_cfg, rescale_zero_terminal_snr, timestep_embedding)
