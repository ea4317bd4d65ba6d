"""lvdm.models.samplers.ddim_multiplecond: imported unconditionally by utils_vc/diffusion_utils.py:10 and selected when
`multiple_cond_cfg` is set (:123-125; never by the guidedvd drivers, configs/infer_config.py).  The three-way text x image classifier-free
guidance is implemented in lvdm_amd.samplers.DDIMSamplerMultiCond (round 6; a stub that refused `cfg_img` until then)."""
from lvdm_amd.samplers import DDIMSamplerMultiCond as DDIMSampler  # noqa: F401
