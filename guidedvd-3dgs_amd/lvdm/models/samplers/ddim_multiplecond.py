"""lvdm.models.samplers.ddim_multiplecond: imported unconditionally by utils_vc/diffusion_utils.py:10, selected only when
`multiple_cond_cfg` is set (never by the guidedvd drivers, configs/infer_config.py).  The single-condition behaviour is the
plain sampler; the three-way CFG combination is refused instead of silently ignored."""
from lvdm_amd.samplers import DDIMSampler as _Plain


class DDIMSampler(_Plain):
    def sample(self, *args, **kwargs):
        if kwargs.get("unconditional_conditioning_img_nonetext") is not None or kwargs.get("cfg_img") not in (None, 1.0):
            raise NotImplementedError("multiple-condition CFG (cfg_img) is not part of the guidedvd hot path")
        return super().sample(*args, **kwargs)
