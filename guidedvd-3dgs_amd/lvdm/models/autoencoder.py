"""lvdm.models.autoencoder.AutoencoderKL (reference: lvdm/models/autoencoder.py:13-107): constructor keywords of the
yaml's `first_stage_config` (configs/inference_pvd_1024.yaml:66-87), `encode` / `decode`, same state-dict keys
(`encoder.*`, `decoder.*`, `quant_conv.*`, `post_quant_conv.*`).  No pytorch-lightning."""
from lvdm_amd.vae import AutoencoderKLDecoder, DiagonalGaussianDistribution  # noqa: F401


class AutoencoderKL(AutoencoderKLDecoder):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None):
        if not ddconfig["double_z"]:
            raise AssertionError("AutoencoderKL needs double_z")
        super().__init__(dict(ddconfig), embed_dim=embed_dim, with_encoder=True)
        self.image_key, self.embed_dim, self.input_dim = image_key, embed_dim, input_dim
        if monitor is not None:
            self.monitor = monitor
        if ckpt_path is not None:
            import torch
            sd = torch.load(ckpt_path, map_location="cpu")
            sd = sd.get("state_dict", sd)
            self.load_state_dict({k: v for k, v in sd.items() if not any(k.startswith(i) for i in ignore_keys)}, strict=False)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior

    def get_last_layer(self):
        return self.decoder.conv_out.weight
