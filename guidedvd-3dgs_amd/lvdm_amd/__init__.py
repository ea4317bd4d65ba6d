"""MI355X-native rebuild of the ViewCrafter DDIM denoise loop.

    schedule   B4   beta / zero-terminal-SNR / DDIM tables, v-parameterisation helpers, timestep embedding
    samplers   B2/B3 DDIMSampler, DDIMSamplerGuidance (same .sample() API as lvdm.models.samplers.*)
    unet       B6-B11 UNetModel (state-dict compatible with the ViewCrafter checkpoint)
    vae        B13  KL-VAE decoder (+ post_quant_conv) used inside the guided step
    guidance   B12  LossGuidance
    model      B1/B5 LatentDiffusion-shaped wrapper the samplers duck-type against (apply_model, decode)
    ops        hot operators; HIP kernels (csrc/diffusion_*.hip) on ROCm devices, no silent CPU fallback
"""

import os as _os

# The U-Net keeps every feature map token-major (channels_last); PyTorch-ROCm only hands NHWC tensors to MIOpen's
# NHWC kernels when this is set (otherwise it transposes to NCHW and back around every convolution).
_os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")

# MIOpen's default (fast-find) heuristic picks a slow generic kernel for several of the U-Net's NHWC shapes; the
# measured per-shape choices for gfx950 / 256 CUs (recorded by an exhaustive find on an MI355X, `GVD_CONV_FIND=1`) ship
# with the package so that a fresh process starts with them instead of re-running a multi-minute find.
_db = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")
if _os.path.isdir(_db) and _os.access(_db, _os.W_OK):
    _os.environ.setdefault("MIOPEN_USER_DB_PATH", _db)

# hipBLASLt / rocBLAS solution choices for the U-Net's GEMM shapes, recorded by PyTorch TunableOp on an MI355X
# (tests/scripts/run_ddim_tunable.sh).  Read-only use: tuning stays off unless the caller turns it on; a file written
# by another PyTorch / hipBLASLt build fails TunableOp's validators and is ignored.  TunableOp looks for
# <name><device ordinal>.csv, so the one recorded file is mirrored for the 8 GPUs of a node.
_tun = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tunableop")
if _os.path.exists(_os.path.join(_tun, "tunableop0.csv")) and "PYTORCH_TUNABLEOP_ENABLED" not in _os.environ:
    try:
        import shutil as _sh
        for _i in range(1, 8):
            _dst = _os.path.join(_tun, f"tunableop{_i}.csv")
            if not _os.path.exists(_dst) or _os.path.getmtime(_dst) < _os.path.getmtime(_os.path.join(_tun, "tunableop0.csv")):
                _sh.copyfile(_os.path.join(_tun, "tunableop0.csv"), _os.path.join(_tun, f"tunableop{_i}.csv"))
    except OSError:
        pass
    _os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    _os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
    _os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", _os.path.join(_tun, "tunableop.csv"))
