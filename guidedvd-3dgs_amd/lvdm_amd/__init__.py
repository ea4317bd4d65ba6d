"""MI355X-native rebuild of the ViewCrafter DDIM denoise loop.

    schedule   B4   beta / zero-terminal-SNR / DDIM tables, v-parameterisation helpers, timestep embedding
    samplers   B2/B3 DDIMSampler, DDIMSamplerGuidance (same .sample() API as lvdm.models.samplers.*)
    unet       B6-B11 UNetModel (state-dict compatible with the ViewCrafter checkpoint)
    vae        B13  KL-VAE decoder / encoder
    conv       the hand-written MFMA convolutions (3x3, upsample, temporal) behind unet / vae
    guidance   B12  LossGuidance
    model      B1/B5 LatentDiffusion-shaped wrapper the samplers duck-type against (apply_model, decode)
    ops        hot operators; HIP kernels (csrc/*.hip) on ROCm devices, no silent CPU fallback

Importing this package changes no process-wide state.  `configure_tuning()` is the explicit opt-in for the recorded
library-solution choices (hipBLASLt via PyTorch TunableOp for the Linear layers; MIOpen find results for the little that still reaches it).
"""
import os as _os
import shutil as _shutil
import tempfile as _tempfile

_HERE = _os.path.dirname(_os.path.abspath(__file__))
_CONFIGURED = False


def _atomic_copy(src, dst):
    """Copy through a temporary file + os.replace: concurrent ranks never see a half-written file."""
    fd, tmp = _tempfile.mkstemp(dir=_os.path.dirname(dst), prefix=".tmp_")
    _os.close(fd)
    try:
        _shutil.copyfile(src, tmp)
        _os.replace(tmp, dst)
    finally:
        if _os.path.exists(tmp):
            _os.unlink(tmp)


def configure_tuning(tunableop=True, miopen_db=True, cache_dir=None):
    """Opt in (once per process, before the first GEMM / convolution of the model) to the library-solution choices recorded
    on an MI355X and shipped read-only in `lvdm_amd/tunableop/` and `lvdm_amd/miopen_db/`:

      * hipBLASLt / rocBLAS solutions for the U-Net's Linear shapes through PyTorch TunableOp (reading only: tuning stays off
        unless the caller turned it on; a file written by another PyTorch / hipBLASLt build fails TunableOp's validators and
        is ignored);
      * MIOpen find-db entries and `PYTORCH_MIOPEN_SUGGEST_NHWC=1` (token-major tensors reach MIOpen's NHWC kernels
        un-transposed) for what still reaches MIOpen: on the 16-bit product path only the CLIP patch embedding, once per
        video -- every 3x3 / stride-2 / upsampling / temporal convolution of the U-Net and the VAE runs the package's own MFMA
        kernel -- plus the fp32 torch-form branches that `ops` warns about.

    Writable copies live in a per-user, per-rank cache directory (default `$XDG_CACHE_HOME/guidedvd_amd/rank<LOCAL_RANK>`),
    never in the source tree; variables the caller already exported are left alone.  Returns the cache directory."""
    global _CONFIGURED
    rank = _os.environ.get("LOCAL_RANK", "0")
    base = cache_dir or _os.path.join(_os.environ.get("XDG_CACHE_HOME") or _os.path.join(_os.path.expanduser("~"), ".cache"),
                                      "guidedvd_amd", f"rank{rank}")
    if _CONFIGURED:
        return base
    try:
        _os.makedirs(base, exist_ok=True)
    except OSError:
        return None
    _os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
    if miopen_db and "MIOPEN_USER_DB_PATH" not in _os.environ:
        src, dst = _os.path.join(_HERE, "miopen_db"), _os.path.join(base, "miopen_db")
        try:
            _os.makedirs(dst, exist_ok=True)
            for f in _os.listdir(src) if _os.path.isdir(src) else []:
                if not _os.path.exists(_os.path.join(dst, f)):
                    _atomic_copy(_os.path.join(src, f), _os.path.join(dst, f))
            _os.environ["MIOPEN_USER_DB_PATH"] = dst
        except OSError:
            pass
    rec = _os.path.join(_HERE, "tunableop", "tunableop0.csv")
    if tunableop and _os.path.exists(rec) and "PYTORCH_TUNABLEOP_ENABLED" not in _os.environ:
        try:
            dst = _os.path.join(base, "tunableop")
            _os.makedirs(dst, exist_ok=True)
            for i in range(8):   # TunableOp looks for <name><device ordinal>.csv
                f = _os.path.join(dst, f"tunableop{i}.csv")
                if not _os.path.exists(f) or _os.path.getmtime(f) < _os.path.getmtime(rec):
                    _atomic_copy(rec, f)
            _os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
            _os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
            _os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", _os.path.join(dst, "tunableop.csv"))
        except OSError:
            pass
    _CONFIGURED = True
    return base
