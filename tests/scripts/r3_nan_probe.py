import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd.model import VIEWCRAFTER_UNET
from lvdm_amd.unet import UNetModel
DEV = "cuda:0"
torch.manual_seed(0)
with torch.device(DEV):
    unet = UNetModel(**VIEWCRAFTER_UNET)
g = torch.Generator(device=DEV).manual_seed(0)
with torch.no_grad():
    for p_ in unet.parameters():
        if float(p_.abs().max()) == 0.0:
            p_.copy_(torch.randn(p_.shape, device=DEV, generator=g) * 0.02)
unet.eval().requires_grad_(False).half().to_token_major()
ctx = torch.randn(1, 333, 1024, device=DEV, generator=g).half()
t, fs = torch.tensor([500], device=DEV), torch.tensor([10], device=DEV)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = torch.randn(1, 8, T, 72, 128, device=DEV, generator=g).half()
bad = []
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(o) and not torch.isfinite(o).all() and len(bad) < 6:
            xi = inp[0] if len(inp) and torch.is_tensor(inp[0]) else None
            bad.append((name, type(mod).__name__, tuple(o.shape), None if xi is None else (tuple(xi.shape), bool(torch.isfinite(xi).all()), float(xi.float().abs().max()))))
    return f
for n, m in unet.named_modules():
    if n:
        m.register_forward_hook(hook(n))
with torch.no_grad():
    y = unet(x, t, context=ctx, fs=fs)
print("finite:", bool(torch.isfinite(y).all()), "std", float(y.float().std()))
for b in bad:
    print(b)
