"""Which autograd nodes does one differentiable U-Net evaluation of the guided step build?  (dev tool)  Counts grad_fn classes of the
batch-2 CFG-pair forward at 320x448 and prints, for the slice / cat / copy / add style nodes, the tensor sizes they will materialise
in the backward (a SliceBackward is a zero-fill + copy of the full source, an AddBackward on a fork an accumulation add)."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd.model import VIEWCRAFTER_UNET, DiffusionWrapper
from lvdm_amd.unet import UNetModel
dev = torch.device("cuda:0")
T, h, w = 25, 40, 56
g = torch.Generator(device=dev).manual_seed(0)
with torch.device(dev):
    unet = UNetModel(**VIEWCRAFTER_UNET)
with torch.no_grad():
    for p in unet.parameters():
        if float(p.abs().max()) == 0.0:
            p.copy_(torch.randn(p.shape, device=dev, generator=g) * 0.02)
unet = unet.half().eval().to_token_major().requires_grad_(False)
m = DiffusionWrapper(unet)
x = torch.randn(2, 4, T, h, w, device=dev, generator=g).requires_grad_(True)
c = {"c_crossattn": [torch.randn(2, 333, 1024, device=dev, generator=g).half()], "c_concat": [(torch.randn(2, 4, T, h, w, device=dev, generator=g) * 0.18).half()]}
e = m(x.half(), torch.tensor([500, 500], device=dev), **c, fs=torch.tensor([10, 10], device=dev))
seen, stack, cnt, big = set(), [e.grad_fn], collections.Counter(), collections.Counter()
while stack:
    fn = stack.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    name = type(fn).__name__
    cnt[name] += 1
    for nxt, _ in fn.next_functions:
        stack.append(nxt)
    for attr in ("_saved_self_sym_sizes", "_saved_self_sizes", "_saved_sizes"):
        if hasattr(fn, attr):
            try:
                sz = tuple(int(v) for v in getattr(fn, attr))
                n = 1
                for v in sz:
                    n *= v
                if n >= 1 << 20:
                    big[(name, sz)] += 1
            except Exception:
                pass
            break
print("autograd nodes of one batch-2 U-Net forward:", sum(cnt.values()))
for k, v in cnt.most_common(40):
    print(f"{v:6d}  {k}")
print("-- nodes with a saved source size >= 1 M elements")
for (name, sz), v in sorted(big.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{v:5d}  {name:28s} {sz}")
