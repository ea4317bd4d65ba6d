"""A few launches of the L0 self-attention (25 frames x 5 heads, N = 9216) forward [and backward] for PMC runs."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch
from lvdm_amd import ops
g = torch.Generator(device="cuda:0").manual_seed(0)
q, k, v = (torch.randn(25, 9216, 320, device="cuda:0", generator=g).half() for _ in range(3))
for _ in range(4):
    o, lse = ops._hip_attention_fwd(q, k, v, 5, False, want_lse=True)
if len(sys.argv) > 1:
    go = torch.randn_like(o)
    for _ in range(2):
        ops._hip_attention_bwd(q, k, v, o, go, lse, 5, False)
torch.cuda.synchronize()
