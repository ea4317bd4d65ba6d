"""How much of the blend kernels' lane work is useful?  For one C2 view, from the operator's own sorted lists and
preprocessed records: the number of (entry, pixel) pairs with alpha >= 1/255 (and before the pixel's last contributor),
and the number of (entry, block) steps a perfect cull would leave at block sizes 16x16 / 8x8 / 4x4 (x64, x16 lanes).
python tests/scripts/lane_stats.py [view]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synthetic as syn
from diff_gaussian_rasterization import _C

dev = "cuda:0"
P, W, H, D = 200000, 640, 480, 3
sc = syn.scene_c2(P=P, W=W, H=H, sh_degree=D)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
c = sc["cameras"][int(sys.argv[1]) if len(sys.argv) > 1 else 0]
R, color, depth, alpha, radii, geom, binb, img = _C.rasterize_gaussians(
    t(sc["bg"]), t(sc["means3D"]), torch.empty(0, device=dev), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0,
    torch.empty(0, device=dev), t(c["viewmatrix"]), t(c["projmatrix"]), c["tanfovx"], c["tanfovy"], H, W, t(sc["shs"]), D,
    t(c["campos"]), False, False)
torch.cuda.synchronize()
v = _C.chunk_views(P, W, H, R, geom, binb, img)
gx, gy = (W + 15) // 16, (H + 15) // 16
ranges = v["ranges"].cpu().numpy().astype(np.int64)
plist = v["point_list"].long()
xy, co, ncon = v["means2D"], v["conic_opacity"], v["n_contrib"].long()
tot_pairs = act_pairs = 0
steps = {16: 0, 8: 0, 4: 0, 2: 0}
rect84 = rect48 = bbox4 = bbox4q = 0
steps_8x8 = steps_pair84 = steps_pair48 = steps_quad44 = 0   # wave steps: one list per quadrant / max over its 2 halves / its 4 blocks
fwd_live = 0
comp_steps = {2: 0, 4: 0, 8: 0, 16: 0}   # round 6: wave steps if every pixel lane walked ITS OWN active entries of a window of k kept entries
chain_bound = 0                          # ... and of the whole list (the busiest pixel's chain: the floor of any lane-compacting scheme)
yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
for tile in range(gx * gy):
    r0, r1 = ranges[tile]
    n = int(r1 - r0)
    if n == 0:
        continue
    tx, ty = tile % gx, tile // gx
    ids = plist[r0:r1]
    px = (tx * 16 + xx).float().reshape(1, 256)
    py = (ty * 16 + yy).float().reshape(1, 256)
    inside = ((tx * 16 + xx) < W) & ((ty * 16 + yy) < H)
    dx = xy[ids, 0:1] - px
    dy = xy[ids, 1:2] - py
    cn = co[ids]
    power = -0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) - cn[:, 1:2] * dx * dy
    a = torch.clamp(cn[:, 3:4] * torch.exp(power), max=0.99)
    pyc = (ty * 16 + yy).clamp(max=H - 1).reshape(-1)
    pxc = (tx * 16 + xx).clamp(max=W - 1).reshape(-1)
    last = ncon[pyc, pxc].reshape(1, 256) * inside.reshape(1, 256)
    ordv = torch.arange(n, device=dev).reshape(n, 1)
    act = (power <= 0) & (a >= 1.0 / 255.0) & (ordv < last)
    tot_pairs += n * 256
    act_pairs += int(act.sum())
    A = act.reshape(n, 16, 16)
    for bs in steps:
        steps[bs] += int(A.reshape(n, 16 // bs, bs, 16 // bs, bs).any(dim=4).any(dim=2).sum())
    q88 = A.reshape(n, 2, 8, 2, 8).any(dim=4).any(dim=2).sum(dim=0)                      # [2, 2] list length per quadrant
    h84 = A.reshape(n, 2, 2, 4, 2, 8).any(dim=5).any(dim=3).sum(dim=0)                   # [qy, half, qx]: 8 wide x 4 high halves
    h48 = A.reshape(n, 2, 8, 2, 2, 4).any(dim=5).any(dim=2).sum(dim=0)                   # [qy, qx, half]: 4 wide x 8 high halves
    b44 = A.reshape(n, 2, 2, 4, 2, 2, 4).any(dim=6).any(dim=3).sum(dim=0)                # [qy, sy, qx, sx]
    steps_8x8 += int(q88.sum())
    Aq = A.reshape(n, 2, 8, 2, 8).permute(1, 3, 0, 2, 4).reshape(4, n, 64)                # [quadrant, entry, pixel]
    for qd in range(4):
        Ak = Aq[qd][Aq[qd].any(dim=1)]                                                    # the entries the quadrant's wave walks
        nk = Ak.shape[0]
        if nk == 0:
            continue
        chain_bound += int(Ak.sum(dim=0).max())
        for kk in comp_steps:
            pad = (-nk) % kk
            Ap = torch.cat([Ak, torch.zeros(pad, 64, dtype=Ak.dtype, device=dev)]) if pad else Ak
            comp_steps[kk] += int(Ap.reshape(-1, kk, 64).sum(dim=1).max(dim=1).values.sum())
    steps_pair84 += int(h84.max(dim=1).values.sum())
    steps_pair48 += int(h48.max(dim=2).values.sum())
    steps_quad44 += int(b44.permute(0, 2, 1, 3).reshape(2, 2, 4).max(dim=2).values.sum())
    rect84 += int(A.reshape(n, 4, 4, 2, 8).any(dim=4).any(dim=2).sum())   # blocks 8 wide x 4 high
    rect48 += int(A.reshape(n, 2, 8, 4, 4).any(dim=4).any(dim=2).sum())
    # bounding box of the alpha >= 1/255 ellipse (q <= tau = log(255 o)): half extents sqrt(2 tau C / det), sqrt(2 tau A / det)
    tau = torch.log(255.0 * cn[:, 3]).clamp(min=0)
    det = (cn[:, 0] * cn[:, 2] - cn[:, 1] ** 2).clamp(min=1e-12)
    hx = torch.sqrt(2 * tau * cn[:, 2] / det) + 1e-3
    hy = torch.sqrt(2 * tau * cn[:, 0] / det) + 1e-3
    mx, my = xy[ids, 0], xy[ids, 1]
    bx = torch.arange(4, device=dev).float() * 4 + tx * 16
    by = torch.arange(4, device=dev).float() * 4 + ty * 16
    ox = ((mx + hx)[:, None] >= bx[None]) & ((mx - hx)[:, None] <= bx[None] + 3)     # [n, 4] block columns
    oy = ((my + hy)[:, None] >= by[None]) & ((my - hy)[:, None] <= by[None] + 3)
    ok = (cn[:, 3] >= 1 / 255.0)[:, None, None]
    bb = oy[:, :, None] & ox[:, None, :] & ok                                                  # [n, 4(y), 4(x)]
    # restricted to entries before the block's last contributor, like the kernel's staging
    lastb = last.reshape(16, 16).reshape(4, 4, 4, 4).permute(0, 2, 1, 3).reshape(4, 4, 16).max(dim=2).values  # [4(y),4(x)]
    bb = bb & (ordv.reshape(n, 1, 1) < lastb[None])
    bbox4 += int(bb.sum())
    A8 = A.reshape(n, 2, 8, 2, 8).any(dim=4).any(dim=2)                                        # exact 8x8
    bbq = bb & A8.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    bbox4q += int(bbq.sum())
print(f"R={R}  list pairs (R*256)={tot_pairs/1e6:.1f} M   active pairs={act_pairs/1e6:.2f} M ({act_pairs/tot_pairs:.3f})")
print(f"  perfect 8x4: {rect84/1e6:.3f} M  4x8: {rect48/1e6:.3f} M   bbox 4x4: {bbox4/1e6:.3f} M   bbox & exact-8x8: {bbox4q/1e6:.3f} M")
print(f"  WAVE steps with perfect culls: one list per 8x8 quadrant {steps_8x8/1e6:.3f} M | two 8x4 halves, max {steps_pair84/1e6:.3f} M | "
      f"two 4x8 halves, max {steps_pair48/1e6:.3f} M | four 4x4 blocks, max {steps_quad44/1e6:.3f} M")
print("  serial steps of a quadrant wave if each pixel lane walked only ITS active entries inside windows of k kept entries (max over the 64 lanes per window): "
      + " | ".join(f"k={kk}: {v/1e6:.3f} M ({v/steps_8x8:.3f} of the {steps_8x8/1e6:.3f} M entry steps)" for kk, v in comp_steps.items())
      + f" | whole list (busiest pixel's chain): {chain_bound/1e6:.3f} M ({chain_bound/steps_8x8:.3f})")
for bs, s in steps.items():
    print(f"  perfect cull at {bs}x{bs}: {s/1e6:.3f} M (entry, block) steps = {s*bs*bs/1e6:.1f} M lane slots, useful {act_pairs/(s*bs*bs):.3f}")
