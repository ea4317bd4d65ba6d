"""Dense (non-tiled per-pixel-vectorised) differentiable restatement of the rasterizer in
torch float64 -- an independent check of the C oracle's forward AND of its hand-derived
backward (SURVEY.md section 8c cross-check (iii)).  O(P*H*W); only for tiny scenes.

Discrete decisions (tile rectangles = which pixels a Gaussian may touch) are taken from the
oracle state as constants, exactly like the reference treats radii/rects as non-differentiable.
"""
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh_rgb(deg, sh, dirs):
    # forward.cu:20-71 semantics, sh [P,16,3], dirs [P,3] unit
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def dense_render(means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, bg,
                 W, H, tan_fovx, tan_fovy, sh_degree, rects, radii, scale_modifier=1.0,
                 colors_precomp=None, cov3D_precomp=None):
    """All tensor args float64 torch (leafs may require grad).  rects [P,4] int (minx,miny,maxx,maxy
    in tile units), radii [P] int: constants from the oracle.  Returns color[3,H,W], depth[H,W], alpha[H,W]."""
    dt = means3D.dtype
    P = means3D.shape[0]
    vm = viewmatrix.reshape(4, 4)  # row-vector convention: p_row @ vm
    pm = projmatrix.reshape(4, 4)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1) @ pm
    pw = 1.0 / (ph[:, 3] + 0.0000001)
    ndc = ph[:, :2] * pw[:, None]
    t = torch.cat([means3D, ones], 1) @ vm
    tz = t[:, 2]
    if cov3D_precomp is None:
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rstd = torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
            torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
            torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)  # [P,3,3]
        S = torch.diag_embed(scale_modifier * scales)
        L = Rstd @ S
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1),
                             torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                             torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], 1)
    fx = W / (2.0 * tan_fovx)
    fy = H / (2.0 * tan_fovy)
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    txc = torch.clamp(t[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(t[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * txc) / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -(fy * tyc) / (tz * tz)], -1)], 1)  # [P,2,3]
    Wr = vm[:3, :3].T  # world->view rotation (column-vector convention)
    cov2 = J @ Wr @ Sigma @ Wr.T @ J.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c2 = cov2[:, 1, 1] + 0.3
    det = a * c2 - b * b
    conA, conB, conC = c2 / det, -b / det, a / det
    mx = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    my = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    if colors_precomp is None:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = eval_sh_rgb(sh_degree, shs, d)
    else:
        rgb = colors_precomp
    vis = torch.as_tensor(radii) > 0
    order = sorted([i for i in range(P) if bool(vis[i])], key=lambda i: (float(tz[i].detach().to(torch.float32)), i))
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    tyi, txi = ys // 16, xs // 16
    pxf, pyf = xs.to(dt), ys.to(dt)
    T = torch.ones(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    C = [torch.zeros(H, W, dtype=dt) for _ in range(3)]
    Wt = torch.zeros(H, W, dtype=dt)
    D = torch.zeros(H, W, dtype=dt)
    rects = torch.as_tensor(rects)
    for i in order:
        rx0, ry0, rx1, ry1 = [int(v) for v in rects[i]]
        inrect = (txi >= rx0) & (txi < rx1) & (tyi >= ry0) & (tyi < ry1) & ~done
        if not bool(inrect.any()):
            continue
        dx = mx[i] - pxf
        dy = my[i] - pyf
        power = -0.5 * (conA[i] * dx * dx + conC[i] * dy * dy) - conB[i] * dx * dy
        alpha = torch.clamp_max(opacities[i] * torch.exp(power), 0.99)
        valid = inrect & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        newly_done = valid & (test_T < 0.0001)
        contrib = valid & ~newly_done
        w = torch.where(contrib, alpha * T, torch.zeros_like(T))
        for ch in range(3):
            C[ch] = C[ch] + rgb[i, ch] * w
        Wt = Wt + w
        D = D + tz[i] * w
        T = torch.where(contrib, test_T, T)
        done = done | newly_done
    color = torch.stack([C[ch] + T * bg[ch] for ch in range(3)], 0)
    return color, D, Wt
