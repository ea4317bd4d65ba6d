"""Synthetic raster workloads of SURVEY.md section 8(d) (numpy only, seeded).

C1: 1k Gaussians, 128x128, single view (BASELINE.json configs[0]).
C2: ~200k Gaussians on the walls of a 6x4x3 m room, 640x480, 6 ring cameras with
    Replica-like intrinsics fx=fy=320 (BASELINE.json configs[1]).

Camera conventions follow the reference: world_view_transform is the transposed
world-to-view matrix (scene/cameras.py:60), the projection is the reference's
non-standard P with P[2,2]=P[3,2]=1 (utils/graphics_utils.py:51-75), and
full_proj_transform = world_view_transform @ projection^T (scene/cameras.py:61-62).
"""
import math

import numpy as np

SH_C0 = 0.28209479177387814


def rgb2sh(rgb):
    return (rgb - 0.5) / SH_C0


def look_at(eye, target, up=(0.0, -1.0, 0.0)):
    """Returns the 4x4 world-to-view matrix (column-vector convention), camera looks down +z,
    x right, y down (COLMAP convention used by scene/dataset_readers.py)."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    upv = np.asarray(up, np.float64)
    r = np.cross(-upv, f)  # y is down => "up" is -y
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    Rc2w = np.stack([r, d, f], axis=1)  # columns: camera axes in world
    w2v = np.eye(4)
    w2v[:3, :3] = Rc2w.T
    w2v[:3, 3] = -Rc2w.T @ eye
    return w2v


def make_camera(w2v, fovx, fovy, W, H):
    """dict with the tensors GaussianRasterizationSettings wants (float32 numpy)."""
    tanx, tany = math.tan(fovx * 0.5), math.tan(fovy * 0.5)
    Pm = np.zeros((4, 4), np.float32)
    Pm[0, 0] = 1.0 / tanx
    Pm[1, 1] = 1.0 / tany
    Pm[2, 2] = 1.0
    Pm[3, 2] = 1.0
    wvt = np.float32(w2v).T.copy()  # world_view_transform
    proj_t = Pm.T.copy()
    full = (wvt @ proj_t).astype(np.float32)
    campos = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    return dict(viewmatrix=wvt, projmatrix=full, campos=campos, tanfovx=tanx, tanfovy=tany,
                image_width=W, image_height=H, FoVx=fovx, FoVy=fovy)


def _quats(rng, P):
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def _sh(rng, P, std=0.2):
    sh = rng.normal(0.0, std, size=(P, 16, 3))
    sh[:, 0, :] = rgb2sh(rng.uniform(0.0, 1.0, size=(P, 3)))
    return sh.astype(np.float32)


def scene_c1(P=1000, W=128, H=128, seed=0, sh_degree=3):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.0, 1.0, size=(P, 3))
    xyz[:, 2] = xyz[:, 2] * 2.0 + 4.0  # z in [2,6] in camera space (camera at origin, identity view)
    scales = np.exp(rng.normal(math.log(0.05), 0.3, size=(P, 3))).astype(np.float32)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.0, size=(P, 1))))).astype(np.float32)
    fov = math.radians(60.0)
    cam = make_camera(look_at((0.0, 0.0, 0.0), (0.02, -0.01, 1.0)), fov, fov, W, H)
    return dict(means3D=xyz.astype(np.float32), scales=scales, rotations=_quats(rng, P), opacities=opac,
                shs=_sh(rng, P), sh_degree=sh_degree, bg=np.zeros(3, np.float32), cameras=[cam])


def scene_c2(P=200_000, W=640, H=480, seed=0, sh_degree=3, n_cams=6, scale_mu=0.02):
    """Room box 6 x 4 x 3 m (x,z footprint 6x4, height 3 along -y up), points on its 6 faces."""
    rng = np.random.default_rng(seed)
    ext = np.array([6.0, 3.0, 4.0])  # x, y(height), z
    areas = np.array([ext[1] * ext[2], ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[2], ext[0] * ext[1], ext[0] * ext[1]])
    face = rng.choice(6, size=P, p=areas / areas.sum())
    u = rng.uniform(-0.5, 0.5, size=(P, 3)) * ext
    axis = face // 2
    sign = np.where(face % 2 == 0, -0.5, 0.5)
    u[np.arange(P), axis] = sign * ext[axis]
    xyz = u + rng.normal(0.0, 0.02, size=(P, 3))
    scales = np.exp(rng.normal(math.log(scale_mu), 0.5, size=(P, 3))).astype(np.float32)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.5, size=(P, 1))))).astype(np.float32)
    fx = 320.0 * W / 640.0
    fovx = 2.0 * math.atan(W / (2.0 * fx))
    fovy = 2.0 * math.atan(H / (2.0 * fx))
    cams = []
    for k in range(n_cams):
        a = 2.0 * math.pi * k / n_cams
        # on a ring close to the walls, looking across the room (sees the far wall + floor/ceiling/sides)
        eye = (2.4 * math.cos(a), 0.15 * math.sin(3 * a), 1.5 * math.sin(a))
        tgt = (-0.6 * math.cos(a + 0.25), 0.05, -0.4 * math.sin(a + 0.25))
        cams.append(make_camera(look_at(eye, tgt), fovx, fovy, W, H))
    return dict(means3D=xyz.astype(np.float32), scales=scales, rotations=_quats(rng, P), opacities=opac,
                shs=_sh(rng, P), sh_degree=sh_degree, bg=np.zeros(3, np.float32), cameras=cams)
