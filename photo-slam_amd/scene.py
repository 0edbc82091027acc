"""Synthetic Gaussian clouds + keyframe cameras for parity tests and benchmarks.

There is no dataset offline, so BASELINE.json's configs are realised as seeded
synthetic scenes.  Camera tensors follow the reference's
conventions exactly (src/gaussian_keyframe.cpp:119-204, include/graphics_utils.h:48-51):

  world_view_transform_ = W2C^T                (row-major [4,4]; flat[4c+r] = W2C(r,c))
  full_proj_transform_  = (Proj . W2C)^T
  camera_center_        = C2W[:3, 3]
  Proj: P00=1/tan(fovx/2), P11=1/tan(fovy/2), P22=zf/(zf-zn), P23=-zf*zn/(zf-zn), P32=1

numpy only (no torch) so the oracle tests can use it without a GPU stack.
"""
import math
from dataclasses import dataclass, field

import numpy as np

# name: P, W, H, fx, fy  (BASELINE.json configs[0..4])
CONFIGS = {
    "C1": dict(P=50_000, W=640, H=480, fx=535.4, fy=539.2, note="tiny COLMAP-like, CPU oracle case"),
    "C2": dict(P=500_000, W=1200, H=680, fx=600.0, fy=600.0, note="Replica office0 shape"),
    "C3": dict(P=2_000_000, W=1920, H=1080, fx=960.0, fy=960.0, note="Replica room0 @1080p shape"),
    "C4": dict(P=2_000_000, W=640, H=480, fx=535.4, fy=539.2, note="TUM fr3_office shape, 8-keyframe batch"),
    "C5": dict(P=4_000_000, W=752, H=480, fx=458.654, fy=457.296, note="EuRoC MH_01 shape, SH degree 3"),
}

ZNEAR, ZFAR = 0.01, 100.0  # cfg/gaussian_mapper/RGB-D/Replica/replica_rgbd.yaml:16-17


def focal2fov(focal, pixels):
    """include/graphics_utils.h:48-51"""
    return 2.0 * math.atan(pixels / (2.0 * focal))


def projection_matrix(znear, zfar, fovx, fovy):
    """GaussianKeyframe::getProjectionMatrix, src/gaussian_keyframe.cpp:176-204 (math matrix, not transposed)."""
    tan_y, tan_x = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class Camera:
    """The per-view inputs GaussianRenderer::render assembles (src/gaussian_renderer.cpp:51-66)."""
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray   # [4,4] = W2C^T  (world_view_transform_)
    projmatrix: np.ndarray   # [4,4] = (Proj W2C)^T (full_proj_transform_)
    campos: np.ndarray       # [3]


def make_camera(W, H, fx, fy, R_c2w, cam_center):
    fovx, fovy = focal2fov(fx, W), focal2fov(fy, H)
    c2w = np.eye(4, dtype=np.float64)
    c2w[:3, :3] = R_c2w
    c2w[:3, 3] = cam_center
    w2c = np.linalg.inv(c2w).astype(np.float32)
    proj = projection_matrix(ZNEAR, ZFAR, fovx, fovy)
    view_t = np.ascontiguousarray(w2c.T)
    full_t = np.ascontiguousarray((proj @ w2c).T.astype(np.float32))
    return Camera(W, H, float(np.float32(math.tan(fovx * 0.5))), float(np.float32(math.tan(fovy * 0.5))), view_t,
                  full_t, np.asarray(cam_center, np.float32).copy())


def look_rotation(yaw, pitch):
    """Camera-to-world rotation; camera looks down +z, x right, y down (COLMAP/3DGS)."""
    fwd = np.array([math.cos(pitch) * math.sin(yaw), math.sin(pitch), math.cos(pitch) * math.cos(yaw)])
    up_hint = np.array([0.0, -1.0, 0.0])
    right = np.cross(up_hint, fwd)  # so that (right, down, fwd) is right-handed
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    return np.stack([right, down, fwd], axis=1)


@dataclass
class Cloud:
    """Raw (pre-activation) parameters, as GaussianModel stores them (include/gaussian_model.h)."""
    xyz: np.ndarray            # [P,3]
    features_dc: np.ndarray    # [P,1,3]
    features_rest: np.ndarray  # [P,15,3]
    scaling: np.ndarray        # [P,3]  log-scale
    rotation: np.ndarray       # [P,4]  unnormalised quaternion (r,x,y,z)
    opacity: np.ndarray        # [P,1]  logit
    cameras: list = field(default_factory=list)
    extent: float = 1.0

    # activations, src/gaussian_model.cpp:48-71
    def get_scaling(self):
        return np.exp(self.scaling).astype(np.float32)

    def get_rotation(self):
        n = np.linalg.norm(self.rotation, axis=1, keepdims=True)
        return (self.rotation / np.maximum(n, 1e-12)).astype(np.float32)

    def get_opacity(self):
        return (1.0 / (1.0 + np.exp(-self.opacity.astype(np.float64)))).astype(np.float32)

    def get_features(self):
        return np.concatenate([self.features_dc, self.features_rest], axis=1).astype(np.float32)


BOX = np.array([3.0, 1.5, 3.0])  # half extents: room-scale box like Replica


def make_cloud(P, W, H, fx, fy, seed=0, n_views=1, scale_k=0.2, sh_rest_sigma=0.05):
    """Deterministic room-scale cloud: xyz ~ U(box), 30% snapped to the
    6 walls; anisotropic log-normal scales around s = scale_k * (vol/P)^(1/3) (scale_k chosen
    so that instances-per-visible-Gaussian R/V is 5-10 at 16-px tiles, like a converged map);
    random rotations; opacity logit ~ N(0,2^2); SH dc ~ N(0,0.5^2), rest ~ N(0,0.05^2).
    Cameras: random yaw, +-15 deg pitch, backed against the wall behind them, n_views poses on a 1 m arc."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xyz = (rng.random((P, 3)) * 2 - 1) * BOX
    snap = rng.random(P) < 0.3
    axis = rng.integers(0, 3, P)
    side = rng.integers(0, 2, P) * 2 - 1
    idx = np.nonzero(snap)[0]
    xyz[idx, axis[idx]] = side[idx] * BOX[axis[idx]]
    vol = float(np.prod(2 * BOX))
    s_bar = scale_k * (vol / max(P, 1)) ** (1.0 / 3.0)
    scaling = np.log(s_bar) + 0.5 * rng.standard_normal((P, 3))
    rotation = rng.standard_normal((P, 4))
    opacity = 2.0 * rng.standard_normal((P, 1))
    f_dc = 0.5 * rng.standard_normal((P, 1, 3))
    f_rest = sh_rest_sigma * rng.standard_normal((P, 15, 3))
    cams = []
    # camera backed against the wall behind it (75% of the half extent), looking across the room,
    # so that a keyframe sees roughly half of the map (V/P ~ 0.4-0.5) like an indoor SLAM keyframe.
    yaw0 = rng.random() * 2 * math.pi
    pitch = (rng.random() * 2 - 1) * math.radians(15)
    fwd_h = np.array([math.sin(yaw0), 0.0, math.cos(yaw0)])
    center = -0.75 * fwd_h * BOX + (rng.random(3) * 2 - 1) * BOX * 0.15
    for v in range(n_views):
        a = (v / max(n_views - 1, 1) - 0.5) if n_views > 1 else 0.0  # 1 m arc
        c = center + np.array([math.cos(yaw0), 0.0, -math.sin(yaw0)]) * a
        cams.append(make_camera(W, H, fx, fy, look_rotation(yaw0 + 0.35 * a, pitch), c))
    return Cloud(xyz.astype(np.float32), f_dc.astype(np.float32), f_rest.astype(np.float32),
                 scaling.astype(np.float32), rotation.astype(np.float32), opacity.astype(np.float32), cams,
                 extent=float(np.linalg.norm(BOX)))


def make_config(name, seed=0, n_views=1, P=None, **kw):
    c = CONFIGS[name]
    return make_cloud(P if P is not None else c["P"], c["W"], c["H"], c["fx"], c["fy"], seed=seed, n_views=n_views,
                      **kw)
