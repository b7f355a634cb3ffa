"""Host-side software renderer for ``render('rgb_array')``.

The reference renders through Bullet's OpenGL camera (bullet_env.py:156-185); there is no GL
here, so this draws a stick figure: the arm's link frames joined by line segments, the pointer
and the target sphere, seen by the camera of ``RenderConfig`` (yaw/pitch/roll about
``camera_target`` at ``camera_distance``, z up; vertical FOV projection with near/far clipping).
Pure NumPy on the host — rendering is not on the step path and touches no env state.
"""
import numpy as np

from . import model
from .config import RenderConfig


def _rot(axis, q):
    x, y, z = axis
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=np.float64)
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


def link_origins(joint_positions) -> np.ndarray:
    """World positions of every link frame origin along the chain (12 points incl. world)."""
    R, p, qi = np.eye(3), np.zeros(3), 0
    pts = [p.copy()]
    for j in model.JOINTS:
        p = p + R @ np.asarray(j.xyz, dtype=np.float64)
        if j.type == "revolute":
            R = R @ _rot(j.axis, float(joint_positions[qi]))
            qi += 1
        pts.append(p.copy())
    return np.array(pts)


def view_matrix(cfg: RenderConfig) -> np.ndarray:
    """Camera pose from yaw/pitch/roll (degrees) about the target, z up."""
    yaw, pitch, roll = np.radians([cfg.camera_yaw, cfg.camera_pitch, cfg.camera_roll])
    target = np.asarray(cfg.camera_target, dtype=np.float64)
    # start looking along +y from distance d behind the target, then pitch about x, yaw about z
    fwd = _rot((0, 0, 1), yaw) @ _rot((1, 0, 0), pitch) @ np.array([0.0, 1.0, 0.0])
    up0 = _rot((0, 0, 1), yaw) @ _rot((1, 0, 0), pitch) @ np.array([0.0, 0.0, 1.0])
    up = _rot(tuple(fwd), roll) @ up0
    eye = target - cfg.camera_distance * fwd
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    V = np.eye(4)
    V[0, :3], V[1, :3], V[2, :3] = right, up, -fwd
    V[:3, 3] = -V[:3, :3] @ eye
    return V


def project(points, cfg: RenderConfig):
    """World points [k,3] -> pixel coords [k,2], depth [k] (NaN where behind the near plane)."""
    V = view_matrix(cfg)
    pc = (V[:3, :3] @ np.asarray(points, dtype=np.float64).T).T + V[:3, 3]
    depth = -pc[:, 2]
    f = 1.0 / np.tan(np.radians(cfg.projection_fov) / 2)
    aspect = cfg.render_width / cfg.render_height
    with np.errstate(divide="ignore", invalid="ignore"):
        ndc_x = (f / aspect) * pc[:, 0] / depth
        ndc_y = f * pc[:, 1] / depth
    ok = (depth > cfg.projection_near) & (depth < cfg.projection_far)
    px = np.stack([(ndc_x + 1) * 0.5 * cfg.render_width, (1 - ndc_y) * 0.5 * cfg.render_height], axis=1)
    px[~ok] = np.nan
    return px, depth


def _line(img, a, b, color, width):
    if np.isnan(a).any() or np.isnan(b).any():
        return
    n = int(max(abs(b[0] - a[0]), abs(b[1] - a[1]))) + 1
    xs = np.linspace(a[0], b[0], n); ys = np.linspace(a[1], b[1], n)
    for dx in range(-(width // 2), width // 2 + 1):
        for dy in range(-(width // 2), width // 2 + 1):
            xi = np.clip(np.round(xs + dx).astype(int), 0, img.shape[1] - 1)
            yi = np.clip(np.round(ys + dy).astype(int), 0, img.shape[0] - 1)
            inside = (xs + dx >= 0) & (xs + dx < img.shape[1]) & (ys + dy >= 0) & (ys + dy < img.shape[0])
            img[yi[inside], xi[inside]] = color


def _disc(img, c, radius, color):
    if np.isnan(c).any():
        return
    r = max(1, int(round(radius)))
    y0, y1 = max(0, int(c[1]) - r), min(img.shape[0], int(c[1]) + r + 1)
    x0, x1 = max(0, int(c[0]) - r), min(img.shape[1], int(c[0]) + r + 1)
    if y0 >= y1 or x0 >= x1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    m = (yy - c[1]) ** 2 + (xx - c[0]) ** 2 <= r * r
    img[y0:y1, x0:x1][m] = color


def render_rgb(joint_positions, target_position, cfg: RenderConfig = None, target_radius: float = 0.2) -> np.ndarray:
    """uint8 image [render_height, render_width, 3]."""
    cfg = cfg or RenderConfig()
    img = np.full((cfg.render_height, cfg.render_width, 3), 255, dtype=np.uint8)
    # ground grid for depth cue
    g = np.arange(-30, 31, 10.0)
    for v in g:
        (a, b), _ = project([[v, -30, 0], [v, 30, 0]], cfg)
        _line(img, a, b, (225, 225, 225), 1)
        (a, b), _ = project([[-30, v, 0], [30, v, 0]], cfg)
        _line(img, a, b, (225, 225, 225), 1)
    pts = link_origins(joint_positions)
    px, depth = project(pts, cfg)
    colors = {True: (18, 64, 138), False: (255, 128, 13)}      # arm / hinge colours of the reference's materials
    for i in range(len(pts) - 1):
        if np.allclose(pts[i], pts[i + 1]):
            continue
        _line(img, px[i], px[i + 1], colors[i % 2 == 0], 5)
    f = 1.0 / np.tan(np.radians(cfg.projection_fov) / 2) * cfg.render_height / 2
    _disc(img, px[-1], max(2.0, 0.2 * f / max(depth[-1], 1e-6) * 3), (26, 230, 26))          # pointer
    (tpx,), (td,) = project([target_position], cfg)
    _disc(img, tpx, max(2.0, target_radius * f / max(td, 1e-6) * 3), (255, 0, 0))             # target
    return img
