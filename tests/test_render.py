"""Host-side renderer: projection geometry and image contract of render('rgb_array') (bullet_env.py:156-185)."""
import numpy as np

from oracle import COracle
from pioneer_amd.config import RenderConfig
from pioneer_amd.render import link_origins, project, render_rgb


def test_link_origins_end_at_the_oracle_pointer():
    o = COracle(1)
    rng = np.random.RandomState(0)
    for _ in range(20):
        q = rng.uniform(o.r_lo, o.r_hi)
        pts = link_origins(q)
        assert pts.shape == (12, 3) and np.abs(pts[-1] - o.fk([q])[0]).max() < 1e-12


def test_projection_centres_the_camera_target():
    cfg = RenderConfig(camera_target=(3.0, -2.0, 5.0), camera_distance=40.0)
    px, depth = project([cfg.camera_target], cfg)
    assert np.allclose(px[0], [cfg.render_width / 2, cfg.render_height / 2]) and abs(depth[0] - 40.0) < 1e-9
    # a point above the target (world +z) appears higher in the image (smaller y)
    up, _ = project([(3.0, -2.0, 9.0)], cfg)
    assert up[0][1] < px[0][1]
    behind, _ = project([np.array(cfg.camera_target) + 1000.0], cfg)
    assert np.isnan(behind).all()        # outside the far plane


def test_render_rgb_contract():
    cfg = RenderConfig(camera_distance=60.0, render_width=320, render_height=200)
    img = render_rgb(np.zeros(6), (20.0, 0.0, 4.0), cfg)
    assert img.shape == (200, 320, 3) and img.dtype == np.uint8
    assert (img == np.array([255, 0, 0], np.uint8)).all(axis=2).any()        # the target disc
    assert (img == np.array([26, 230, 26], np.uint8)).all(axis=2).any()      # the pointer
    moved = render_rgb(np.array([0.5, 0.3, -0.4, 1.0, 0.2, 0.0]), (20.0, 0.0, 4.0), cfg)
    assert (moved != img).any()
