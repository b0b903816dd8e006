"""Software stick-figure renderer: the `render(mode="rgb_array")` of the environments without a MuJoCo viewer.

The reference renders through mujoco-py's OpenGL viewer (src/deepmimic_env.py:301 MujocoEnv.render, used by the eval dashboard
of src/sb3_ppo.py:44).  Here a frame is an orthographic side view (x forward, z up, camera following the root) of the body tree:
one line per body from its parent's origin to its own, the floor line and a tick every metre — enough for the dashboard's
"is it walking / lying / getting up" panel.  Pure PIL + numpy, host side.
"""
from __future__ import annotations

import numpy as np


def stick_figure(xpos, parents, size=(320, 240), cam_x=None, span=2.4):
    """xpos [nbody, 3] world positions (body 0 = world), parents [nbody] -> uint8 [H, W, 3]."""
    from PIL import Image, ImageDraw
    W, H = size
    xpos = np.asarray(xpos, float)
    cam_x = float(xpos[1, 0]) if cam_x is None else cam_x
    sc = W / span

    def px(p):
        return (W / 2 + (p[0] - cam_x) * sc, H * 0.85 - p[2] * sc)

    img = Image.new("RGB", (W, H), (235, 238, 242))
    d = ImageDraw.Draw(img)
    d.line([(0, H * 0.85), (W, H * 0.85)], fill=(90, 110, 90), width=2)
    for m in range(int(np.floor(cam_x - span)), int(np.ceil(cam_x + span)) + 1):
        x = W / 2 + (m - cam_x) * sc
        d.line([(x, H * 0.85), (x, H * 0.85 + 6)], fill=(90, 110, 90), width=1)
    depth = np.argsort(-xpos[:, 1])          # far side first
    for b in depth:
        p = int(parents[b])
        if b < 2 or p < 1:
            continue
        shade = int(np.clip(120 - 200 * xpos[b, 1], 40, 200))
        d.line([px(xpos[p]), px(xpos[b])], fill=(shade, shade // 2, 40), width=3)
    r = 3
    for b in range(1, len(xpos)):
        x, y = px(xpos[b])
        d.ellipse([x - r, y - r, x + r, y + r], fill=(30, 60, 160))
    return np.asarray(img, np.uint8)
