"""Fixture ``reference_gif_demo11.json`` -- numbers READ OFF ``images/OBCA_dynObs_demo11.gif`` of the reference repository: a screen
recording (2520 x 1080, 180 frames = 60 animation frames shown three times, no title) of the reference's own closed loop on its
checked-in setting ``demo11`` (src/demo_setting.py:248-269: corridor 80 x 10, start (3, 4, 0), goal (77, 4, 0), 3 x 3 obstacles
moving up at x = 30.5 and down at x = 39.5 with 0.1 m/s), drawn by ``plotClass.fullDimension_closedLoop_animate`` (src/draw.py:333-456):
the closed-loop poses ``x_closed[0:k+1]`` as orange markers (src/draw.py:407).  The same run is Figure 11 of the project report,
whose four frame titles tests/golden/reference_report_figures.json holds.

The recording is palette-dithered, so the markers are located on a smoothed "orangeness" image by a greedy disc fit (largest
response of a disc of the marker's radius, remove it, repeat); centres are good to about 0.15 m (the markers partly overlap and
sit on the blue dots of the reference path).  Pixel -> metre: the x axis from the two vertical spines (0 and 80 m), the same scale
for y (equal aspect), y = 4 on the row of the blue reference-path dots (the A* path of this corridor is the line y = 4).  The
first two poses (under the start marker) and the last ones (under the car) cannot be read: marker i is pose i + 2.
Data only: no reference source text or image is stored.

    python tests/golden/make_gif_demo11_fixture.py        # rewrites tests/golden/reference_gif_demo11.json
"""
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
GIF = "/root/reference/images/OBCA_dynObs_demo11.gif"
RADIUS = 10.0          # pixels: a marker of markersize 3 on this recording
FIRST_POSE = 2         # poses 0 and 1 lie under the start marker


def closeness(rgb, colour, width=120.0):
    """1 at the colour, 0 at `width` (Euclidean, RGB) away from it -- on the image smoothed over the dither pattern"""
    sm = np.stack([ndimage.gaussian_filter(rgb[..., c], 1.5) for c in range(3)], -1)
    d = np.sqrt(((sm - np.asarray(colour, float)) ** 2).sum(-1))
    return np.clip(1.0 - d / width, 0.0, 1.0)


def greedy_discs(o, thr=0.45):
    o = o.copy()
    R = int(RADIUS) + 2
    gy, gx = np.mgrid[-R:R + 1, -R:R + 1]
    disc = ((gy ** 2 + gx ** 2) <= RADIUS ** 2).astype(float)
    out = []
    while True:
        score = ndimage.correlate(o, disc / disc.sum())
        py, px = np.unravel_index(np.argmax(score), score.shape)
        if score[py, px] < thr:
            break
        w = o[py - R:py + R + 1, px - R:px + R + 1] * disc
        out.append((px + (w * gx).sum() / w.sum(), py + (w * gy).sum() / w.sum()))
        o[py - R:py + R + 1, px - R:px + R + 1] *= 1.0 - ((gy ** 2 + gx ** 2) <= (RADIUS + 1.5) ** 2)
    return out


def main():
    im = Image.open(GIF)
    n_frames = im.n_frames
    im.seek(n_frames - 1)
    rgb = np.asarray(im.convert("RGB")).astype(float)
    dark = rgb.sum(2) < 150
    cols = np.where(dark.sum(0) > 200)[0]
    x0p, x1p = cols[cols < 1000].mean(), cols[cols > 1000].mean()
    sx = (x1p - x0p) / 80.0
    blue = greedy_discs(closeness(rgb, (65, 105, 225)), thr=0.40)          # 'royalblue': the reference path, y = 4
    row4 = float(np.median([p[1] for p in blue]))
    pts = sorted(greedy_discs(closeness(rgb, (255, 165, 0))))
    markers = [[round(float((px - x0p) / sx), 3), round(float(4.0 + (row4 - py) / sx), 3)] for px, py in pts]
    doc = {"source": "images/OBCA_dynObs_demo11.gif of the reference repository, last frame (screen recording of its closed loop on demo11, "
                     "src/demo_setting.py:248-269; the run of report Figure 11)",
           "setting": "problemSetting('demo11'), closedLoop defaults as checked in (N_free = N_fix = 6, senseDis = 10), no stop at k = 30",
           "frames": n_frames, "animation_frames": n_frames // 3,
           "markers_xy": markers, "first_marker_is_pose": FIRST_POSE, "metres_per_pixel": float(1.0 / sx),
           "reference_path_dots_found": len(blue),
           "accuracy": "about 0.15 m per marker (dithered recording, markers overlap each other and the path dots)"}
    with open(os.path.join(HERE, "reference_gif_demo11.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("frames", n_frames, "markers", len(markers), "path dots", len(blue), "row of y = 4:", row4)


if __name__ == "__main__":
    main()
