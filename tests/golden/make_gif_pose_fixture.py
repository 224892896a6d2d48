"""Fixture ``reference_gif_demo9_poses.json`` -- what every FRAME of the reference's demo9 GIF draws besides its title
(``plotClass.fullDimension_closedLoop_animate``, src/draw.py:333-452; the titles and the stand-alone markers of the last frame are in
reference_gif_demo9.json, tests/golden/make_gif_fixture.py):

* frame k draws the car as a black 3.4 m x 1.5 m rectangle at the closed-loop pose ``xOpt[k]`` with heading ``xOpt[k, 2]``
  (src/draw.py:407-413, 431 ``carBox``) -- the only place the GIF shows theta.  Read here by fitting that rectangle's outline to
  the dark pixels of the frame (coarse-to-fine search around the previous frame's pose; the first frame's search starts at the
  start pose of the setting, an INPUT of the run): x, y, theta of all 84 closed-loop poses, to about a pixel (0.163 m) and 0.05 rad.
* frame k draws the open-loop plan of solve k, ``x_openLoop[k]`` -- the N + 1 = 6 poses IPOPT returned for that step -- as magenta
  markers joined by a line (src/draw.py:408).  Stored as the magenta ink of the frame (pixel runs), which the plan of this
  build must cover and stay on.

Data only: pixel read-outs and fitted numbers, no reference source text or image is stored.

    python tests/golden/make_gif_pose_fixture.py
"""
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage

from make_gif_fixture import GIF

HERE = os.path.dirname(os.path.abspath(__file__))
START_POSE = (1.0, 5.0, 0.0)            # problemSetting('demo9').startPose (src/demo_setting.py)
CAR_L, CAR_W = 3.4, 1.5                 # ego = [1.7, 0.75, 1.7, 0.75] (src/closed_loop.py:60): the box is centred on the pose


def axes_map(rgb):
    """pixel <-> metre map from the axes spines (0..40 m x 0..60 m), exactly as tests/golden/make_gif_fixture.py: read_markers -- the map
    the MARKERS follow (measured there: stand-alone marker centres against this build's poses, mean offset 0.00 / 0.02 m)"""
    dark = rgb.sum(2) < 120
    cols, rows = np.where(dark.sum(0) > 300)[0], np.where(dark.sum(1) > 200)[0]
    x0p, x1p, y1p, y0p = cols.min(), cols.max(), rows.min(), rows.max()
    return float(x0p), float(y0p), (x1p - x0p) / 40.0, (y0p - y1p) / 60.0


# static walls of demo9 (src/demo_setting.py, INPUTS of the run) long enough to be found by their length alone: the vertical edges
# x = 4 (y 10..55) and x = 34 (both blocks, y 14..30 and 34..49), the horizontal edges y = 55 (x 4..33) and y = 6 (x 8..40)
WALLS_X, WALLS_Y = (4.0, 34.0), (55.0, 6.0)


def _runs(idx):
    cut = np.flatnonzero(np.diff(idx) > 1) + 1
    return [float(r.mean()) for r in np.split(idx, cut)]


def line_map(rgb):
    """pixel <-> metre map from drawn LINES (the car box is made of lines of the same width): centres of the four long static walls.
    The axes spines themselves are 1 px wide and sit half a pixel off the 1.5 px lines inside (measured: 1.1 px at x = 6.7 m)."""
    dark = rgb.sum(2) < 120
    cols, rows = _runs(np.flatnonzero(dark.sum(0) > 150)), _runs(np.flatnonzero(dark.sum(1) > 150))
    # first and last run = the spines (0..40 m, 0..60 m): they tell which of the other runs is which wall
    near = lambda runs, frac: min(runs[1:-1], key=lambda r: abs(r - (runs[0] + frac * (runs[-1] - runs[0]))))
    c4, c34 = near(cols, WALLS_X[0] / 40.0), near(cols, WALLS_X[1] / 40.0)
    r55, r6 = near(rows, 1.0 - WALLS_Y[0] / 60.0), near(rows, 1.0 - WALLS_Y[1] / 60.0)
    assert c4 % 1 == 0.5 and c34 % 1 == 0.5 and r55 % 1 == 0.5 and r6 % 1 == 0.5          # each a two-pixel line
    sx, sy = (c34 - c4) / (WALLS_X[1] - WALLS_X[0]), (r6 - r55) / (WALLS_Y[0] - WALLS_Y[1])
    return c4 - WALLS_X[0] * sx, r6 + WALLS_Y[1] * sy, sx, sy


def outline(n=16):
    t = np.linspace(-1, 1, 2 * n + 1)
    pts = [(s * CAR_L / 2, sg * CAR_W / 2) for s in t for sg in (1, -1)] + [(sg * CAR_L / 2, s * CAR_W / 2) for s in t[::2] for sg in (1, -1)]
    return np.array(pts)


def fit_boxes(frames, box):
    """per frame: (x, y, theta, mean distance in pixels from the fitted outline to the nearest dark pixel, capped at 3)"""
    x0p, y0p, sx, sy = box
    O = outline()

    def score(dt, cx, cy, th):
        c, s = np.cos(th), np.sin(th)
        X = cx[:, None] + c[:, None] * O[None, :, 0] - s[:, None] * O[None, :, 1]
        Y = cy[:, None] + s[:, None] * O[None, :, 0] + c[:, None] * O[None, :, 1]
        px = np.clip(np.rint(x0p + X * sx).astype(int), 0, dt.shape[1] - 1)
        py = np.clip(np.rint(y0p - Y * sy).astype(int), 0, dt.shape[0] - 1)
        return np.minimum(dt[py, px], 3.0).mean(1)
    out, prev = [], np.array(START_POSE)
    for rgb in frames:
        dt = ndimage.distance_transform_edt(~(rgb.sum(2) < 200))
        # 2.4 m and 1.3 rad around the previous frame's pose (a step is at most 0.6 m/s x Tmax), then two refinements
        for dx, dth, nx, nth in ((0.2, 0.06, 12, 22), (0.05, 0.015, 5, 5), (0.0125, 0.004, 4, 4)):
            gx, gy, gt = prev[0] + dx * np.arange(-nx, nx + 1), prev[1] + dx * np.arange(-nx, nx + 1), prev[2] + dth * np.arange(-nth, nth + 1)
            G = np.array(np.meshgrid(gx, gy, gt, indexing="ij")).reshape(3, -1)
            sc = score(dt, G[0], G[1], G[2])
            i = int(sc.argmin())
            best, prev = (float(sc[i]),), G[:, i].copy()
        out.append([round(float(prev[0]), 3), round(float(prev[1]), 3), round(float(prev[2]), 4), round(best[0], 3)])
    return out


def magenta_runs(rgb):
    """[row, first column, last column] of every run of magenta pixels ('m' = (191, 0, 191), also where the translucent lidar disc
    is drawn over it)"""
    mg = (rgb[..., 0] > 120) & (rgb[..., 2] > 120) & (rgb[..., 1] < 110) & (np.abs(rgb[..., 0] - rgb[..., 2]) < 60)
    runs = []
    for r in np.flatnonzero(mg.any(1)):
        c = np.flatnonzero(mg[r])
        cut = np.flatnonzero(np.diff(c) > 1) + 1
        for seg in np.split(c, cut):
            runs.append([int(r), int(seg[0]), int(seg[-1])])
    return runs


def main():
    im = Image.open(GIF)
    frames = []
    for k in range(im.n_frames):
        im.seek(k)
        frames.append(np.asarray(im.convert("RGB")).astype(int))
    box = line_map(frames[-1])
    boxes = fit_boxes(frames, box)
    # frame 0 draws the start box and the car box on top of each other at the start pose
    assert abs(boxes[0][0] - 1.0) < 0.2 and abs(boxes[0][1] - 5.0) < 0.2 and abs(boxes[0][2]) < 0.1, boxes[0]
    amap = axes_map(frames[-1])
    as_dict = lambda m: {"x0": round(m[0], 4), "y0": round(m[1], 4), "per_metre_x": round(m[2], 5), "per_metre_y": round(m[3], 5)}
    doc = {"source": "images/" + os.path.basename(GIF) + " of the reference repository, every frame (src/draw.py:333-452)",
           "pixel_of_metre": dict(as_dict(amap), note="column = x0 + x * per_metre_x, row = y0 - y * per_metre_y; from the axes spines: the map the markers follow"),
           "pixel_of_metre_lines": dict(as_dict(box), note="the same from the centres of four static walls: the map the car box was fitted with"),
           "car_box": {"what": "per frame k: [x, y, theta, fit residual in pixels] of the car rectangle = closed-loop pose k (src/draw.py:410-431)",
                       "length_width": [CAR_L, CAR_W], "poses": boxes},
           "plan_ink": {"what": "per frame k: runs [row, first column, last column] of magenta pixels = the open-loop plan of solve k, 6 markers of "
                                "5 px joined by a 1.5 px line (src/draw.py:408); later artists (lidar disc, car box, start marker) lie on top",
                        "runs": [magenta_runs(f) for f in frames]}}
    with open(os.path.join(HERE, "reference_gif_demo9_poses.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
        f.write("\n")
    b = np.array(boxes)
    print("frames", len(boxes), "worst fit residual %.2f px" % b[:, 3].max(), "ink pixels per frame %d..%d" % (
        min(sum(r[2] - r[1] + 1 for r in fr) for fr in doc["plan_ink"]["runs"]), max(sum(r[2] - r[1] + 1 for r in fr) for fr in doc["plan_ink"]["runs"])))


if __name__ == "__main__":
    main()
