"""Fixture ``reference_gif_demo9.json`` -- numbers READ OFF the one solver output the reference repository holds.

``images/FullDim_dynObsAvoid_demo9_N5_SensorDis8_terminalDis = 4_N_83_ulimit_0.60_0.52.gif`` is what the reference's own
closed loop wrote (``closedLoop.closed_loop_mpc4`` -> ``plotClass.fullDimension_closedLoop_animate``, file name built at
src/closed_loop.py:439 and src/draw.py:450) for demo9 with the settings its author lists under "demo 9" in
src/simulation.py:68-74 -- CasADi/IPOPT/MUMPS solved every step of it.  Frame k carries

* the title ``'Spend Time = %.2f (sec)' % sum(Ts_opt[:k])`` (src/draw.py:380): the cumulative free time IPOPT returned,
* the closed-loop poses ``xOpt[0:k+1]`` as orange markers (src/draw.py:407).

This script (run HERE, where /root/reference exists; the fixture travels, the GIF does not) reads the 84 titles by matching
their digit glyphs against the same title rendered by matplotlib (same font, size and canvas), and the centres of the
markers of the last frame that do not touch a neighbour.  Data only: no reference source text is stored.

    python tests/golden/make_gif_fixture.py            # rewrites tests/golden/reference_gif_demo9.json
"""
import glob
import io
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
GIF = "/root/reference/images/FullDim_dynObsAvoid_demo9_N5_SensorDis8_terminalDis = 4_N_83_ulimit_0.60_0.52.gif"
TITLE_ROWS = slice(30, 56)          # the title line of a 640 x 480 canvas
N_PREFIX, N_SUFFIX = 10, 5          # glyph runs of "Spend Time =" and "(sec)"


def runs(mask):
    out, x = [], 0
    while x < len(mask):
        if mask[x]:
            x0 = x
            while x < len(mask) and mask[x]:
                x += 1
            out.append((x0, x))
        else:
            x += 1
    return out


def glyphs(gray):
    """ink of every glyph of the title line (grey levels 0..1), left to right"""
    ink = gray < 150
    ink[:, :200] = False            # the y axis' top tick label shares these rows
    out = []
    for x0, x1 in runs(ink.any(0)):
        r = np.where(ink[:, x0:x1].any(1))[0]
        out.append(1.0 - gray[r[0]:r[-1] + 1, x0:x1] / 255.0)
    return out


def render_title(text):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig = plt.figure(figsize=(6.4, 4.8), dpi=100)
    ax = fig.add_subplot(111)
    ax.set_aspect("equal", adjustable="box")
    plt.suptitle("Dynamic Avoidance With OBCA", fontweight="bold")
    plt.title(text)
    ax.set_xlim(0, 40)
    ax.set_ylim(0, 60)
    buf = io.BytesIO()
    fig.savefig(buf, format="png", dpi=100)
    plt.close(fig)
    return np.asarray(Image.open(buf).convert("L"))[TITLE_ROWS]


def glyph_distance(a, b):
    H, W = max(a.shape[0], b.shape[0]) + 2, max(a.shape[1], b.shape[1]) + 2
    A = np.zeros((H, W))
    A[1:1 + a.shape[0], 1:1 + a.shape[1]] = a
    best = np.inf
    for dy in range(H - b.shape[0] + 1):
        for dx in range(W - b.shape[1] + 1):
            B = np.zeros((H, W))
            B[dy:dy + b.shape[0], dx:dx + b.shape[1]] = b
            best = min(best, np.abs(A - B).sum())
    return best


def read_titles(im):
    tmpl = {}
    for grp in ("0 1 2 3 4", "5 6 7 8 9"):
        g = glyphs(render_title("Spend Time = %s (sec)" % grp))
        assert len(g) == N_PREFIX + 5 + N_SUFFIX, len(g)
        for i, d in enumerate(grp.split()):
            tmpl[d] = g[N_PREFIX + i]
    out = []
    for k in range(im.n_frames):
        im.seek(k)
        g = glyphs(np.asarray(im.convert("L"))[TITLE_ROWS])
        s = ""
        for q in g[N_PREFIX:-N_SUFFIX]:
            if q.shape[0] <= 4 and q.shape[1] <= 4:
                s += "."
            else:
                s += min(tmpl, key=lambda d: glyph_distance(q, tmpl[d]))
        assert len(s.split(".")) == 2 and len(s.split(".")[1]) == 2, (k, s)
        out.append(s)
    return out


def read_markers(im):
    """centres (map coordinates) of the orange closed-loop markers of the last frame that stand alone (a marker of
    markersize 3 covers 21 pixels on this canvas); the axes box gives the pixel <-> metre map (0..40 m x 0..60 m)."""
    im.seek(im.n_frames - 1)
    rgb = np.asarray(im.convert("RGB")).astype(int)
    dark = rgb.sum(2) < 120
    cols, rows = np.where(dark.sum(0) > 300)[0], np.where(dark.sum(1) > 200)[0]
    x0p, x1p, y1p, y0p = cols.min(), cols.max(), rows.min(), rows.max()
    sx, sy = (x1p - x0p) / 40.0, (y0p - y1p) / 60.0
    orange = (abs(rgb[..., 0] - 255) < 40) & (abs(rgb[..., 1] - 165) < 40) & (rgb[..., 2] < 80)
    lab, n = ndimage.label(orange)
    pts = []
    for i in range(1, n + 1):
        ys, xs = np.where(lab == i)
        if len(ys) == 21:
            pts.append([round(float((xs.mean() - x0p) / sx), 3), round(float((y0p - ys.mean()) / sy), 3)])
    pts.sort(key=lambda p: (p[1], p[0]))
    return pts, float(1.0 / sx)


GIF_DEMO1 = "/root/reference/images/OBCA_dynObs_demo1.gif"


def read_demo1_markers():
    """``images/OBCA_dynObs_demo1.gif``: a screen recording (2520 x 1080, no title, 147 frames) of the same animation for
    demo1 -- which version of the code and which settings produced it is not recorded; its last frame shows the whole
    closed-loop trajectory as orange markers.  Centres of the markers that stand alone, in map coordinates (axes box
    0..39 m x 0..10 m)."""
    im = Image.open(GIF_DEMO1)
    im.seek(im.n_frames - 1)
    rgb = np.asarray(im.convert("RGB")).astype(int)
    dark = rgb.sum(2) < 150
    cols, rows = np.where(dark.sum(0) > 300)[0], np.where(dark.sum(1) > 1200)[0]
    x0p, x1p = cols[cols < 1000].mean(), cols[cols > 1000].mean()
    y10p, y0p = rows.min() + 0.5, float(rows.max())
    sx, sy = (x1p - x0p) / 39.0, (y0p - y10p) / 10.0
    orange = (rgb[..., 0] > 200) & (abs(rgb[..., 1] - 165) < 45) & (rgb[..., 2] < 90)
    lab, n = ndimage.label(orange)
    pts = []
    for i in range(1, n + 1):
        ys, xs = np.where(lab == i)
        if 300 <= len(ys) <= 420:
            pts.append([round(float((xs.mean() - x0p) / sx), 3), round(float((y0p - ys.mean()) / sy), 3)])
    pts.sort()
    return pts, float(1.0 / sx)


def main():
    im = Image.open(GIF)
    titles = read_titles(im)
    markers, m_per_px = read_markers(im)
    doc = {
        "source": "images/" + os.path.basename(GIF) + " of the reference repository (written by src/closed_loop.py:439 / src/draw.py:380,407,450)",
        "setting": {"demo": "demo9", "N_free": 5, "N_fix": 5, "senseDis": 8, "Q_free": 0.5, "R_free": [0.01, 0.1], "Q_fix": 0.001,
                    "R_fix": [0.01, 1.0], "terminal_set": "[[5, 30], [x0[1] + 4, 60]]", "uU": [0.6, "pi/6"],
                    "listed_at": "src/simulation.py:68-74 (\"demo 9\")", "frames": im.n_frames},
        "spend_time": [float(t) for t in titles],
        "spend_time_text": titles,
        "markers_xy": markers,
        "metres_per_pixel": m_per_px,
    }
    with open(os.path.join(HERE, "reference_gif_demo9.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("frames", len(titles), "last", titles[-1], "markers", len(markers))
    pts, mpp = read_demo1_markers()
    doc1 = {"source": "images/OBCA_dynObs_demo1.gif of the reference repository, last frame (screen recording; code version and "
                      "settings of the run not recorded -- compared with the checked-in demo1 defaults)",
            "markers_xy": pts, "metres_per_pixel": mpp}
    with open(os.path.join(HERE, "reference_gif_demo1.json"), "w") as f:
        json.dump(doc1, f, indent=1)
    print("demo1 markers", len(pts))


if __name__ == "__main__":
    main()
