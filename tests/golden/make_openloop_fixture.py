"""Fixture ``reference_openloop_demo9.json`` -- positions READ OFF the second solver output the reference repository holds.

``images/aStar_vs_openLoopOBCA.png`` (also Figure 10 of the project report and the picture in README.md) is what the reference's
``simulation.run_aStar`` draws (src/simulation.py:114-123): for the demo's map the A* route and, as magenta dots, the N + 1 = 51
poses of ONE open-loop free-time solve -- ``mpc.N_free = 50; mpc.mpc_openLoop_freeTime()`` (src/closed_loop.py:113-120:
obca_mpc4 on the start/goal-only reference, all static obstacles of demo9) -- as CasADi/IPOPT returned them.  The report gives the
weights of that figure (Q = 0.5 I, R = 0.01 I for the input, 0.1 I for its rate: the settings of the GIF run as well).

This script (run HERE, where /root/reference exists; the fixture travels, the picture does not) finds the axes frame (x 0..40 m,
y 0..60 m) and the centres of the magenta dots that do not touch another mark: 47 of the 51 (the first and the last lie under
the start / goal markers).  One pixel is 0.091 m.  Data only: no reference source text is stored.

    python tests/golden/make_openloop_fixture.py       # rewrites tests/golden/reference_openloop_demo9.json
"""
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
PNG = "/root/reference/images/aStar_vs_openLoopOBCA.png"


def main():
    im = np.asarray(Image.open(PNG).convert("RGB")).astype(int)
    H, W, _ = im.shape
    dark = im.sum(2) < 120
    cols = [i for i in range(W) if dark[:, i].sum() > 0.6 * H]            # the two vertical sides of the axes frame
    rows = [i for i in range(H) if dark[i, :].sum() > 0.6 * W]            # top and bottom (the walls of the map are shorter)
    x0, x1, y1, y0 = cols[0], cols[-1], rows[0], rows[-1]                   # pixel columns of x = 0 / 40, rows of y = 60 / 0
    mag = (np.abs(im[:, :, 0] - 191) < 40) & (im[:, :, 1] < 90) & (np.abs(im[:, :, 2] - 191) < 40)      # matplotlib's 'm'
    lab, n = ndimage.label(mag)
    c = np.array(ndimage.center_of_mass(mag, lab, range(1, n + 1)))
    size = ndimage.sum(mag, lab, range(1, n + 1))
    x = (c[:, 1] - x0) / (x1 - x0) * 40.0
    y = (y0 - c[:, 0]) / (y0 - y1) * 60.0
    keep = (size >= 15) & ~((x > 24) & (y < 12))                            # whole dots; not the legend's sample
    pts = np.stack([x, y], 1)[keep]
    pts = pts[np.argsort(pts[:, 1] + 0.01 * pts[:, 0])]
    doc = {
        "source": "images/aStar_vs_openLoopOBCA.png of the reference repository (src/simulation.py:114-123 run_aStar, demo9, N_free = 50)",
        "settings": {"demo": "demo9", "N": 50, "Q_free": 0.5, "note": "weights as the project report states for this figure; start/goal-only reference, all static obstacles"},
        "pixel_m": 40.0 / (x1 - x0),
        "frame_px": {"x0": int(x0), "x40": int(x1), "y60": int(y1), "y0": int(y0)},
        "markers_total": 51, "markers_read": int(len(pts)),
        "markers_xy": [[round(float(a), 3), round(float(b), 3)] for a, b in pts],
    }
    with open(os.path.join(HERE, "reference_openloop_demo9.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(len(pts), "markers; pixel = %.4f m" % doc["pixel_m"], "frame", doc["frame_px"])


if __name__ == "__main__":
    main()
