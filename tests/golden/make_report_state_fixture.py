"""Fixtures ``reference_state_openloop_demo9.json`` and ``reference_state_closedloop_demo9.json`` -- numbers READ OFF the two
state-history plots the reference repository holds in its project report (``ME231_Team9_Project_Technical_Report.pdf``):

* PDF object 47: x, y, theta against the knot index of the N = 50 open-loop free-time plan of demo9 -- the plan
  ``simulation.run_aStar`` draws (src/simulation.py:114-123, ``mpc_openLoop_freeTime`` src/closed_loop.py:113-120); the picture
  images/aStar_vs_openLoopOBCA.png shows the same 51 poses as dots (tests/golden/make_openloop_fixture.py), without theta.
* PDF object 48, "ClosedLoop OBCA State (horizon N = 5)": x, y, theta against the step of the demo9 closed loop -- the run of the
  GIF (src/closed_loop.py:345-441, src/simulation.py:125-208; tests/golden/make_gif_fixture.py), 0 .. 85.

Both are matplotlib line plots (three stacked axes, a black polyline through one vertex per step) embedded as raster images.
What the script measures: the axes frames, the tick marks (centre of darkness, sub-pixel), the curve's ink per pixel column.
What is transcribed by eye: the tick LABELS (listed below).  The pixel -> value maps are least-squares lines through the ticks
(residuals asserted <= 0.8 px: matplotlib snaps a tick to a whole pixel before the picture was resampled into the PDF).  Stored per panel: the map, the curve's ink per column (first row, last row, darkness-weighted
centre) and -- derived from those -- one read-out per step with a flag where the curve is too steep for a column read-out.
Data only: no reference source text or image is stored.

    python tests/golden/make_report_state_fixture.py
"""
import json
import os

import numpy as np

from make_report_fixture import pdf_images

HERE = os.path.dirname(os.path.abspath(__file__))
# tick labels as printed, top to bottom for the value axes, left to right for the step axis
PLOTS = {
    "reference_state_openloop_demo9.json": dict(
        pdf_object=47, what="x, y, theta against the knot index of the open-loop free-time plan of demo9 at N = 50 (src/simulation.py:114-123)",
        step_ticks=[0, 10, 20, 30, 40, 50],
        panels=[("x", "m", [30, 20, 10, 0]), ("y", "m", [60, 50, 40, 30, 20, 10]), ("theta", "rad", [1.5, 1.0, 0.5, 0.0])]),
    "reference_state_closedloop_demo9.json": dict(
        pdf_object=48, what="x, y, theta against the step of the demo9 closed loop the GIF records (src/closed_loop.py:345-441)",
        step_ticks=[0, 20, 40, 60, 80],
        panels=[("x", "m", [30, 20, 10, 0]), ("y", "m", [60, 50, 40, 30, 20, 10]), ("theta", "rad", [2.5, 2.0, 1.5, 1.0, 0.5, 0.0])]),
}


def clusters(idx):
    """runs of consecutive integers -> list of arrays"""
    idx = np.asarray(idx)
    if len(idx) == 0:
        return []
    cut = np.flatnonzero(np.diff(idx) > 1) + 1
    return np.split(idx, cut)


def frames(dk):
    """axes frames from the long dark lines: (left, right) columns and [(top, bottom)] rows per panel, as float centres.  A frame
    line spans the whole axes (> 70 % of the picture's width / > 20 % of its height) and is ~1.5 px thick"""
    H, W = dk.shape
    col_lines = [float(np.average(c, weights=dk[:, c].sum(0))) for c in clusters(np.flatnonzero((dk > 0.35).sum(0) > 0.2 * H))]
    left, right = col_lines[0], col_lines[-1]
    row_lines = [float(np.average(r, weights=dk[r, :].sum(1))) for r in clusters(np.flatnonzero((dk > 0.35).sum(1) > 0.7 * W))]
    return left, right, row_lines


def tick_centres(profile, lo):
    """centres of darkness of the tick marks along a 1-D darkness profile"""
    out = []
    for c in clusters(np.flatnonzero(profile > 0.25)):
        out.append(lo + float(np.average(c, weights=profile[c])))
    return out


def fit(px, val):
    a = np.vstack([np.asarray(val, float), np.ones(len(val))]).T
    (k, b), *_ = np.linalg.lstsq(a, np.asarray(px, float), rcond=None)
    res = np.abs(a @ np.array([k, b]) - px).max()
    assert res <= 0.8, ("tick residual", res, px, val)
    return float(k), float(b), float(res)          # pixel = k * value + b


def digitise(im, spec):
    dk = (255.0 - im.mean(2)) / 255.0
    H, W = dk.shape
    left, right, rows = frames(dk)
    if len(rows) == 5:                              # the top frame of the first panel is cropped off the picture (object 47 keeps it at row 1)
        rows = [0.0] + rows
    assert len(rows) == 6, rows
    li, ri = int(round(left)), int(round(right))
    out = []
    for p, (name, unit, labels) in enumerate(spec["panels"]):
        top, bot = rows[2 * p], rows[2 * p + 1]
        ti, bi = int(round(top)), int(round(bot))
        # step ticks hang below the bottom frame, value ticks stick out to the left of the left frame (3 px long)
        sx = tick_centres(dk[bi + 2:bi + 4, :].mean(0), 0)
        sx = [c for c in sx if left - 1 <= c <= right + 1]
        assert len(sx) == len(spec["step_ticks"]), (name, sx)
        kx, bx, rx = fit(sx, spec["step_ticks"])
        vy = tick_centres(dk[ti - 1:bi + 2, li - 3:li - 1].mean(1), ti - 1)
        assert len(vy) == len(labels), (name, vy)
        ky, by, ry = fit(vy, labels)
        # the curve: ink strictly inside the frame
        c0, c1, r0, r1 = li + 2, ri - 1, ti + 2, bi - 1
        cols = []
        for c in range(c0, c1):
            col = dk[r0:r1, c]
            ink = np.flatnonzero(col > 0.5)
            if len(ink) == 0:
                continue
            w = np.where(col > 0.1, col, 0.0)
            # only the connected stretch of darkness around the ink (tick labels never reach inside, but be safe)
            cols.append([c, int(r0 + ink[0]), int(r0 + ink[-1]), round(float(r0 + np.average(np.arange(len(col)), weights=w)), 3)])
        cols = np.asarray(cols)
        n_steps = int(round((cols[-1, 0] - bx) / kx)) + 1
        read = []
        for k in range(n_steps):
            cx = kx * k + bx
            j = np.searchsorted(cols[:, 0], cx)
            j = min(max(j, 1), len(cols) - 1)
            ca, cb = cols[j - 1], cols[j]
            t = min(max((cx - ca[0]) / max(cb[0] - ca[0], 1e-9), 0.0), 1.0)
            row = (1 - t) * ca[3] + t * cb[3]
            span = max(ca[2] - ca[1], cb[2] - cb[1]) + 1
            read.append({"step": k, "value": round((row - by) / ky, 4), "ink_rows_at_the_column": int(span)})
        out.append({"name": name, "unit": unit, "tick_labels": labels, "tick_rows": [round(v, 2) for v in vy], "step_ticks": spec["step_ticks"],
                    "step_tick_columns": [round(v, 2) for v in sx],
                    "column_of_step": {"per_step": round(kx, 5), "at_step_0": round(bx, 3)},
                    "row_of_value": {"per_unit": round(ky, 5), "at_value_0": round(by, 3)},
                    "value_per_pixel": round(abs(1.0 / ky), 5), "tick_fit_residual_px": [round(rx, 3), round(ry, 3)],
                    "curve_columns": "[column, first ink row, last ink row, darkness-weighted centre row] for every pixel column the curve crosses",
                    "curve": [[int(c[0]), int(c[1]), int(c[2]), float(c[3])] for c in cols],
                    "readout": read})
        print(spec["pdf_object"], name, "steps", n_steps, "px/step %.3f" % kx, "%s/px %.4f" % (unit, abs(1 / ky)), "tick residuals", round(rx, 2), round(ry, 2))
    return out


def main():
    ims = pdf_images()
    for fname, spec in PLOTS.items():
        panels = digitise(ims[spec["pdf_object"]], spec)
        doc = {"source": "ME231_Team9_Project_Technical_Report.pdf of the reference repository, PDF object %d" % spec["pdf_object"],
               "what": spec["what"],
               "method": "frames, tick marks and the curve's ink measured by tests/golden/make_report_state_fixture.py; tick labels transcribed by eye; "
                         "pixel = per_unit * value + offset, least squares through the ticks",
               "steps": len(panels[0]["readout"]), "panels": panels}
        assert len({len(p["readout"]) for p in panels}) == 1
        with open(os.path.join(HERE, fname), "w") as f:
            json.dump(doc, f, separators=(",", ":"))
            f.write("\n")


if __name__ == "__main__":
    main()
