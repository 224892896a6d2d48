"""The two state-history plots of the reference's project report (fixtures tests/golden/reference_state_openloop_demo9.json and
reference_state_closedloop_demo9.json, read off PDF objects 47 and 48 by tests/golden/make_report_state_fixture.py): x, y and
theta against the knot index of the N = 50 open-loop plan (src/simulation.py:114-123, 138-160) and against the step of the demo9
closed loop (src/simulation.py:125-208, src/closed_loop.py:345-441).  theta is pinned by nothing else the repository holds: the
GIFs and the open-loop picture show positions only.

Two measures, both in PIXELS of the plot so that one tolerance serves the three panels:

* ``ink_distance`` -- distance from the point (column of the step, row of the value) to the nearest ink of the drawn polyline.
  Geometric, so it needs no exception on the steep flanks (theta jumps by 1 rad within one step at the corners of the plan).
* ``readout_error`` -- value read at the step's column minus the value compared, for steps where the curve crosses the column in
  at most ``FLAT_ROWS`` rows of ink (elsewhere a column read-out smears the flank: stated rule, the ink distance covers them).
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# 2 px: what a tick's snapping to whole pixels (<= 0.6 px, the fixture's `tick_fit_residual_px`), the resampling of the picture into
# the PDF and the 2 px line allow; = 0.04 rad on the closed-loop theta panel, 0.027 rad on the plan's, 0.5-0.9 m on x and y
PIXEL_TOL = 2.0
FLAT_ROWS = 3
PANEL = {"x": 0, "y": 1, "theta": 2}


def fixture(which):
    """which: "openloop" | "closedloop" """
    with open(os.path.join(HERE, "golden", "reference_state_%s_demo9.json" % which)) as f:
        return json.load(f)


def panel(fx, name):
    return fx["panels"][PANEL[name]]


def to_pixels(p, steps, values):
    c = p["column_of_step"]["per_step"] * np.asarray(steps, float) + p["column_of_step"]["at_step_0"]
    r = p["row_of_value"]["per_unit"] * np.asarray(values, float) + p["row_of_value"]["at_value_0"]
    return c, r


def ink_distance(p, steps, values):
    """pixels between each (step, value) and the nearest ink of the panel's curve (0 inside the ink)"""
    cur = np.asarray(p["curve"], float)          # column, first ink row, last ink row, centre
    cs, rs = to_pixels(p, steps, values)
    out = []
    for c, r in zip(cs, rs):
        near = cur[np.abs(cur[:, 0] - c) <= 8.0]
        dx = np.maximum(np.abs(near[:, 0] - c) - 0.5, 0.0)
        dy = np.maximum(np.maximum(near[:, 1] - 0.5 - r, r - near[:, 2] - 0.5), 0.0)
        out.append(float(np.sqrt(dx * dx + dy * dy).min()) if len(near) else np.inf)
    return np.asarray(out)


def readout(p):
    """-> (values read at the step columns, mask of the steps where the read-out is sharp)"""
    v = np.asarray([r["value"] for r in p["readout"]])
    flat = np.asarray([r["ink_rows_at_the_column"] <= FLAT_ROWS for r in p["readout"]])
    return v, flat


def readout_error_px(p, steps, values):
    """(read-out - value) in pixels at the given steps, NaN where the curve is too steep for a column read-out"""
    v, flat = readout(p)
    steps = np.asarray(steps, int)
    e = (v[steps] - np.asarray(values, float)) / p["value_per_pixel"]
    return np.where(flat[steps], e, np.nan)


def compare(fx, X, steps):
    """X: (3, n) states of this build at the given steps -> {panel: (largest ink distance, largest |read-out error| on the flat steps)} in pixels"""
    out = {}
    for name, i in PANEL.items():
        p = panel(fx, name)
        d = ink_distance(p, steps, X[i])
        e = readout_error_px(p, steps, X[i])
        out[name] = (float(d.max()), float(np.nanmax(np.abs(e))), int(np.isfinite(e).sum()))
    return out
