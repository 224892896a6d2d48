"""Fixture ``reference_report_figures.json`` -- numbers READ OFF closed-loop frames the reference repository holds in its
project report (``ME231_Team9_Project_Technical_Report.pdf``, Figures 11 and 12: four frames each of
``plotClass.fullDimension_closedLoop_animate``, titled ``'Spend Time = %.2f (sec)' % sum(Ts_opt[:k])``, src/draw.py:380 -- the
cumulative step lengths CasADi/IPOPT returned, as in the demo9 GIF).

* Figure 12: demo1 of src/demo_setting.py (corridor 39 x 10, box 10..15 x 1..5, one 3 x 3 obstacle moving up at x = 22.5 with
  0.2 m/s), the closed loop as checked in (closed_loop_mpc4: N_free = N_fix = 6, senseDis = 10).
* Figure 11: demo11 of src/demo_setting.py:248-269 (corridor 80 x 10, start (3, 4, 0), goal (77, 4, 0), two 3 x 3 obstacles: one
  moving up at x = 30.5, one moving down at x = 39.5, 0.1 m/s), the closed loop as checked in.  The same run is recorded in
  images/OBCA_dynObs_demo11.gif (tests/golden/make_gif_demo11_fixture.py reads its markers).

The titles are four-digit numbers in raster images embedded in the PDF; they are TRANSCRIBED BY EYE (listed below) and
cross-checked by this script against something it measures itself: the drawn position of the moving obstacle, which the
reference advances by its speed times the spent time (src/closed_loop.py:445-486) -- centre y = y0 + v t within a few
pixels.  Data only: no reference source text or image is stored.

    python tests/golden/make_report_fixture.py        # rewrites tests/golden/reference_report_figures.json
"""
import json
import os
import re
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PDF = "/root/reference/ME231_Team9_Project_Technical_Report.pdf"
# object numbers of the frames inside the PDF, the title of each as read by eye, and the obstacle used for the cross-check
FIG12 = dict(objs=(60, 61, 62, 63), titles=(8.77, 23.74, 31.87, 48.13), xlim=(0.0, 39.0), box_x=22.5, y0=0.0, v=0.2, thr=330)
FIG11 = dict(objs=(54, 55, 56, 57), titles=(48.29, 38.32, 64.92, 98.55), xlim=(0.0, 80.0), box_x=30.5, y0=0.0, v=0.1, thr=200)


def pdf_images():
    data = open(PDF, "rb").read()
    out = {}
    for m in re.finditer(rb"(\d+) 0 obj(.*?)endobj", data, re.S):
        body = m.group(2)
        i = body.find(b"stream")
        hdr = body[:i]
        if i < 0 or b"/Image" not in hdr or b"/DeviceRGB" not in hdr:
            continue
        w, h = int(re.search(rb"/Width (\d+)", hdr).group(1)), int(re.search(rb"/Height (\d+)", hdr).group(1))
        raw = body[i + 6:].lstrip(b"\r\n")
        raw = raw[:raw.rfind(b"endstream")]
        out[int(m.group(1))] = np.frombuffer(zlib.decompress(raw), np.uint8).reshape(h, w, 3).astype(int)
    return out


def box_centre_y(im, xlim, box_x, thr):
    """centre y (data units) of the 3 x 3 rectangle drawn around x = box_x.  Scale from the corridor walls (the two longest dark
    rows: y = 9 above, y = 1 below; they span the map in x); the rectangle's horizontal edges are the other rows that are dark
    along [box_x - 1, box_x + 1]; with one edge outside the picture the centre is 1.5 m beyond the visible one"""
    dark = im.sum(2) < thr
    order = np.argsort(-dark.sum(1))
    walls = [int(order[0])]
    for r in order[1:]:
        if abs(int(r) - walls[0]) > 20:
            walls.append(int(r))
            break
    r9, r1 = min(walls), max(walls)
    wall_cols = np.flatnonzero(dark[r9])
    x0, x1 = wall_cols[0], wall_cols[-1]
    ppm_y = (r1 - r9) / 8.0
    px = lambda x: int(round(x0 + (x - xlim[0]) / (xlim[1] - xlim[0]) * (x1 - x0)))
    seg = dark[:, px(box_x - 1.0):px(box_x + 1.0)]
    rows = [r for r in range(seg.shape[0]) if seg[r].mean() > 0.85 and min(abs(r - r9), abs(r - r1)) > 3]
    ys = np.array([1.0 + (r1 - r) / ppm_y for r in rows])
    ys = ys[(ys > -0.5) & (ys < 10.5)]
    if len(ys) == 0:
        return None
    span = ys.max() - ys.min()
    if 2.7 < span < 3.3:
        return 0.5 * (ys.min() + ys.max())               # both horizontal edges seen, 3 m apart
    e = float(np.mean(ys))
    if span < 0.2 and e - 3.0 < 1.0:                      # one edge only and the other would lie below the wall y = 1: the top edge
        return e - 1.5
    if span < 0.2 and e + 3.0 > 9.0:                      # ... above the wall y = 9: the bottom edge
        return e + 1.5
    return None                                          # other marks (the car, the plan) cross the strip: no measurement


def main():
    ims = pdf_images()
    doc = {"source": "ME231_Team9_Project_Technical_Report.pdf of the reference repository, Figures 11 and 12 (frames of its closed loop)",
           "titles": "transcribed by eye; cross-check: drawn position of the moving obstacle = y0 + v * title"}
    for name, F in (("figure12_demo1", FIG12), ("figure11_demo11", FIG11)):
        frames = []
        for o, t in zip(F["objs"], F["titles"]):
            yc = box_centre_y(ims[o], F["xlim"], F["box_x"], F["thr"])
            frames.append({"pdf_object": o, "spend_time": t, "moving_box_centre_y_measured": None if yc is None else round(float(yc), 2),
                           "moving_box_centre_y_from_title": round(F["y0"] + F["v"] * t, 2)})
            print(name, o, t, "box centre y measured", frames[-1]["moving_box_centre_y_measured"], "from the title", frames[-1]["moving_box_centre_y_from_title"])
        doc[name] = {"frames": frames}
    doc["figure12_demo1"]["setting"] = "problemSetting('demo1'), closedLoop defaults as checked in (N_free = N_fix = 6, senseDis = 10), no stop at k = 30"
    doc["figure11_demo11"]["setting"] = "problemSetting('demo11') (src/demo_setting.py:248-269), closedLoop defaults as checked in (N_free = N_fix = 6, senseDis = 10), no stop at k = 30"
    with open(os.path.join(HERE, "reference_report_figures.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
