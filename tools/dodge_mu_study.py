"""dev tool: what if the dodge rung's FIRST level started at IPOPT's mu_init 0.1 as its second level does (csrc/obca_device.h:
OBCA_DODGE_LEVEL1_MU; host build of the structured core compiled with -DOBCA_DODGE_LEVEL1_MU=0.1 into a scratch library)?  The
reference-held runs that use the rung -- demo11 (Figure 11's titles, the GIF's markers) and demo1 (Figure 12) -- replayed with it.
    python tools/dodge_mu_study.py [mu]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import native_build, reference_report  # noqa: E402

mu = sys.argv[1] if len(sys.argv) > 1 else "0.1"
out = "/tmp/libnative_host_dodgemu.so"
subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-Wno-unknown-pragmas", "-DOBCA_DODGE_LEVEL1_MU=" + mu,
                native_build.SRC, "-o", out], check=True)
for name, lib in (("product (mu 1)", None), ("first level at mu " + mu, out)):
    if lib:
        native_build._lib = None
        native_build.OUT = lib
        native_build.DEPS = []
    fx = reference_report.fixture()
    gif = reference_report.gif_demo11()
    for which, st, key in (("demo11", reference_report.demo11_setting(), "figure11_demo11"), ("demo1", reference_report.demo1_setting(), "figure12_demo1")):
        s = native_build.LpiObca()
        cum, cl = reference_report.replay(st, s, 200)
        titles = sorted(f["spend_time"] for f in fx[key]["frames"])
        hits = reference_report.match(cum, titles)
        line = "%-22s %-6s titles at steps %s, distances %s" % (name, which, [k for k, _ in hits], [round(e, 4) for _, e in hits])
        if which == "demo11":
            dots = np.asarray(gif["closed_loop_markers"])[:, :2] if isinstance(gif, dict) and "closed_loop_markers" in gif else None
            if dots is not None:
                xs = np.asarray(cl.x_closed)[:, :2]
                d = np.sqrt(((xs[:, None, :] - dots[None]) ** 2).sum(-1)).min(0)
                line += "; markers max %.3f mean %.3f m" % (d.max(), d.mean())
        print(line, flush=True)
