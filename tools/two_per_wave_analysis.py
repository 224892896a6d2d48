"""dev tool (round-5 review item 4a, 'two instances per wavefront'): what lock step between two instances of one wavefront costs BEFORE any
gain from better lane use, from the per-instance iteration and factorisation counts of the headline batch (structured core on the host =
the device's iterates).  Two instances in one wavefront share the program counter: the wave runs until BOTH are done and executes every
iteration's longest path (inertia-correction retries, line-search trials, second-order corrections).  Optimistic bound (only the totals
are synchronised, not the branches inside an iteration): cost = max over the pair of the KKT factorisations / iterations.
    python tools/two_per_wave_analysis.py [B]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from tests import native_build                                                                          # noqa: E402
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
b = sc.make_batch(B, 5)
o = native_build.lpi_solve(b["variant"], 5, b["m"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"])
it, nf = o["iters"].astype(float), o["info"][:, 3]
print("C2, %d instances, default start ladder: iterations mean %.2f max %d, factorisations mean %.2f max %d" % (B, it.mean(), it.max(), nf.mean(), nf.max()))
for name, v in (("iterations", it), ("factorisations", nf)):
    adj = np.maximum(v[0::2], v[1::2]).mean() / v.mean()
    srt = np.sort(v)
    best = np.maximum(srt[0::2], srt[1::2]).mean() / v.mean()
    print("  %s: pairs in submission order run %.3f x the mean per instance (an oracle that pairs equal counts: %.3f x)" % (name, adj, best))
