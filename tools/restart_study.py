"""Measurements behind the restart phase (oracle/ipm_dense.py:solve, csrc/obca_device.h: OBCA_RESTART_*), CPU only.

  python tools/restart_study.py resto        IPOPT's restoration problem started AT the infeasible stationary point of the
                                             golden scenario demo1_dyn_mpc6 (numpy spec): does it move?
  python tools/restart_study.py c3 [n]       n C3-gated instances at N = 20 (structured core on the host): share converged
                                             without / with the restart phase, iteration counts of the restart passes
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, ipm_dense  # noqa: E402


def resto():
    from tests.test_oracle_nlp import build
    case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "nlp_eval.json"))) if c["name"] == "demo1_dyn_mpc6"][0]
    p = build(case)
    r = ipm_dense.solve(p, dict(no_restart=True))
    print("cold start: status %d, largest elastic variable %.3e, f %.4f" % (r.status, r.elastic, r.f))
    xR = r.x.copy()

    class Resto:                      # min rho ||c||_1 + zeta/2 ||D_R (x - x_R)||^2, Waechter & Biegler (2006) section 3.3
        def __init__(self, zeta):
            self.zeta, self.D2 = zeta, np.minimum(1.0, 1.0 / np.maximum(np.abs(xR), 1e-300)) ** 2

        def __getattr__(self, k):
            return getattr(p, k)

        def start_point(self):
            return xR.copy()

        def objective(self, x, grad=False, hess=False):
            d = x - xR
            f = 0.5 * self.zeta * np.sum(self.D2 * d * d)
            if grad:
                return f, self.zeta * self.D2 * d
            if hess:
                return f, np.diag(self.zeta * self.D2)
            return f
    for zeta, rho in ((np.sqrt(0.1), 1e3), (1e-3, 1e3), (np.sqrt(0.1), 1e5)):
        rr = ipm_dense._solve_once(Resto(zeta), dict(rho=rho))
        print("restoration problem from x_R (zeta %.3g, rho %g): status %d after %d iterations, largest elastic variable %.3e, "
              "moved %.1e" % (zeta, rho, rr.status, rr.iters, rr.elastic, np.max(np.abs(rr.x - xR))))
    r2 = ipm_dense.solve(p)
    print("restart phase (reference window, mu0 = %g): status %d, f %.6f (SURVEY Appendix C witness: 0.029735)" % (ipm_dense.RESTART_MU, r2.status, r2.f))


def c3(n):
    from tests import native_build
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    N = 20
    b = sc.make_batch_c3(n, N, gated=True, procs=6)
    args = (b["variant"], N, b["m"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"])
    cold = native_build.lpi_solve(*args, params=c_oracle.default_params(restart=-1))
    full = native_build.lpi_solve(*args)
    bad = ~np.isin(cold["status"], (0, 1))
    ri = full["iters"][bad] - cold["iters"][bad]
    ok = np.isin(full["status"][bad], (0, 1))
    print("C3 gated, N = 20, %d instances: converged without restart %.2f %%, with %.2f %%" %
          (n, 100 * (1 - bad.mean()), 100 * np.isin(full["status"], (0, 1)).mean()))
    print("first passes that fail: iterations (quantiles 0 / 50 / 90 / 99 / 100 %%) %s" % np.quantile(cold["iters"][bad], [0, .5, .9, .99, 1]))
    print("restart passes: %d, recovered %d; iterations of the recovered ones (0 / 50 / 90 / 99 / 100 %%) %s" %
          (bad.sum(), ok.sum(), np.quantile(ri[ok], [0, .5, .9, .99, 1])))
    print("mean iterations per instance: %.1f without, %.1f with the restart phase" % (cold["iters"].mean(), full["iters"].mean()))


if __name__ == "__main__":
    if sys.argv[1] == "resto":
        resto()
    else:
        c3(int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
