"""dev tool (round 6): the reference-held closed-loop runs replayed with the INDEPENDENT IPOPT-style oracle (oracle/ipopt_like.py: hard
equalities, slack bounds, restoration phase) from the reference's OWN start -- every variable 0, Topt = 1 (src/obca.py:856) -- one
solve per step, no ladder, no elastic form: which steps of the reference's runs does IPOPT's published algorithm reproduce from
the reference's start?      python tools/ipopt_like_study.py demo9|demo1|demo11 [steps] [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ipopt_like  # noqa: E402
from oracle.obca_nlp import Problem  # noqa: E402


class IpoptLikeObca:
    """`obca`-shaped object (src/obca.py:828, 1361, 1564) on oracle/ipopt_like.py"""

    def __init__(self):
        self.calls = []

    def _run(self, variant, a, term=None):
        p = Problem.from_reference_args(variant, *a[:18], terminal_set=term)
        t0 = time.time()
        r = ipopt_like.solve(p)
        self.calls.append(dict(variant=variant, status=r.status, name=r.status_name, iters=r.iters, restorations=r.restorations,
                               f=r.f, Ts_opt=r.Ts_opt, seconds=time.time() - t0))
        print("  step %3d obca_mpc%d %-28s it %4d resto %2d f %.5f Ts_opt %.4f (%.1f s)" % (len(self.calls), variant, r.status_name, r.iters, r.restorations, r.f, r.Ts_opt, time.time() - t0), flush=True)
        return r.xopt, r.uopt, bool(r.feas), float(r.Ts_opt)

    def obca_mpc4(self, *a):
        return self._run(4, a)

    def obca_mpc6(self, *a, single_start=False):
        return self._run(6, a, term=a[19])

    def obca_mpc8(self, *a):
        return self._run(8, a)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "demo9"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    out = sys.argv[3] if len(sys.argv) > 3 else None
    s = IpoptLikeObca()
    if which == "demo9":
        from tests import reference_gif
        cum, xs, cl = reference_gif.replay(s, n)
        ref = np.asarray(reference_gif.fixture()["spend_time"][1:])
        m = min(len(cum), len(ref))
        err = np.abs(cum[:m] - ref[:m])
        first_off = int(np.argmax(err > reference_gif.TIME_TOL)) + 1 if np.any(err > reference_gif.TIME_TOL) else None
        res = dict(run="demo9 GIF", steps=len(cum), consecutive_steps_matched=(first_off - 1) if first_off else m, title_errors=[round(float(e), 4) for e in err],
                   steps_within_tolerance=int((err <= reference_gif.TIME_TOL).sum()), Ts_opt=[float(v) for v in cl.T_closed], x_closed=np.asarray(xs).tolist())
    else:
        from tests import reference_report
        st = reference_report.demo1_setting() if which == "demo1" else reference_report.demo11_setting()
        cum, cl = reference_report.replay(st, s, n)
        fx = reference_report.fixture()["figure12_demo1" if which == "demo1" else "figure11_demo11"]
        titles = sorted(f["spend_time"] for f in fx["frames"])
        hits = reference_report.match(cum, titles)
        res = dict(run=which, steps=len(cum), titles=titles, nearest_step=[k for k, _ in hits], distance_s=[round(e, 4) for _, e in hits],
                   Ts_opt=[float(v) for v in cl.T_closed], x_closed=np.asarray(cl.x_closed).tolist())
    res["calls"] = s.calls
    print(json.dumps({k: v for k, v in res.items() if k not in ("calls", "x_closed", "Ts_opt", "title_errors")}))
    if out:
        with open(out, "w") as f:
            json.dump(res, f)


if __name__ == "__main__":
    main()
