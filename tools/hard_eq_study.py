"""dev tool (CPU only): the initial-state and dynamics rows as HARD equalities in the numpy spec
(oracle/ipm_dense.py: HARD_KINDS) on the replay of the reference's GIF run (tests/reference_gif.py), from the literal all-zero start
or the x0 start, one start per solve + the penalty escalation.  python tools/hard_eq_study.py rot,init,dyn zeros|x0 [steps]"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ipm_dense
from oracle.obca_nlp import Problem
from tests import reference_gif

hard = tuple(sys.argv[1].split(',')) if len(sys.argv) > 1 else ('rot', 'init', 'dyn')
start = sys.argv[2] if len(sys.argv) > 2 else 'zeros'
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 83
ipm_dense.HARD_KINDS = hard

class NumpyObca:
    def __init__(self):
        self.calls = []
    def _solve(self, variant, *a):
        p = Problem.from_reference_args(variant, *a)
        t = time.time()
        xs = None if start == 'zeros' else ipm_dense.x0_start(p)
        r = ipm_dense._solve_once(p, None, None, x_start=xs)
        if r.status == ipm_dense.STATUS_INFEASIBLE and variant == 4:
            r2 = ipm_dense._solve_once(p, dict(rho=ipm_dense.DEFAULTS['rho'] * 100), None, x_start=xs)
            r2.iters += r.iters; r = r2
        self.calls.append(dict(variant=variant, status=r.status, iters=r.iters, f=r.f, t=time.time() - t))
        print("  step %2d mpc%d status %d iters %4d f %.4f Ts %.4f (%.1fs)" % (len(self.calls), variant, r.status, r.iters, r.f, r.Ts_opt, time.time() - t), flush=True)
        return r.xopt, r.uopt, bool(r.feas), float(r.Ts_opt)
    def obca_mpc4(self, *a): return self._solve(4, *a)
    def obca_mpc6(self, *a, single_start=False): return self._solve(6, *a)
    def obca_mpc8(self, *a): return self._solve(8, *a)

ref = np.asarray(reference_gif.fixture()["spend_time"][1:])
s = NumpyObca()
cum, xs, cl = reference_gif.replay(s, nsteps)
k = min(len(cum), 83)
bad = np.where(np.abs(cum[:k] - ref[:k]) > reference_gif.TIME_TOL)[0]
print("hard=%s start=%s: %d consecutive steps matched of %d run; mean iters %.1f; statuses %s" % (hard, start, int(bad[0]) if len(bad) else k, len(cum), np.mean([c['iters'] for c in s.calls]), sorted(set(c['status'] for c in s.calls))))
