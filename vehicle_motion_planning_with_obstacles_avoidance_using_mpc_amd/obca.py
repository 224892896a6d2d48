"""Drop-in for the reference module ``obca`` (class ``obca``; reference src/obca.py).

Same positional signatures and return tuples as the reference's live variants
    obca_mpc4  (src/obca.py:828)   free time
    obca_mpc6  (src/obca.py:1361)  fixed time + terminal set
    obca_mpc8  (src/obca.py:1564)  fixed time, no terminal set
so a driver written for the reference (`self.obca_solver = obca()`, src/closed_loop.py:22) runs unchanged.
Each call is a batch of one on the MI355X kernel; like the reference it never raises on solver failure
and returns the last iterate with ``feas=False``.
"""
import numpy as np
import torch

from .solver import BatchSolver, SolverParams, pack_reference_call


class obca:
    def __init__(self):
        self._solvers = {}
        # The start ladder of every solve (include/obca_mpc.h: start_order): "default" -- reference window -> x0 -> zeros for every variant
        # (x0 first for single-start calls); "x0" / "window" / "zeros": that start first for every variant ("zeros" = the reference's
        # literal all-zero start, src/obca.py:856).
        self.start_order = "default"
        # True: every call runs the first start of the order only.  A driver that answers a failed obca_mpc6 with obca_mpc8
        # itself (this package's closedLoop) asks for that per call instead: obca_mpc6(..., single_start=True).
        self.single_start = False
        # include/obca_mpc.h: the ladder's last rung for obca_mpc6 / 8 (the window moved to either side, the better answer stays);
        # the closed-form terminal-set screen of obca_mpc6 (a call that cannot succeed is answered feas=False without a solve)
        self.dodge = True
        self.terminal_screen = True

    def _solver(self, N, m):
        key = (int(N), tuple(m))
        if key not in self._solvers:
            s = BatchSolver(N, m, max_batch=1)
            for mode in ("multiwave", "global"):      # a batch of one leaves the GPU idle: give the instance a whole CU (four
                try:                                  # wavefronts; LDS resident, or with the rows in the HBM workspace where the shape
                    s.set_mode(mode)                  # is beyond the LDS), which shortens the call; same words as one wavefront
                    break
                except RuntimeError:
                    pass                              # neither holds the shape: auto mode (the lane kernel)
            self._solvers[key] = s
        return self._solvers[key]

    def _run(self, variant, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0,
             terminal_set=None, single_start=False, start_order=None):
        m, x0v, u0v, xr, A, b, Tsv, term = pack_reference_call(variant, Ts, N, x0, xref, nObs, vObs, AObs, bObs, u0,
                                                                terminal_set)
        kw = dict(xL=xL, xU=xU, uL=uL, uU=uU, ego=ego, dmin=dmin)
        kw.update(start_order=self.start_order if start_order is None else start_order, single_start=bool(single_start or self.single_start),
                  dodge=self.dodge, terminal_screen=self.terminal_screen)
        if variant == 4:
            prm = SolverParams(Q_free=Q, R_free=R, P_free=P, **kw)
        else:
            prm = SolverParams(Q_fix=Q, R_fix=R, P_fix=P, **kw)
        s = self._solver(N, m)
        out = s.solve(variant, x0v[None], u0v[None], xr[None], A[None], b[None], np.array([Tsv]), term[None], prm)
        torch.cuda.synchronize()
        feas = bool(out.feas[0].item())
        self.last = dict(status=int(out.status[0]), iters=int(out.iters[0]), f=float(out.info[0, 0]), elastic=float(out.info[0, 1]), E0=float(out.info[0, 2]), nfact=int(out.info[0, 3]))
        x_Opt = out.xopt[0].cpu().numpy()
        u_Opt = out.uopt[0].cpu().numpy()
        return x_Opt, u_Opt, feas, float(out.ts_opt[0].item())

    def obca_mpc4(self, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0, start_order=None):
        """start_order (not a reference argument): the order of the ladder's starts for THIS call -- for a caller whose reference is
        no trajectory (the open-loop plan's start-and-goal reference: "x0"); None = the object's ``start_order``"""
        return self._run(4, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0, start_order=start_order)

    def obca_mpc6(self, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0, uOpt,
                  terminal_set, single_start=False):
        """single_start (not a reference argument): the first start of the ladder only -- for a caller whose own fallback
        (obca_mpc8, src/closed_loop.py:393-398) follows a failure"""
        return self._run(6, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0,
                         terminal_set, single_start=single_start)

    def obca_mpc8(self, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0, uOpt):
        return self._run(8, Ts, P, Q, R, N, x0, xL, xU, uL, uU, xref, nObs, vObs, AObs, bObs, dmin, ego, u0)
