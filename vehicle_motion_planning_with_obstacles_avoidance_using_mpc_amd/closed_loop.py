"""Receding-horizon MPC driver -- host-side mirror of the reference's ``closed_loop.closedLoop``
(reference src/closed_loop.py:16-629) plus a batched driver that advances many rollouts with one GPU solve
per step.

``closedLoop(problem_setting)`` keeps the reference's attribute and method names (``closed_loop_mpc4``,
``mpc_openLoop_freeTime``, ``mpc_openLoop_fixTime``, ``update_obstacle``, ``sensor``,
``update_obstacle_constraint``, ``update_reference_trajectory``, ``update_path``) and the behaviour the survey
lists as quirks q7-q10 (``self.Ts`` overwritten after a fixed-time step, unfiltered dynamic vertex lists, cold
start every solve, stop at k = 30).  New: ``step()`` -- one iteration of the reference's ``while`` body
(src/closed_loop.py:345-432) -- split into ``prepare_step`` / ``finish_step`` so that ``BatchClosedLoop`` can put
the solves of B rollouts into one ``obca_solve_batch`` launch.  Plotting (src/draw.py) is out of scope; the
arrays the reference hands to its plot routine are kept on the object (``x_openLoop``, ``u_closed``,
``Ts_opt`` list, ``dyn_loc``).
"""
import numpy as np

from .a_star import a_star
from .model_obstacle import obstacleModel
from .solver import pack_reference_call


class closedLoop:
    def __init__(self, problem_setting, solver=None):
        self.setting = problem_setting
        self.obs_model = obstacleModel()
        if solver is None:
            from .obca import obca
            solver = obca()
        self.obca_solver = solver
        # this driver answers a failed obca_mpc6 with obca_mpc8 itself (step(), mpc_openLoop_fixTime), so it asks obca_mpc6 --
        # where the solver object offers the keyword -- for the first start of the ladder only (include/obca_mpc.h:
        # single_start); the device-resident rollouts do the same (csrc/obca_rollout.hip).  Per call: the injected solver
        # object is not changed, a bare obca_mpc6 call on it still runs the whole ladder.
        import inspect
        try:
            has_kw = "single_start" in inspect.signature(solver.obca_mpc6).parameters
        except (AttributeError, TypeError, ValueError):
            has_kw = False
        self._mpc6_kw = {"single_start": True} if has_kw else {}
        # likewise per call: the open-loop plan's reference is start and goal only (mpc_openLoop_freeTime), so that call asks -- where
        # the solver object offers the keyword -- for the x0 start first
        try:
            has_so = "start_order" in inspect.signature(solver.obca_mpc4).parameters
        except (AttributeError, TypeError, ValueError):
            has_so = False
        self._open_loop_kw = {"start_order": "x0"} if has_so else {}
        st = self.setting
        self.path_solver = a_star(st.org_gridMap, (st.startPose[1], st.startPose[0]), (st.goalPose[1], st.goalPose[0]))
        # constants of src/closed_loop.py:32-101
        self.Ts = 0.1
        self.nx, self.nu = 3, 2
        self.xL = [st.xL[0], st.xL[1], -np.pi]
        self.xU = [st.xU[0], st.xU[1], np.pi]
        self.uL = [-0.6, -np.pi / 6]
        self.uU = [0.6, np.pi / 6]
        self.x0 = st.startPose
        self.xF = st.goalPose
        self.u0 = [0, 0]
        self.fixtime = 0
        self.nObs, self.vObs, self.AObs, self.bObs = 0, np.ones(0, dtype=int), [], []
        self.xref, self.uref = [], []
        self.ego = [1.7, 0.75, 1.7, 0.75]
        self.dmin = 0.05
        self.xOpt, self.uOpt, self.feas = [], [], False
        self.Ts_opt = self.Ts
        self.Q_free = 0.1 * np.eye(3)
        self.R_free = [0.01 * np.eye(2), 0.1 * np.eye(2)]
        self.P_free = self.Q_free
        self.N_free = 6
        self.Q_fix = 0.001 * np.eye(3)
        self.R_fix = [0.01 * np.eye(2), 1 * np.eye(2)]
        self.P_fix = self.Q_fix
        self.N_fix = 6
        self.terminal_set = []
        self.dyn_orignal_info = st.dyn_obs_info          # same list objects: advanced in place (reference :109)
        self.dyn_loc = []
        # closed-loop bookkeeping
        self.k = 0
        self.a_star_ref = None
        self.x_closed, self.u_closed, self.T_closed, self.x_openLoop = [], [], [], []
        self.done = False

    # ------------------------------------------------------------------ open loop (src/closed_loop.py:113-140)
    def mpc_openLoop_freeTime(self):
        self.update_obstacle_constraint(self.N_free, self.Ts, 0)
        self.xref = self.update_path(self.N_free, self.x0, self.xF, allAviable=0, type="startGoal_only")
        # the reference of this call is start and goal only -- no trajectory a solve could start from (the default order starts at
        # the reference window): where the solver object offers the keyword, x0 goes first for THIS call unless the object was
        # told otherwise (the same optimum either way; from the straight line through the obstacles it takes three times the
        # iterations).  The solver object itself is not touched (re-entrant); one without the keyword runs its own default.
        kw = dict(self._open_loop_kw)
        if kw and getattr(self.obca_solver, "start_order", "default") != "default":
            kw = {}
        self.xOpt, self.uOpt, self.feas, self.Ts_opt = self.obca_solver.obca_mpc4(
            self.Ts, self.P_free, self.Q_free, self.R_free, self.N_free, self.x0, self.xL, self.xU, self.uL, self.uU,
            self.xref, self.nObs, self.vObs, self.AObs, self.bObs, self.dmin, self.ego, self.u0, **kw)

    def mpc_openLoop_fixTime(self):
        self.xref = self.xOpt
        self.xref = self.update_path(0, 0, 0, allAviable=1, type="")
        self.update_obstacle_constraint(self.N_fix, self.Ts_opt, 1)
        self.terminal_set = self.setting.terminal_set
        self.fixtime = 1
        args = (self.Ts, self.P_fix, self.Q_fix, self.R_fix, self.N_fix, self.x0, self.xL, self.xU, self.uL, self.uU,
                self.xref, self.nObs, self.vObs, self.AObs, self.bObs, self.dmin, self.ego, self.u0, self.uOpt)
        self.xOpt, self.uOpt, self.feas, self.Ts_opt = self.obca_solver.obca_mpc6(*args, self.terminal_set, **self._mpc6_kw)
        if self.feas == False:  # noqa: E712  (same test as the reference)
            self.xOpt, self.uOpt, self.feas, self.Ts_opt = self.obca_solver.obca_mpc8(*args)

    # ------------------------------------------------------------------ one receding-horizon step
    def goal_reached(self):
        g = self.setting.goalPose
        return not ((self.x0[0] - g[0]) ** 2 + (self.x0[1] - g[1]) ** 2 >= 0.1)

    def prepare_step(self):
        """Everything of the reference's loop body before the solve (:349-378).  Returns the pending call
        ``(variant, args)`` with args in the reference's positional order."""
        if self.a_star_ref is None:
            self.a_star_ref = self.update_path(0, self.x0, self.xF, 0, "A_star")
            self.x_closed.append(self.x0)
        k = self.k
        self.update_obstacle(k, self.Ts_opt)
        self.sensor()
        if k == 0 or self.fixtime == 0:
            self.update_obstacle_constraint(self.N_free, self.Ts, 0)
            self.xref = self.update_reference_trajectory(self.N_free, self.a_star_ref, self.x0)
            return 4, (self.Ts, self.P_free, self.Q_free, self.R_free, self.N_free, self.x0, self.xL, self.xU, self.uL,
                       self.uU, self.xref, self.nObs, self.vObs, self.AObs, self.bObs, self.dmin, self.ego, self.u0)
        self.xref = self.update_reference_trajectory(self.N_fix, self.a_star_ref, self.x0)
        for i in range(self.N_fix - 5):
            self.xref[:, i] = self.xOpt[:, i + 1]
        self.xref = self.update_path(0, 0, 0, allAviable=1, type="")
        self.terminal_set = np.array([[self.x0[0] + 5, 99], [1, 9]])
        self.update_obstacle_constraint(self.N_fix, self.Ts_opt, 1)
        return 6, (self.Ts, self.P_fix, self.Q_fix, self.R_fix, self.N_fix, self.x0, self.xL, self.xU, self.uL, self.uU,
                   self.xref, self.nObs, self.vObs, self.AObs, self.bObs, self.dmin, self.ego, self.u0, self.uOpt,
                   self.terminal_set)

    def finish_step(self, result):
        """State advance after the solve (:400-432).  Returns False when the rollout stops."""
        self.xOpt, self.uOpt, self.feas, self.Ts_opt = result
        if self.feas != True:  # noqa: E712
            self.done = True
            return False
        self.u0 = self.uOpt[:, 0].T
        self.x0 = self.xOpt[:, 1].T
        self.x_closed.append(self.x0)
        self.u_closed.append(self.u0)
        self.T_closed.append(self.Ts_opt)
        self.x_openLoop.append(self.xOpt.T)
        self.k += 1
        if self.k == 30 or self.goal_reached():
            self.done = True
            return False
        return True

    def step(self):
        """One iteration of the reference's ``while`` body, solver included."""
        variant, args = self.prepare_step()
        if variant == 4:
            res = self.obca_solver.obca_mpc4(*args)
        else:
            res = self.obca_solver.obca_mpc6(*args, **self._mpc6_kw)
            if res[2] == False:  # noqa: E712
                res = self.obca_solver.obca_mpc8(*args[:-1])
        return self.finish_step(res)

    def closed_loop_mpc4(self):
        if not self.goal_reached():
            while self.step():
                pass
        self.xOpt = np.asarray(self.x_closed).T
        self.xref = self.a_star_ref
        self.Ts_opt = self.T_closed
        return self.x_openLoop, self.x_closed, self.u_closed, self.T_closed

    # ------------------------------------------------------------------ helpers (same names as the reference)
    def update_obstacle(self, k, Ts_opt):
        """src/closed_loop.py:445-486: obstacles appear at k == t_start, afterwards advance by Ts_opt*v"""
        st = self.setting
        st.dyn_obs_info = []
        verts = []
        for info in self.dyn_orignal_info:
            if k < info[9]:
                continue
            if k > info[9]:
                info[0] += Ts_opt * info[5] * np.cos(info[2])
                info[1] += Ts_opt * info[5] * np.sin(info[2])
            st.dyn_obs_info.append(info)
            verts.append(st.get_obstacle(info[0], info[1], info[2], info[3], info[4]))
        self.dyn_loc.append(verts)
        st.add_dynamic_obstacle(st.dyn_obs_info)

    def update_obstacle_constraint(self, N, Ts, dynobs_exist):
        """src/closed_loop.py:488-500"""
        st = self.setting
        st.rebuild_lObs(N, Ts, dynObs_exist=dynobs_exist)
        self.lObs, self.nObs, self.vObs = st.lObs, st.nObs, st.vObs
        full_v = np.array([len(p) for p in self.lObs], dtype=int)
        self.AObs, self.bObs = self.obs_model.obstacle_H_Represent(len(self.lObs), full_v, self.lObs)

    def update_reference_trajectory(self, N, ref_trajectory, current_state):
        """src/closed_loop.py:502-528: window of N+1 points from the closest path point, clamped at the end"""
        ref = np.asarray(ref_trajectory)
        P = ref.shape[1]
        d = (current_state[0] - ref[0]) ** 2 + (current_state[1] - ref[1]) ** 2
        i0 = int(np.argmin(d)) if d.min() < 100000 else 0          # first strict minimum below the initial 1e5
        idx = np.minimum(i0 + np.arange(N + 1), P - 1)
        return np.array(ref[:np.size(current_state, 0)][:, idx], dtype=float)

    def update_path(self, N, x0, xF, allAviable, type):
        """src/closed_loop.py:530-589"""
        if allAviable == 0:
            ref_x = np.zeros((self.nx, N + 1))
            if type == "startGoal_only":
                ref_x[:, 0] = x0[:3]
                ref_x[:, 1:] = np.asarray(xF[:3], float)[:, None]
            elif type == "startGoal_smooth":
                # src/closed_loop.py:545-553: N + 1 points on the straight line start -> goal, heading of every point = direction
                # to its successor (last heading repeated)
                steps = np.arange(N + 1)
                ref_x[0] = ((xF[0] - x0[0]) / N) * steps + x0[0]
                ref_x[1] = ((xF[1] - x0[1]) / N) * steps + x0[1]
                ref_x[2, :N] = np.arctan2(np.diff(ref_x[1]), np.diff(ref_x[0]))
                ref_x[2, N] = ref_x[2, N - 1]
            elif type == "A_star" and getattr(self.setting, "ref_path", None) is not None:
                ref_x = np.array(self.setting.ref_path, dtype=float)      # Monte-Carlo worlds carry their own path
            elif type == "A_star":
                st = self.setting
                start = (st.startPose[1], st.startPose[0])
                goal = (st.goalPose[1], st.goalPose[0])
                route = self.path_solver.solve(st.org_gridMap, start, goal)
                path = self.path_solver.create_reference_path(self.path_solver.rebuild_path(route))
                ref_x = np.asarray(path).T
            elif type != "":
                # the reference returns its all-zero ref_x for a type it does not know (src/closed_loop.py:529-566): same here,
                # with a warning instead of silence
                import warnings
                warnings.warn("update_path: unknown reference type %r -- the all-zero reference is returned, as the reference does" % (type,), stacklevel=2)
            return ref_x
        # allAviable == 1: resample the current reference to N_fix segments, recompute yaw, rescale the step
        ratio = int(self.N_fix / self.N_free)
        pts = []
        for i in range(self.N_free):
            xx = np.linspace(self.xref[0][i], self.xref[0][i + 1], num=ratio, endpoint=False)
            yy = np.linspace(self.xref[1][i], self.xref[1][i + 1], num=ratio, endpoint=False)
            pts += [[xx[j], yy[j]] for j in range(ratio)]
        pts.append([self.xref[0][-1], self.xref[1][-1]])
        ref = self.path_solver.create_reference_path(pts)
        self.N_fix = len(ref) - 1
        self.Ts_opt = (self.N_free * self.Ts_opt) / self.N_fix
        self.Ts = self.Ts_opt                                  # q7: permanent overwrite (reference :587)
        return np.asarray(ref).T

    def sensor(self):
        """src/closed_loop.py:591-629: lidar gate on the car-front point; any sensed obstacle -> fixtime = 1"""
        cx, cy, th = self.x0[0], self.x0[1], self.x0[2]
        l, w = self.ego[0], self.ego[1]
        v2 = [cx + l * np.cos(th) - w * np.sin(th), cy + l * np.sin(th) + w * np.cos(th)]
        v3 = [cx + l * np.cos(th) + w * np.sin(th), cy + l * np.sin(th) - w * np.cos(th)]
        front = [(v2[0] + v3[0]) / 2, (v2[1] + v3[1]) / 2]
        self.car_front = front
        st = self.setting
        sensed = []
        self.fixtime = 0
        for i, obs in enumerate(self.dyn_loc[-1]):
            hit = 0
            for j in range(4):
                if np.sqrt((front[0] - obs[j][0]) ** 2 + (front[1] - obs[j][1]) ** 2) <= st.senseDis:
                    hit = 1
                    self.fixtime = 1
                    sensed.append(st.dyn_obs_info[i])
                    break
            obs.append(hit)
        st.dyn_obs_info = sensed
        st.dyn_nObs = len(sensed)


class BatchClosedLoop:
    """B independent rollouts advanced in lock-step: the harness of each rollout runs on the host, the
    solves of one step go to the GPU as one batch per problem shape (N, obstacle edge counts); instances
    whose fixed-time solve (variant 6) fails are re-solved as variant 8 in a second batch, like the
    reference's fallback (src/closed_loop.py:393-398)."""

    def __init__(self, rollouts, params_kw=None):
        from .solver import BatchSolver, SolverParams
        self._BatchSolver, self._SolverParams = BatchSolver, SolverParams
        self.rollouts = list(rollouts)
        self.start_order = (params_kw or {}).get("start_order", 0)           # include/obca_mpc.h: start_order
        self.solvers = {}
        self.steps_solved = 0
        self.steps_converged = 0

    def _solve_group(self, calls):
        """calls: list of (rollout index, variant, args) with identical (N, m)."""
        import torch
        packed = []
        for _, variant, a in calls:
            packed.append(pack_reference_call(variant, a[0], a[4], a[5], a[10], a[11], a[12], a[13], a[14], a[17],
                                              a[19] if variant == 6 else None))
        m = packed[0][0]
        N = calls[0][2][4]
        key = (N, tuple(m))
        B = len(calls)
        if key not in self.solvers or self.solvers[key].max_batch < B:
            self.solvers[key] = self._BatchSolver(N, m, max_batch=max(B, 64))
        a0 = calls[0][2]
        free = calls[0][1] == 4
        kw = dict(xL=a0[6], xU=a0[7], uL=a0[8], uU=a0[9], ego=a0[16], dmin=a0[15])
        kw["start_order"] = self.start_order
        kw["single_start"] = calls[0][1] == 6   # obca_mpc8 follows a failed obca_mpc6 (step()): the first start only
        prm = self._SolverParams(Q_free=a0[2], R_free=a0[3], P_free=a0[1], **kw) if free else \
            self._SolverParams(Q_fix=a0[2], R_fix=a0[3], P_fix=a0[1], **kw)
        st = lambda j: np.stack([p[j] for p in packed])
        variant = np.array([c[1] for c in calls], dtype=np.int32)
        out = self.solvers[key].solve(variant, st(1), st(2), st(3), st(4), st(5), np.array([p[6] for p in packed]),
                                      st(7), prm)
        torch.cuda.synchronize()
        x, u = out.xopt.cpu().numpy(), out.uopt.cpu().numpy()
        ts, feas = out.ts_opt.cpu().numpy(), out.feas.cpu().numpy()
        return [(x[i], u[i], bool(feas[i]), float(ts[i])) for i in range(B)]

    def _solve(self, pending):
        groups = {}
        for idx, variant, args in pending:
            m = tuple(int(args[12][i]) - 1 for i in range(int(args[11])))
            # one launch = one set of solver constants: weights, boxes, footprint and clearance are part of the key
            const = tuple(np.asarray(args[j], float).round(15).tobytes() for j in (1, 2, 6, 7, 8, 9, 16)) + \
                tuple(np.asarray(r, float).tobytes() for r in args[3]) + (float(args[15]),)
            groups.setdefault((variant, args[4], m, const), []).append((idx, variant, args))
        results = {}
        for calls in groups.values():
            for (idx, _, _), res in zip(calls, self._solve_group(calls)):
                results[idx] = res
        return results

    def step(self):
        active = [i for i, r in enumerate(self.rollouts) if not r.done and not r.goal_reached()]
        if not active:
            return 0
        pending = []
        for i in active:
            variant, args = self.rollouts[i].prepare_step()
            pending.append((i, variant, args))
        results = self._solve(pending)
        retry = [(i, 8, args[:-1]) for i, variant, args in pending if variant == 6 and not results[i][2]]
        if retry:
            results.update(self._solve(retry))
        for i in active:
            self.steps_solved += 1
            self.steps_converged += bool(results[i][2])
            self.rollouts[i].finish_step(results[i])
        return len(active)

    def run(self, max_steps=30):
        for _ in range(max_steps):
            if self.step() == 0:
                break
        return self
