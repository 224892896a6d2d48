"""Device-resident closed loop: B receding-horizon rollouts advanced on the GPU without host round trips.

Batched counterpart of the reference's ``closedLoop.closed_loop_mpc4`` (reference src/closed_loop.py:323-443):
``pack_worlds`` turns B ``problemSetting`` objects into the structure-of-arrays the C ABI takes
(``obca_rollouts_*`` in include/obca_mpc.h), ``DeviceRollouts`` drives them.  The per-rollout Python class
``closed_loop.closedLoop`` stays the readable mirror of the reference; tests compare the two step by step.
"""
import ctypes

import numpy as np

from . import _lib
from .a_star import a_star
from .model_obstacle import obstacleModel
from .solver import SolverParams


class PackedWorlds:
    """start [B,3], goal [B,2], path [B,3,P], path_len [B], static_A [B,Ms,2], static_b [B,Ms], dyn [B,nDyn,13]"""
    __slots__ = ("m_static", "n_dyn", "start", "goal", "path", "path_len", "static_A", "static_b", "dyn", "sense_dis",
                 "xL", "xU")

    @property
    def batch(self):
        return self.start.shape[0]

    def slice(self, lo, hi):
        w = PackedWorlds()
        w.m_static, w.n_dyn, w.sense_dis = self.m_static, self.n_dyn, self.sense_dis
        w.xL, w.xU = self.xL, self.xU
        for k in ("start", "goal", "path", "path_len", "static_A", "static_b", "dyn"):
            setattr(w, k, getattr(self, k)[lo:hi])
        return w


def reference_path(setting):
    """the (3,P) reference the closed loop tracks: the setting's own path or the A* route (src/closed_loop.py:340)"""
    if getattr(setting, "ref_path", None) is not None:
        return np.asarray(setting.ref_path, float)
    start = (setting.startPose[1], setting.startPose[0])
    goal = (setting.goalPose[1], setting.goalPose[0])
    planner = a_star(setting.org_gridMap, start, goal)
    route = planner.solve(setting.org_gridMap, start, goal)
    return np.asarray(planner.create_reference_path(planner.rebuild_path(route)), float).T


def device_reference_paths(settings):
    """the A* references of all settings in one GPU launch (planner.plan_batch); settings that carry their own
    ``ref_path`` keep it.  Grids must share one shape."""
    from .planner import plan_batch
    todo = [i for i, s in enumerate(settings) if getattr(s, "ref_path", None) is None]
    paths = [None if i in todo else np.asarray(s.ref_path, float) for i, s in enumerate(settings)]
    if todo:
        grids = np.stack([np.asarray(settings[i].org_gridMap) for i in todo]).astype(np.uint8)
        starts = [(settings[i].startPose[1], settings[i].startPose[0]) for i in todo]
        goals = [(settings[i].goalPose[1], settings[i].goalPose[0]) for i in todo]
        path, plen = plan_batch(grids, starts, goals)
        path, plen = path.cpu().numpy(), plen.cpu().numpy()
        for j, i in enumerate(todo):
            if plen[j] < 1:
                raise RuntimeError("no A* route for setting %d (code %d)" % (i, plen[j]))
            paths[i] = path[j, :, :plen[j]].copy()
    return paths


def pack_worlds(settings, path_max=None, planner="host"):
    """B settings of one shape (same static edge counts, same number of moving rectangles) -> PackedWorlds.
    planner: "host" = the per-setting Python A* mirror, "device" = one batched GPU search."""
    settings = list(settings)
    om = obstacleModel()
    m0 = [int(v) - 1 for v in settings[0].static_vObs]
    nd = len(settings[0].dyn_obs_info)
    if nd > _lib.OBCA_MAX_DYN:
        raise ValueError("at most %d moving obstacles per rollout" % _lib.OBCA_MAX_DYN)
    paths = device_reference_paths(settings) if planner == "device" else [reference_path(s) for s in settings]
    P = max(p.shape[1] for p in paths) if path_max is None else int(path_max)
    B, Ms = len(settings), sum(m0)
    w = PackedWorlds()
    w.m_static, w.n_dyn = m0, nd
    w.start = np.zeros((B, 3)); w.goal = np.zeros((B, 2)); w.path = np.zeros((B, 3, P))
    w.path_len = np.zeros(B, dtype=np.int32)
    w.static_A = np.zeros((B, Ms, 2)); w.static_b = np.zeros((B, Ms)); w.dyn = np.zeros((B, nd, 13))
    w.sense_dis = float(settings[0].senseDis)
    # the position box the reference hands to every solve (setting.xL / xU, src/closed_loop.py:38-39): one per batch
    w.xL = tuple(float(v) for v in settings[0].xL[:2])
    w.xU = tuple(float(v) for v in settings[0].xU[:2])
    for i, s in enumerate(settings):
        if [int(v) - 1 for v in s.static_vObs] != m0 or len(s.dyn_obs_info) != nd or float(s.senseDis) != w.sense_dis:
            raise ValueError("setting %d has a different shape than setting 0" % i)
        if tuple(float(v) for v in s.xL[:2]) != w.xL or tuple(float(v) for v in s.xU[:2]) != w.xU:
            raise ValueError("setting %d has a different position box (xL, xU) than setting 0: one batch, one map size" % i)
        w.start[i] = np.asarray(s.startPose[:3], float)
        w.goal[i] = np.asarray(s.goalPose[:2], float)
        p = paths[i]
        w.path[i, :, :p.shape[1]] = p
        w.path[i, :, p.shape[1]:] = p[:, -1:]
        w.path_len[i] = p.shape[1]
        A, b = om.obstacle_H_Represent(s.static_nObs, s.static_vObs, s.static_lObs)
        w.static_A[i], w.static_b[i] = A, b[:, 0]
        for j, d in enumerate(s.dyn_obs_info):
            w.dyn[i, j, :11] = np.asarray(d[:11], float)
            w.dyn[i, j, 11], w.dyn[i, j, 12] = np.cos(d[2]), np.sin(d[2])
    return w


def rollout_dims(w, N, max_steps, device=0, N_fix=None):
    d = _lib.ObcaRolloutDims()
    d.N, d.n_static, d.n_dyn = int(N), len(w.m_static), int(w.n_dyn)
    d.N_fix = int(N if N_fix is None else N_fix)
    for i, v in enumerate(w.m_static):
        d.m_static[i] = int(v)
    d.path_max, d.batch, d.max_steps, d.device = int(w.path.shape[2]), int(w.batch), int(max_steps), int(device)
    return d


class RolloutCohorts:
    """The batch cut into ``cohorts`` independent lock-step groups, each on its own HIP stream.  One step of a
    group lasts as long as its slowest solve (an infeasible obca_mpc6 runs to max_iter = 1000 before the obca_mpc8
    retry, exactly like the reference); with several groups in flight the GPU works on the others meanwhile.
    Rollouts are independent, so the results do not depend on the cut."""

    def __init__(self, worlds, cohorts=8, **kw):
        import torch
        self.torch = torch
        w = worlds if isinstance(worlds, PackedWorlds) else pack_worlds(worlds)
        B = w.batch
        cohorts = max(1, min(int(cohorts), B))
        cuts = [B * i // cohorts for i in range(cohorts + 1)]
        self.streams = [torch.cuda.Stream() for _ in range(cohorts)]
        self.parts = []
        for i in range(cohorts):
            with torch.cuda.stream(self.streams[i]):
                self.parts.append(DeviceRollouts(w.slice(cuts[i], cuts[i + 1]), **kw))
        self.max_steps = self.parts[0].max_steps

    def _each(self, fn):
        res = []
        for st, p in zip(self.streams, self.parts):
            with self.torch.cuda.stream(st):
                res.append(fn(p))
        return res

    def reset(self):
        self._each(lambda p: p.reset())

    def run(self, n_steps=None):
        n = self.max_steps if n_steps is None else int(n_steps)
        if all(p.mode == 0 for p in self.parts):
            self._each(lambda p: p.run(n))                  # fused kernels: one launch per cohort
        else:
            for _ in range(n):
                self._each(lambda p: p.step())
        return self

    def read(self):
        outs = self._each(lambda p: p.read())
        for st in self.streams:
            st.synchronize()
        return {k: self.torch.cat([o[k] for o in outs]) for k in outs[0]}


class DeviceRollouts:
    """B rollouts on one GPU.  ``step()`` enqueues one receding-horizon step of every running rollout;
    ``run()`` all of them; ``read()`` returns state and history (torch tensors on the device)."""

    def __init__(self, worlds, N=6, params=None, Ts0=0.1, max_steps=30, device=None, warm_start=None, N_fix=None):
        """N: horizon of the free-time problem, N_fix (default N): of the fixed-time problems (the reference's committed
        default is N_free = N_fix = 6, src/closed_loop.py:84,91; N_fix must be a multiple of N with N_fix - 5 <= N).
        params: None = the reference's controller constants with the position box of the worlds (setting.xL / xU).
        warm_start: None = the reference's cold start of every solve; a float mu_init = start each step whose
        problem shape equals the previous step's from the shifted previous plan (NOT reference behaviour)."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceRollouts needs a ROCm GPU; there is no CPU fallback on the product path")
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.w = worlds if isinstance(worlds, PackedWorlds) else pack_worlds(worlds)
        self.N, self.max_steps, self.Ts0 = int(N), int(max_steps), float(Ts0)
        self.N_fix = self.N if N_fix is None else int(N_fix)
        self.params = params or SolverParams(xL=getattr(self.w, "xL", (0.0, 0.0)), xU=getattr(self.w, "xU", (39.0, 10.0)))
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._dims = rollout_dims(self.w, N, max_steps, dev_index, self.N_fix)
        h = ctypes.c_void_p()
        _lib.check(self.lib.obca_rollouts_create(ctypes.byref(self._dims), ctypes.byref(h)))
        self._h = h
        self.steps_enqueued = 0
        self.mode = 0
        if warm_start is not None:
            _lib.check(self.lib.obca_rollouts_set_warm_start(self._h, 1, float(warm_start)))
        self.reset()

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self):
        t, w = self.torch, self.w
        dev = lambda a, dt: t.as_tensor(np.ascontiguousarray(a), dtype=dt, device=self.device)
        self._inputs = [dev(w.start, t.float64), dev(w.goal, t.float64), dev(w.path, t.float64),
                        dev(w.path_len, t.int32), dev(w.static_A, t.float64), dev(w.static_b, t.float64),
                        dev(w.dyn if w.n_dyn else np.zeros((w.batch, 1, 13)), t.float64)]
        self._cparams = self.params.to_c()
        ptrs = [ctypes.c_void_p(x.data_ptr()) for x in self._inputs]
        _lib.check(self.lib.obca_rollouts_reset(self._h, *ptrs, self.Ts0, w.sense_dis, ctypes.byref(self._cparams),
                                                self._stream()))
        self.steps_enqueued = 0

    def step(self):
        _lib.check(self.lib.obca_rollouts_step(self._h, self._stream()))
        self.steps_enqueued += 1

    def run(self, n_steps=None):
        """all steps; one persistent-kernel launch when every shape fits the wave kernel (see obca_rollouts_run)"""
        n = self.max_steps if n_steps is None else int(n_steps)
        _lib.check(self.lib.obca_rollouts_run(self._h, n, self._stream()))
        self.steps_enqueued += n
        return self

    def queue_mode(self):
        """2: one work queue per XCD (MI355X default), 1: one global queue, 0: one workgroup per rollout (obca_rollouts_queue_mode)"""
        return int(self.lib.obca_rollouts_queue_mode(self._h))

    def set_mode(self, mode):
        """'fused' (default where it fits) | 'lockstep' (one launch per problem shape and step)"""
        self.mode = {"fused": 0, "lockstep": 1}.get(mode, mode)
        _lib.check(self.lib.obca_rollouts_set_mode(self._h, self.mode))

    def read(self):
        t, B, S, N1, nd = self.torch, self.w.batch, self.max_steps, max(self.N, self.N_fix) + 1, self.w.n_dyn
        f = lambda *shape: t.empty(*shape, dtype=t.float64, device=self.device)
        i = lambda *shape: t.empty(*shape, dtype=t.int32, device=self.device)
        out = {"x_closed": f(B, S + 1, 3), "u_closed": f(B, S, 2), "T_closed": f(B, S), "x_openloop": f(B, S, 3, N1),
               "variant": i(B, S), "iters": i(B, S), "status": i(B, S), "dyn": f(B, S, nd, 4) if nd else t.zeros(B, S, 1, 4, dtype=t.float64, device=self.device), "steps": i(B), "flags": i(B)}
        order = ("x_closed", "u_closed", "T_closed", "x_openloop", "variant", "iters", "status", "dyn", "steps", "flags")
        ptrs = [ctypes.c_void_p(out[k].data_ptr()) if (k != "dyn" or nd) else None for k in order]
        _lib.check(self.lib.obca_rollouts_read(self._h, *ptrs, self._stream()))
        return out

    def debug_harness(self, k, Ts_opt, x0=None, g=0):
        """test hook (obca_rollouts_debug_harness): the harness part of a step alone with step counter, inherited step length
        and pose given; returns what it handed the solver of group g: (variant [B], A [B,N_g+1,M_g,2], b [B,N_g+1,M_g]) as numpy"""
        B, Ng = self.w.batch, (self.N if g == 0 else self.N_fix)
        Mg = self.w.static_A.shape[1] + 4 * g
        var, A, b = np.zeros(B, np.int32), np.zeros((B, Ng + 1, Mg, 2)), np.zeros((B, Ng + 1, Mg))
        x0a = None if x0 is None else np.ascontiguousarray(x0, float)
        p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(self.lib.obca_rollouts_debug_harness(self._h, int(k), float(Ts_opt), p(x0a), int(g), p(var), p(A), p(b), self._stream()))
        return var, A, b

    def close(self):
        if getattr(self, "_h", None):
            self.lib.obca_rollouts_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reference_lists(out, worlds, i, N_free=None):
    """History of rollout ``i`` in the form the reference's ``closedLoop`` keeps for its plot routine
    (src/closed_loop.py:416-441, src/draw.py:333-456): ``x_openLoop`` list of (N+1,3) arrays, ``x_closed`` list of
    poses, ``u_closed`` list of inputs, ``Ts_opt`` list, ``dyn_loc`` list (per step) of the present moving
    rectangles as 5 clockwise vertices + lidar flag.  ``out`` = ``DeviceRollouts.read()`` (tensors or arrays)."""
    from .model_obstacle import rectangle_vertices
    g = lambda k: np.asarray(out[k].cpu() if hasattr(out[k], "cpu") else out[k])
    k = int(g("steps")[i])
    tried = int((g("variant")[i] > 0).sum())              # a failed last step still updated the obstacles
    var = g("variant")[i]
    n_cols = g("x_openloop").shape[3]
    n_free = N_free                       # only needed when N_fix > N_free: free-time plans fill N_free + 1 of the columns
    cols = lambda j: (n_free + 1) if (n_free is not None and var[j] == 4) else n_cols
    res = {"x_openLoop": [g("x_openloop")[i, j][:, :cols(j)].T.copy() for j in range(k)],
           "x_closed": [g("x_closed")[i, j].copy() for j in range(k + 1)],
           "u_closed": [g("u_closed")[i, j].copy() for j in range(k)],
           "Ts_opt": [float(v) for v in g("T_closed")[i, :k]],
           "dyn_loc": []}
    dyn = g("dyn")[i]
    for j in range(tried):
        verts = []
        for q in range(worlds.n_dyn):
            cx, cy, present, hit = dyn[j, q]
            if present:
                d = worlds.dyn[i, q]
                verts.append(rectangle_vertices(cx, cy, d[2], d[3], d[4]) + [int(hit)])
        res["dyn_loc"].append(verts)
    return res
