"""Row N3 of SURVEY.md 8(f): the reference's open-loop two-stage planner (src/simulation.py:20-62 ->
closedLoop.mpc_openLoop_freeTime, src/closed_loop.py:113-120, then mpc_openLoop_fixTime, :122-140: the free-time plan
resampled to N_fix points, obca_mpc6 with the setting's terminal set, obca_mpc8 where that fails) and the long horizons of
src/simulation.py:225-231 (N = 10: 3.69 s, N = 74: 136.7 s published), on the GPU through the drop-in `obca` class.

Checker: the same driver on the CPU build of the structured core (tests/native; same algorithm, pinned to the dense C
oracle by tests/test_lpi_core_cpu.py) -- equal feas flags, trajectories to 1e-6 -- plus model-level properties
(dynamics, bounds, terminal constraints, geometric clearance) that do not depend on any solver."""
import time

import numpy as np
import pytest

from tests import kkt_check, native_build
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting

pytestmark = pytest.mark.gpu
EGO, DMIN = (1.7, 0.75, 1.7, 0.75), 0.05


def _gpu_solver():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
    return obca()


def _dyn_res(x, u, h):
    nxt = x[:, :-1] + h * np.stack([u[0] * np.cos(x[2, :-1]), u[0] * np.sin(x[2, :-1]), u[1]])
    return float(np.max(np.abs(nxt - x[:, 1:])))


def _static_rows(cl, N):
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import pack_reference_call
    m, _, _, _, A, b, _, _ = pack_reference_call(4, cl.Ts, N, cl.x0, np.zeros((3, N + 1)), cl.nObs, cl.vObs, cl.AObs, cl.bObs, cl.u0)
    return m, A, b


@pytest.mark.parametrize("demo,n_free,n_fix", [("demo1", 10, 20), ("demo8", 5, 10)])
def test_two_stage_open_loop_planner(demo, n_free, n_fix):
    """the reference's recommended open-loop setting (N_free = 10, N_fix = 20: resampling ratio 2, src/simulation.py:23-25)"""
    runs = {}
    for name, solver in (("gpu", _gpu_solver()), ("cpu", native_build.LpiObca())):
        cl = closedLoop(problemSetting(demo), solver=solver)
        cl.N_free, cl.N_fix = n_free, n_fix
        cl.mpc_openLoop_freeTime()
        free = (np.array(cl.xOpt), np.array(cl.uOpt), bool(cl.feas), float(cl.Ts_opt))
        m, A, b = _static_rows(cl, n_free)
        cl.mpc_openLoop_fixTime()
        if name == "cpu":
            runs["cpu_calls"] = solver.calls
        runs[name] = dict(free=free, fix=(np.array(cl.xOpt), np.array(cl.uOpt), bool(cl.feas), float(cl.Ts_opt)), N_fix=cl.N_fix,
                          xref=np.array(cl.xref), term=np.array(cl.terminal_set, float), rows=(m, A, b), goal=cl.xF)
    g, c = runs["gpu"], runs["cpu"]
    assert g["free"][2] and c["free"][2]
    np.testing.assert_allclose(g["free"][0], c["free"][0], rtol=0, atol=1e-6)          # stage 1: the same plan
    np.testing.assert_allclose(g["free"][1], c["free"][1], rtol=0, atol=1e-6)
    assert g["free"][3] == pytest.approx(c["free"][3], abs=1e-8)
    # stage 2 is a long non-convex fixed-time solve with a nearly flat objective (Q = 0.001 I): roundoff decides which of
    # several equivalent plans the iteration settles in, so GPU and CPU are held to the MODEL, not to each other
    assert g["fix"][2] and c["fix"][2]
    assert g["N_fix"] == n_fix and g["fix"][0].shape == (3, n_fix + 1)      # resampled: N_free segments x ratio + 1 points
    xf, uf, _, ts = g["free"]
    assert _dyn_res(xf, uf, ts) < 1e-7 and np.max(np.abs(xf[:, -1] - np.asarray(g["goal"], float))) < 1e-6   # reaches the goal
    m, A, b = g["rows"]
    assert kkt_check.min_clearance(xf, EGO, m, A, b) >= DMIN - 1e-6
    for r in (g, c):
        x2, u2, _, ts2 = r["fix"]
        assert ts2 == pytest.approx(n_free * r["free"][3] / n_fix, rel=1e-12)   # Ts_opt <- N_free Ts_opt / N_fix (:586)
        assert _dyn_res(x2, u2, ts2) < 1e-7
        assert np.abs(u2[0]).max() <= 0.6 + 1e-7 and np.abs(u2[1]).max() <= np.pi / 6 + 1e-7
    # and the GPU's stage-2 answer on exactly the inputs the driver built is a KKT point of the reference-pinned model
    import torch
    from oracle.obca_nlp import Problem
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    call = [q for q in runs["cpu_calls"] if q["variant"] in (6, 8)][-1]
    sp = SolverParams()
    s = BatchSolver(n_fix, call["m"], max_batch=1)
    s.enable_certificates()
    out = s.solve(call["variant"], call["x0"][None], call["u0"][None], call["xref"][None], call["A"][None], call["b"][None],
                  [call["Ts"]], call["term"][None], sp)
    torch.cuda.synchronize()
    assert int(out.status[0]) in (0, 1)
    p = Problem(call["variant"], n_fix, call["m"], call["x0"], call["u0"], call["xref"], call["A"], call["b"], call["Ts"], sp.Q_fix,
                sp.R_fix[0], sp.R_fix[1], sp.P_fix, sp.xL, sp.xU, sp.uL, sp.uU, sp.ego, sp.dmin,
                term=call["term"] if call["variant"] == 6 else None)
    cert = kkt_check.certificate(p, s.cert_z[0].cpu().numpy(), s.cert_y[0].cpu().numpy())
    for k in ("stationarity", "primal", "dual_sign", "complementarity"):
        assert cert[k] <= 1e-6, (k, cert)
    assert kkt_check.min_clearance(out.xopt[0].cpu().numpy(), EGO, call["m"], call["A"], call["b"]) >= DMIN - 1e-6


# absolute ceilings beside the self-relative bounds below (a uniformly slower build must not pass): warm solve of one instance, seconds;
# measured 0.02 (N = 10) ... 0.45 (demo9, N = 74), the reference 3.69 s / 136.7 s
ABS_CEILING_S = {10: 0.3, 40: 0.6, 50: 1.0, 66: 1.5, 74: 1.5}


@pytest.fixture(scope="module")
def stage_cost_ref():
    """seconds per (iteration x stage) of the shortest horizon (demo1, N = 10, one-CU kernel) on this box -- measured here, not taken
    from whichever parametrisation happens to run first (-k, xdist, reordering)"""
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    cpu = native_build.LpiObca()
    cl = closedLoop(problemSetting("demo1"), solver=cpu)
    cl.N_free = 10
    cl.mpc_openLoop_freeTime()
    call = cpu.calls[-1]
    s = BatchSolver(10, call["m"], max_batch=1)
    args = (4, call["x0"][None], call["u0"][None], call["xref"][None], call["A"][None], call["b"][None], [call["Ts"]], call["term"][None],
            SolverParams(xL=cl.xL[:2], xU=cl.xU[:2]))
    s.solve(*args)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = s.solve(*args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    cost = dt / max(int(out.iters[0]), 1) / 10
    s.close()
    return cost


@pytest.mark.parametrize("demo,N", [("demo1", 10), ("demo1", 40), ("demo1", 74), ("demo9", 50), ("demo9", 66), ("demo9", 74), ("demo9", 10)])
def test_long_horizon_free_time_solves(demo, N, stage_cost_ref):
    """src/simulation.py:225-231: `mpc.N_free = 10 # np.size(a_start_path, 0)` -- 3.69 s at N = 10 and 136.7 s at N = 74
    on demo9.  Cold start, start/goal-only reference.  Beyond the LDS (N > 26) the plan runs on the four-wavefront kernel with
    its rows in an HBM workspace (obca_ipm_kernel_gm): every solve in well under a second (the same solve
    took 76 s on one lane of the lane kernel).  demo9 at N = 10 is infeasible by construction (the time-scale bound max_Topt
    allows a path of (dx + dy) + 0.6 m in ten straight segments, the obstacles need a detour) and is reported as such by GPU
    and CPU alike.  demo9 at N = 70 ... 74 -- the case the reference timed (main.py:28) -- looked like a lottery until round 4
    (outcome flipping with a 1e-10 perturbation of the start pose): its objective is ~8e4 and the l1 penalty is exact only from
    rho = 1e7; with the second level of the penalty escalation it converges from every start."""
    import torch
    from oracle.obca_nlp import Problem
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    cpu = native_build.LpiObca()
    cl = closedLoop(problemSetting(demo), solver=cpu)
    cl.N_free = N
    t = time.perf_counter()
    cl.mpc_openLoop_freeTime()
    t_cpu = time.perf_counter() - t
    call = cpu.calls[-1]
    sp = SolverParams(xL=cl.xL[:2], xU=cl.xU[:2])                # the setting's map size (demo9: 40 x 60)
    s = BatchSolver(N, call["m"], max_batch=1)
    s.enable_certificates()
    args = (4, call["x0"][None], call["u0"][None], call["xref"][None], call["A"][None], call["b"][None], [call["Ts"]], call["term"][None], sp)
    t = time.perf_counter()
    out = s.solve(*args)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t
    st, it = int(out.status[0]), int(out.iters[0])
    print("%s N=%d: ONE instance on the GPU %.3f s (%d iterations, status %d), on one CPU core %.3f s (%d); reference, unspecified hardware: %s"
          % (demo, N, t_gpu, it, st, t_cpu, call["iters"], {10: "3.69 s", 74: "136.7 s"}.get(N, "not published")))
    # time bound from a measured cost instead of literals: the SAME solve repeated on the warm handle gives this box's cost per
    # interior-point iteration at this horizon; the first launch (workspace allocation, code-object load) may add a fixed 0.1 s,
    # and an iteration may cost at most 4x what one iteration of the shortest horizon (demo1, N = 10, one-CU kernel) costs per
    # stage -- the O(N) claim of SURVEY 8f-N3 (the reference: 3.69 s -> 136.7 s from N = 10 to 74)
    t = time.perf_counter()
    out = s.solve(*args)
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t
    assert int(out.iters[0]) == it and int(out.status[0]) == st          # deterministic
    assert t_gpu < t_warm + 0.1 + 0.5 * t_warm, (t_gpu, t_warm)
    per_stage_iter = t_warm / max(it, 1) / N
    assert per_stage_iter < 4.0 * stage_cost_ref, (per_stage_iter, stage_cost_ref)
    assert t_warm < ABS_CEILING_S[N], (t_warm, ABS_CEILING_S[N])
    if (demo, N) == ("demo9", 10):
        assert st == 2 and not cl.feas       # infeasible by construction, reported as such
        return
    assert st in (0, 1) and cl.feas
    x, u, ts = out.xopt[0].cpu().numpy(), out.uopt[0].cpu().numpy(), float(out.ts_opt[0])
    if it == call["iters"]:                  # same iterate sequence as the CPU build: the same plan
        np.testing.assert_allclose(x, np.array(cl.xOpt), rtol=0, atol=1e-5)
        assert ts == pytest.approx(float(cl.Ts_opt), abs=1e-7)
    # whichever path the 300-800 iterations took: a KKT point of the reference-pinned model that reaches the goal
    p = Problem(4, N, call["m"], call["x0"], call["u0"], call["xref"], call["A"], call["b"], call["Ts"], sp.Q_free, sp.R_free[0],
                sp.R_free[1], sp.P_free, sp.xL, sp.xU, sp.uL, sp.uU, sp.ego, sp.dmin)
    cert = kkt_check.certificate(p, s.cert_z[0].cpu().numpy(), s.cert_y[0].cpu().numpy())
    # complementarity ends at mu_final / (objective scaling) = 2.5e-9 x max(|grad|, rho) / 100; an instance that needed the
    # penalty escalation (demo9: rho = 1e6, at N = 50 the second level 1e7) is scaled 100 / 1000 x harder: the bound on the
    # other residuals is relative to the objective's size, the one on complementarity is the barrier parameter's last value
    # at the largest penalty of the escalation, 2.5e-9 x 1e7 / 100 (4.7e-9 of the objective at N = 50)
    tol = 1e-6 * max(1.0, 1e-3 * abs(cert["objective"]))
    for k in ("stationarity", "primal", "dual_sign"):
        assert cert[k] <= tol, (k, cert)
    assert cert["complementarity"] <= max(tol, 2.6e-4), cert
    assert _dyn_res(x, u, ts) < 1e-7 and np.max(np.abs(x[:, -1] - np.asarray(cl.xF, float))) < 1e-6
    assert kkt_check.min_clearance(x, EGO, call["m"], call["A"], call["b"]) >= DMIN - 1e-6
    assert np.abs(u[0]).max() <= 0.6 + 1e-7
