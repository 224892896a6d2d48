"""GPU outputs certified against the reference-PINNED model, not against the build's own interior-point method.

oracle/obca_nlp.py is pinned to the reference's model-building code (tests/golden/nlp_eval.json, 1e-12).  These tests take
what the HIP kernels return through the C ABI -- trajectory, full primal vector and multipliers
(obca_set_certificate_buffers) -- and evaluate the first-order optimality conditions of THAT model at it:
stationarity, primal feasibility, multiplier signs and complementarity, all <= 1e-6 in the objective's own units
(observed: 1e-7 / 1e-11 / 0 / 3e-7; complementarity sits at mu_final / objective scaling = 2.5e-9 / 1e-2).  Independent of
any solver, an explicit geometric check (car rectangle against every obstacle polygon, Euclidean distance >= dmin - 1e-6)
runs at the full sizes of BASELINE.json's configs: C2 8192, C3 both halves at 8192 unique seeds, C5 4096 rollouts.
"""
import os

import numpy as np
import pytest
import torch

from tests import kkt_check
from tests.test_oracle_nlp import build

pytestmark = pytest.mark.gpu

TOL = 1e-6
PROCS = max(1, min(32, (os.cpu_count() or 1) // 2))


def _solve(b, N, mode=None, cert=True):
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    B = len(b["variant"])
    s = BatchSolver(N, b["m"], max_batch=B, mode=mode)
    if cert:
        s.enable_certificates()
    out = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
    torch.cuda.synchronize()
    r = dict(x=out.xopt.cpu().numpy(), u=out.uopt.cpu().numpy(), ts=out.ts_opt.cpu().numpy(), st=out.status.cpu().numpy(),
             it=out.iters.cpu().numpy(), info=out.info.cpu().numpy())
    if cert:
        r["z"], r["y"] = s.cert_z[:B].cpu().numpy(), s.cert_y[:B].cpu().numpy()
    s.close()
    return r


def _assert_certified(p, z, y, x, u, what):
    c = kkt_check.certificate(p, z, y)
    for k in ("stationarity", "primal", "dual_sign", "complementarity"):
        assert c[k] <= TOL, (what, k, c)
    xs, us = p.unpack_xu(np.asarray(z)[:p.n])                       # the certified point IS the returned trajectory
    assert np.array_equal(xs, x) and np.array_equal(us, u), what
    return c


GOLDEN = ["demo1_N6_mpc4_step0", "demo9_N5_mpc4_step0", "demo8_N5_mpc4_step0", "slanted_asym_mpc4", "slanted_asym_mpc6",
          "slanted_asym_mpc8", "demo1_dyn_mpc8", "demo1_dyn_mpc6"]
# feas = False expected: demo1 at N = 5 is infeasible by construction.  demo1_dyn_mpc6 -- Appendix C's
# "mpc6 witness": a feasible point with f = 0.029735 exists, passing above the moving box -- is among the certified ones:
# from the reference's cold start the method ends at an infeasible stationary point (the plan that dives below the box),
# the restart phase (oracle/ipm_dense.py:solve) finds the plan above it; f <= 0.02974 is asserted below.
NOT_FEASIBLE = ["demo1_N5_mpc4_step0"]
F_UPPER = {"demo1_dyn_mpc6": 0.02974}


@pytest.mark.parametrize("name", GOLDEN + NOT_FEASIBLE)
def test_golden_cases_carry_a_kkt_certificate_of_the_pinned_model(nlp_golden, name):
    """all nine golden scenarios (reference-shaped inputs: demo worlds, slanted obstacles, asymmetric footprint, full
    weight matrices, time-varying rows) -- eight certified, one reported feas = False (see NOT_FEASIBLE)"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams, pack_reference_call
    case = [c for c in nlp_golden if c["name"] == name][0]
    a = case["inputs"]
    p = build(case)
    m, x0, u0, xr, A, b, ts, term = pack_reference_call(case["variant"], a["Ts"], a["N"], a["x0"], a["xref"], a["nObs"],
                                                        a["vObs"], a["AObs"], a["bObs"], a["u0"], a.get("terminal_set"))
    R = [np.array(r) for r in a["R"]]
    kw = dict(xL=a["xL"], xU=a["xU"], uL=a["uL"], uU=a["uU"], ego=a["ego"], dmin=a["dmin"])
    kw.update(dict(Q_free=a["Q"], R_free=R, P_free=a["P"]) if case["variant"] == 4 else dict(Q_fix=a["Q"], R_fix=R, P_fix=a["P"]))
    s = BatchSolver(a["N"], m, max_batch=1)
    s.enable_certificates()
    out = s.solve(case["variant"], x0[None], u0[None], xr[None], A[None], b[None], [ts], term[None], SolverParams(**kw))
    torch.cuda.synchronize()
    st = int(out.status[0])
    if name in NOT_FEASIBLE:
        assert st == 2                                              # converged with elastic variables left: feas = False
        assert np.all(np.isfinite(out.xopt.cpu().numpy()))          # last iterate returned, like the reference's except:
        return
    assert st in (0, 1)
    c = _assert_certified(p, s.cert_z[0].cpu().numpy(), s.cert_y[0].cpu().numpy(), out.xopt[0].cpu().numpy(),
                          out.uopt[0].cpu().numpy(), name)
    assert kkt_check.min_clearance(out.xopt[0].cpu().numpy(), a["ego"], m, A, b) >= a["dmin"] - 1e-6
    assert abs(c["objective"] - float(out.info[0, 0])) <= 1e-9 * max(1.0, abs(c["objective"]))
    if name in F_UPPER:
        assert c["objective"] <= F_UPPER[name] + 1e-6


def test_c2_batch_is_certified_instance_by_instance():
    """768 seeded C2 instances (config 2's generator): every converged answer is a KKT point of the pinned model"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    B, N = 768, 5
    b = sc.make_batch(B, N)
    r = _solve(b, N)
    ok = (r["st"] == 0) | (r["st"] == 1)
    assert ok.mean() > 0.995
    worst = dict(stationarity=0.0, primal=0.0, dual_sign=0.0, complementarity=0.0)
    for i in np.flatnonzero(ok):
        c = _assert_certified(kkt_check.problem_of(b, i, N), r["z"][i], r["y"][i], r["x"][i], r["u"][i], ("c2", i))
        for k in worst:
            worst[k] = max(worst[k], c[k])
    print("C2 certificate maxima over %d instances: %s" % (ok.sum(), worst))


@pytest.mark.parametrize("gated", [False, True])
def test_c3_instances_are_certified_at_N20(gated):
    """config 3 at its real horizon (N = 20; gated half: obca_mpc6, five obstacles, two moving, time-varying rows):
    24 instances per half through the four-wavefront kernel, each converged one certified on the pinned model"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    N, B = 20, 24
    b = sc.make_batch_c3(B, N, gated=gated)
    r = _solve(b, N)
    ok = (r["st"] == 0) | (r["st"] == 1)
    assert ok.sum() >= (22 if gated else 23)                      # measured at 8192 unique seeds: 99.6 % / 100 %
    for i in np.flatnonzero(ok):
        _assert_certified(kkt_check.problem_of(b, i, N), r["z"][i], r["y"][i], r["x"][i], r["u"][i], ("c3", gated, i))
    cl = kkt_check.min_clearance_boxes(r["x"][ok], (1.7, .75, 1.7, .75), b["m"], b["A"][ok], b["b"][ok])
    assert cl.min() >= 0.05 - 1e-6


def test_every_kernel_hands_out_the_same_certificate():
    """wave, four-wavefront and lane kernels on the same instances: identical iterates, so identical multipliers"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    b = sc.make_batch(64, 5)
    rs = [_solve(b, 5, mode=m) for m in ("wave", "multiwave", "lane")]
    for r in rs[1:]:
        same = r["it"] == rs[0]["it"]
        assert same.mean() > 0.9
        # same algorithm, different expression forms: the points agree to the solver tolerance (1e-8 on the scaled
        # optimality error), the multipliers -- which that tolerance determines less sharply -- to 1e-4 of their size
        assert np.max(np.abs(r["y"][same] - rs[0]["y"][same])) < 1e-4 * max(1.0, np.max(np.abs(rs[0]["y"])))
        assert np.max(np.abs(r["z"][same] - rs[0]["z"][same])) < 1e-6


# ------------------------------------------------------------------------------------------------ full-size clearance
def test_c2_full_size_clearance():
    """B = 8192 seeded instances: geometric clearance of every returned plan, all stages"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    B, N = 8192, 5
    b = sc.make_batch(B, N)
    r = _solve(b, N, cert=False)
    ok = (r["st"] == 0) | (r["st"] == 1)
    assert ok.mean() > 0.999                                        # measured: 8192 of 8192
    cl = kkt_check.min_clearance_boxes(r["x"][ok], sc.EGO, b["m"], b["A"][ok], b["b"][ok])
    assert cl.min() >= sc.DMIN - 1e-6, cl.min()


@pytest.mark.parametrize("gated", [False, True])
def test_c3_full_size_clearance_on_unique_seeds(gated):
    """config 3 at B = 8192 UNIQUE seeded instances per half (N = 20): clearance, dynamics, bounds and terminal set of
    every converged plan; the share of converged instances is reported by bench.py"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    B, N = 8192, 20
    b = sc.make_batch_c3(B, N, gated=gated, procs=PROCS)
    assert len(np.unique(b["x0"], axis=0)) == B
    r = _solve(b, N, cert=False)
    ok = (r["st"] == 0) | (r["st"] == 1)
    assert ok.mean() > (0.97 if gated else 0.995)                  # measured: 99.79 % / 100 % (within 3 points)
    x, u, ts = r["x"][ok], r["u"][ok], r["ts"][ok]
    cl = kkt_check.min_clearance_boxes(x, sc.EGO, b["m"], b["A"][ok], b["b"][ok])
    assert cl.min() >= sc.DMIN - 1e-6, cl.min()
    h = ts[:, None]
    nxt = np.stack([x[:, 0, :-1] + h * u[:, 0] * np.cos(x[:, 2, :-1]), x[:, 1, :-1] + h * u[:, 0] * np.sin(x[:, 2, :-1]),
                    x[:, 2, :-1] + h * u[:, 1]], 1)
    assert np.max(np.abs(nxt - x[:, :, 1:])) < 1e-7
    assert np.abs(u[:, 0]).max() <= 0.6 + 1e-7 and np.abs(u[:, 1]).max() <= np.pi / 6 + 1e-7
    du = np.diff(np.concatenate([b["u0"][ok][:, :, None], u], 2), axis=2) / h[:, None]
    assert np.abs(du[:, 0]).max() <= 0.6 + 1e-6 and np.abs(du[:, 1]).max() <= np.pi / 6 + 1e-6
    assert x[:, 1].min() >= -1e-7 and x[:, 1].max() <= 10 + 1e-7
    if gated:
        t = b["term"][ok]
        assert (x[:, 0, -1] >= t[:, 0] - 1e-6).all() and (x[:, 1, -1] >= t[:, 1] - 1e-6).all() and (x[:, 1, -1] <= t[:, 2] + 1e-6).all()
    else:
        assert np.abs(x[:, :, -1] - b["xref"][ok][:, :, -1]).max() < 1e-6


def test_c5_rollouts_clearance_at_full_size():
    """config 5 at B = 4096 rollouts (two moving boxes): (a) every open-loop plan of every step keeps dmin to the static
    obstacles at all its stages; (b) every pose reached by a fixed-time step (obca_mpc6 / obca_mpc8) keeps dmin to the
    moving boxes that step was given -- the box one step later sits exactly where stage 1 of the plan predicted it
    (advance by Ts_opt, reference src/closed_loop.py:468-471).  Steps on which the lidar gate saw only some of the present
    boxes are left out of (b): there the reference pairs vertex lists and velocities of different boxes."""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    B = 4096
    w = pack_worlds([sc.make_world_c5(i, n_dyn=2) for i in range(B)])
    dr = DeviceRollouts(w, N=5)
    dr.run()
    o = {k: v.cpu().numpy() for k, v in dr.read().items()}
    S = o["variant"].shape[1]
    done = np.arange(S)[None, :] < o["steps"][:, None]                     # successful steps
    plans = o["x_openloop"][done]                                          # [K, 3, N+1]
    idx = np.nonzero(done)[0]
    Ms = w.static_A.shape[1]
    A = np.broadcast_to(w.static_A[idx][:, None], (len(idx), 6, Ms, 2))
    b = np.broadcast_to(w.static_b[idx][:, None], (len(idx), 6, Ms))
    cl = kkt_check.min_clearance_boxes(plans, sc.EGO, w.m_static, A, b)
    assert cl.min() >= sc.DMIN - 1e-6, cl.min()
    # (b) poses reached by fixed-time steps against the boxes of the following step
    dyn = o["dyn"]                                                         # [B, S, 2, 4]: cx, cy, present, sensed
    fixed = done & (o["variant"] >= 6)
    fixed[:, -1] = False
    nxt_recorded = np.zeros_like(fixed)
    nxt_recorded[:, :-1] = o["variant"][:, 1:] > 0                        # the following step ran update_obstacle
    consistent = (dyn[..., 2] == dyn[..., 3]).all(-1)                      # gate saw every present box
    sel = fixed & nxt_recorded & consistent
    bi, si = np.nonzero(sel)
    assert len(bi) > 10000
    worst = np.inf
    for q in range(2):
        on = dyn[bi, si, q, 3] > 0
        cx, cy = dyn[bi[on], si[on] + 1, q, 0], dyn[bi[on], si[on] + 1, q, 1]
        hl, hw = w.dyn[bi[on], q, 3] / 2, w.dyn[bi[on], q, 4] / 2
        Ab = np.broadcast_to(np.array([[1.0, 0], [-1, 0], [0, 1], [0, -1]]), (on.sum(), 1, 4, 2))
        bb = np.stack([cx + hl, -(cx - hl), cy + hw, -(cy - hw)], -1)[:, None]
        pose = o["x_closed"][bi[on], si[on] + 1][:, :, None]
        worst = min(worst, kkt_check.min_clearance_boxes(pose, sc.EGO, [4], Ab, bb).min())
    assert worst >= sc.DMIN - 1e-6, worst
