"""The ladder's dodge rung and the terminal-set screen of obca_mpc6 (include/obca_mpc.h: dodge, terminal_screen) on the GPU: the
kernels against the structured host core (same rule, same arithmetic) on the obca_mpc6 calls of closed loops -- the reference's
demo11 run (N = 6; the dodge rung answers three of its calls) and C5 worlds (N = 5; most failing calls are screened)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _calls(setting=None, world=None, N=None, steps=40):
    from tests import native_build, reference_report
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    s = native_build.LpiObca()
    if setting is not None:
        reference_report.replay(setting, s, steps)
    else:
        cl = closedLoop(sc.make_world_c5(world, n_dyn=2), solver=s)
        cl.N_free = cl.N_fix = N
        cl.closed_loop_mpc4()
    return [c for c in s.calls if c["variant"] == 6]


def _solve_all(calls, modes, **prm):
    """each call's shape through every kernel mode that holds it; -> per call the host core's answer and the kernels'"""
    import torch
    from oracle import c_oracle
    from tests import native_build
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    sp = SolverParams(single_start=True, **prm)
    out = []
    by_shape = {}
    for c in calls:
        by_shape.setdefault((c["xref"].shape[1] - 1, tuple(c["m"])), []).append(c)
    for (N, m), cs in by_shape.items():
        arr = lambda k: np.stack([c[k] for c in cs])
        args = (np.full(len(cs), 6, np.int32), arr("x0"), arr("u0"), arr("xref"), arr("A"), arr("b"), np.array([c["Ts"] for c in cs]), arr("term"))
        host = native_build.lpi_solve(6, N, list(m), *args[1:6], args[6], args[7],
                                      c_oracle.default_params(xL=sp.xL, xU=sp.xU, uL=sp.uL, uU=sp.uU, ego=sp.ego, dmin=sp.dmin, single_start=1, Qx=sp.Q_fix,
                                                              Px=sp.P_fix, R1x=sp.R_fix[0], R2x=sp.R_fix[1], dodge=prm.get("dodge", True),
                                                              terminal_screen=prm.get("terminal_screen", True)))
        for mode in modes:
            s = BatchSolver(N, list(m), max_batch=len(cs))
            try:
                s.set_mode(mode)
            except RuntimeError:
                continue
            o = s.solve(*args, sp)
            torch.cuda.synchronize()
            out.append((mode, host, {k: getattr(o, k).cpu().numpy() for k in ("xopt", "uopt", "ts_opt", "status", "iters", "info")}))
            s.close()
    return out


def test_demo11_calls_kernels_equal_host_core_through_the_dodge_rung():
    from tests import reference_report
    calls = _calls(setting=reference_report.demo11_setting(), steps=32)
    assert len(calls) >= 15
    res = _solve_all(calls, ("auto", "multiwave", "lane"))
    assert len(res) >= 3
    ref, n_dodged = {}, 0
    for mode, host, dev in res:
        assert np.array_equal(dev["status"], host["status"]) and np.all(dev["status"] == 0), mode
        n_dodged += int((host["iters"] > 80).sum()) if mode == "auto" else 0     # x0 start to its stationary point + two dodge passes
        same = dev["iters"] == host["iters"]
        assert same.mean() >= 0.8
        np.testing.assert_allclose(dev["xopt"][same], host["xopt"][same], rtol=0, atol=1e-9)
        np.testing.assert_allclose(dev["xopt"], host["xopt"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(dev["info"][:, 0], host["info"][:, 0], rtol=1e-6, atol=1e-9)      # the SAME side won
        if mode in ("auto", "multiwave"):                        # one-wavefront and four-wavefront kernels: identical words
            key = dev["xopt"].shape
            if key not in ref:
                ref[key] = dev
            else:
                for k in dev:
                    assert np.array_equal(ref[key][k], dev[k]), k
    assert n_dodged >= 3
    # without the rung those calls come back infeasible, on the GPU as on the host (two of the run's three with the default
    # position box this test solves them in)
    off = _solve_all(calls, ("auto",), dodge=False)
    assert all(np.array_equal(dev["status"], host["status"]) for _, host, dev in off) and sum(int((dev["status"] == 2).sum()) for _, _, dev in off) >= 2


def test_c5_calls_screened_alike_and_identical_to_the_host_core():
    calls = sum((_calls(world=w, N=5) for w in (0, 11, 40)), [])
    res = _solve_all(calls, ("auto", "multiwave", "lane"))
    n_scr = 0
    for mode, host, dev in res:
        assert np.array_equal(dev["status"], host["status"]), mode
        scr = host["iters"] == 0
        n_scr += int(scr.sum()) if mode == "auto" else 0
        assert np.array_equal(dev["iters"] == 0, scr)
        for k in ("xopt", "uopt", "ts_opt"):
            assert np.array_equal(dev[k][scr], host[k][scr]), k          # screened: x0 at every stage, zero inputs -- the same words
        np.testing.assert_allclose(dev["info"][scr], host["info"][scr], rtol=1e-9, atol=1e-15)      # (the shortfall holds cos(theta_0): device libm against the host's)
        ok = np.isin(host["status"], (0, 1))
        np.testing.assert_allclose(dev["xopt"][ok], host["xopt"][ok], rtol=0, atol=1e-5)
    assert n_scr >= 20
