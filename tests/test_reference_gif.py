"""Solver-boundary pin on a REFERENCE-HELD OUTPUT: the closed-loop run of demo9 the reference repository ships as a GIF
(CasADi/IPOPT solved it; fixture read off its frames by tests/golden/make_gif_fixture.py).

Frame k shows ``sum(Ts_opt[:k])`` to 0.01 s.  The mirror of ``closedLoop.closed_loop_mpc4`` driven by this build's solver
must show the same numbers: every step's ``Ts_opt`` feeds the next step's obstacle prediction, start pose and input
(src/closed_loop.py:349-432), so 47 chained solves -- 19 x obca_mpc4, the 11 x obca_mpc6 that dodge the moving box, 17 x
obca_mpc4 -- agreeing to the displayed digit means IPOPT and this solver returned the same optimum at every one of them."""
import numpy as np
import pytest

from tests import native_build, reference_gif


@pytest.fixture(scope="module")
def fx():
    return reference_gif.fixture()


def test_fixture_shape(fx):
    t = np.asarray(fx["spend_time"])
    assert len(t) == 84 and t[0] == 0.0 and t[-1] == 129.72
    assert np.all(np.diff(t) > 0)
    assert ["%.2f" % v for v in t] == fx["spend_time_text"]          # %.2f as src/draw.py:380 prints it


@pytest.mark.parametrize("engine", ["lpi", "oracle"])
def test_cpu_solvers_replay_the_reference_run(fx, engine):
    """engine "oracle": the dense C oracle (oracle/obca_oracle.c) -- this is what pins the ORACLE to the reference;
    engine "lpi": the structured core the kernels are built from, compiled for the host."""
    n = reference_gif.MATCHED_STEPS
    cum, xs, cl = reference_gif.replay(native_build.LpiObca(engine), n)
    assert len(cum) == n
    ref = np.asarray(fx["spend_time"][1:n + 1])
    err = np.abs(cum - ref)
    assert err.max() <= reference_gif.TIME_TOL, (int(err.argmax()) + 1, err.max())
    # the three phases of the run: free time, fixed time while the box is sensed (steps 20 .. 30), free time again
    variants = [c["variant"] for c in cl.obca_solver.calls]
    assert variants == [4] * 19 + [6] * 11 + [4] * 17
    assert all(c["status"] == 0 for c in cl.obca_solver.calls)
    # poses: every stand-alone marker of the GIF below the corner (y < 47 m) has a pose of this run on it
    m = np.asarray([p for p in fx["markers_xy"] if p[1] < 47.0])
    assert len(m) >= 25
    d = np.sqrt(((m[:, None, :] - xs[None, :, :2]) ** 2).sum(-1)).min(1)
    assert d.max() <= reference_gif.MARKER_TOL, d.max()


def test_where_the_runs_part(fx):
    """Documented, not hidden: at step 48 (pose (11.43, 48.75, 0.89), the left turn round the block corner at (13, 49)) the
    reference's IPOPT returned Ts_opt = 1.44 s, this solver another stationary point (2.01 s); from there on the closed loops
    differ.  The test keeps the figure honest: if a change moves the first differing step, MATCHED_STEPS must follow."""
    n = reference_gif.MATCHED_STEPS
    cum, _, _ = reference_gif.replay(native_build.LpiObca(), n + 1)
    ref = np.asarray(fx["spend_time"][1:n + 2])
    err = np.abs(cum - ref)
    assert err[:n].max() <= reference_gif.TIME_TOL
    assert err[n] > 0.1
