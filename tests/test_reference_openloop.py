"""Solver-boundary pin on the SECOND reference-held output: the open-loop plan the reference repository shows in
images/aStar_vs_openLoopOBCA.png (README, report Figure 10) -- 51 poses of one CasADi/IPOPT solve of obca_mpc4 at N = 50 on
demo9 (src/simulation.py:114-123).  47 of its dots can be read to a pixel (0.091 m); this build's plan, default parameters, must
put one pose on each of them.  (The problem needs the second level of the penalty escalation: with rho <= 1e6 every start ends
with elastic variables ~3e-3 left -- csrc/obca_device.h: OBCA_RHO_ESCALATION.)"""
import numpy as np
import pytest

from tests import native_build, reference_openloop


@pytest.fixture(scope="module")
def fx():
    return reference_openloop.fixture()


def test_fixture_shape(fx):
    m = np.asarray(fx["markers_xy"])
    assert m.shape == (47, 2) and fx["markers_total"] == 51 and abs(fx["pixel_m"] - 0.0911) < 1e-3
    assert m[:, 0].min() > 1.0 and m[:, 0].max() < 37.0 and m[:, 1].min() > 5.0 and m[:, 1].max() < 58.0   # between start (1, 5) and goal (37, 58)


@pytest.mark.parametrize("order", ["x0", "window"])
def test_structured_core_puts_a_pose_on_every_marker_of_the_references_plan(fx, order):
    """engine: the structured core the kernels are built from, compiled for the host (the dense C oracle needs minutes per
    iteration at this size: 1988 variables).  Default ladder and window-first end at the same plan; 47 markers, 47 distinct
    poses in order, each within a pixel."""
    s = native_build.LpiObca()
    s.start_order = order
    cl = reference_openloop.plan(s)
    assert cl.feas and s.calls[-1]["status"] == 0
    assert abs(cl.Ts_opt - 2.57464) < 1e-4
    d, idx = reference_openloop.marker_distances(fx, cl.xOpt)
    assert d.max() <= reference_openloop.MARKER_TOL, d.max()
    order_along = np.argsort(idx)
    assert len(set(idx.tolist())) == 47 and np.all(np.diff(idx[order_along]) >= 1)
    assert set(range(51)) - set(idx.tolist()) == {0, 1, 49, 50}       # the dots the picture hides: under the start mark and the car box, under the goal mark and its box


def test_neighbouring_weights_move_the_plan_off_the_markers(fx):
    """the comparison has teeth: with the checked-in Q = 0.1 I instead of the figure's 0.5 I the same solve converges to a plan
    that misses the markers by up to 0.9 m"""
    s = native_build.LpiObca()
    cl = reference_openloop.plan(s, q=0.1)
    assert cl.feas
    d, _ = reference_openloop.marker_distances(fx, cl.xOpt)
    assert d.max() > 0.5 and d.mean() > 0.2
