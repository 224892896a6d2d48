"""The reference's own demo9 closed loop (GIF fixture, tests/test_reference_gif.py) through the PRODUCT path: the
``closedLoop`` mirror on the drop-in ``obca`` class, one GPU solve per step through the C ABI."""
import numpy as np
import pytest

from tests import native_build, reference_gif

pytestmark = pytest.mark.gpu


def _replay(order, n):
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
    s = obca()
    s.start_order = order
    return reference_gif.replay(s, n)


@pytest.mark.parametrize("order", ["default", "x0", "window", "zeros"])
def test_product_path_shows_the_references_digits(order):
    """default ladder (the window first), x0 first and window first: 69 consecutive steps; the literal zero start first: 47"""
    fx = reference_gif.fixture()
    n = reference_gif.MATCHED[order]["gpu"]
    cum, xs, _ = _replay(order, n)
    assert len(cum) == n
    err = np.abs(cum - np.asarray(fx["spend_time"][1:n + 1]))
    assert err.max() <= reference_gif.TIME_TOL, (int(err.argmax()) + 1, err.max())
    if n >= 69:
        m = np.asarray([p for p in fx["markers_xy"] if p[1] < 53.3 and p[0] < 31.5])      # every stand-alone marker up to step 69
        assert len(m) >= 40
    else:
        m = np.asarray([p for p in fx["markers_xy"] if p[1] < 47.0])
    d = np.sqrt(((m[:, None, :] - xs[None, :, :2]) ** 2).sum(-1)).min(1)
    assert d.max() <= reference_gif.MARKER_TOL, d.max()


def test_default_run_reaches_the_goal_after_84_steps_like_the_references():
    fx = reference_gif.fixture()
    cum, xs, cl = _replay("default", 120)
    assert cl.goal_reached() and cl.k == 84 == fx["setting"]["frames"]
    ref = np.asarray(fx["spend_time"])
    assert np.abs(cum[:reference_gif.GIF_STEPS] - ref[1:]).max() < 0.5


@pytest.mark.parametrize("order", ["default", "x0", "window", "zeros"])
def test_gpu_replay_is_the_host_cores_replay(order):
    """same closed loop on the CPU build of the structured core: the chained steps agree to solver tolerance"""
    n = reference_gif.MATCHED[order]["gpu"]
    cum, xs, _ = _replay(order, n)
    s = native_build.LpiObca()
    s.start_order = order
    cum_h, xs_h, _ = reference_gif.replay(s, n)
    np.testing.assert_allclose(cum, cum_h, rtol=0, atol=1e-5)
    np.testing.assert_allclose(xs, xs_h, rtol=0, atol=1e-5)
