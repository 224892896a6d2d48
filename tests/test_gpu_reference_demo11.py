"""The reference's demo11 run (tests/test_reference_demo11.py) through the PRODUCT path: the ``closedLoop`` mirror on the drop-in
``obca`` class, one GPU solve per step through the C ABI."""
import pytest

from tests import reference_report
from tests.test_reference_demo11 import check_fourth_title_at_reading_precision, check_markers_at_reading_precision, check_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def run():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
    s = obca()
    cum, cl = reference_report.replay(reference_report.demo11_setting(), s, 61)
    return cum, cl


def test_product_path_shows_figure_11_and_lies_on_the_gif_markers(run):
    cum, cl = run
    check_run(cum, cl.x_closed, reference_report.fixture(), reference_report.gif_demo11())
    assert cl.feas == True      # noqa: E712


@pytest.mark.xfail(strict=True, reason="measured 0.0099 s against the 0.0055 s a title can be read to (tests/reference_report.py)")
def test_product_path_fourth_title_at_reading_precision(run):
    check_fourth_title_at_reading_precision(run[0], run[1].x_closed, reference_report.fixture(), reference_report.gif_demo11())


@pytest.mark.xfail(strict=True, reason="measured: markers of the dodge 0.16-0.33 m from their pose, accuracy 0.15 m (tests/reference_report.py)")
def test_product_path_markers_at_reading_precision(run):
    check_markers_at_reading_precision(run[0], run[1].x_closed, reference_report.fixture(), reference_report.gif_demo11())
