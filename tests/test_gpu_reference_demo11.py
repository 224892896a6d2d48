"""The reference's demo11 run (tests/test_reference_demo11.py) through the PRODUCT path: the ``closedLoop`` mirror on the drop-in
``obca`` class, one GPU solve per step through the C ABI."""
import pytest

from tests import reference_report
from tests.test_reference_demo11 import check_run

pytestmark = pytest.mark.gpu


def test_product_path_shows_figure_11_and_lies_on_the_gif_markers():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
    s = obca()
    cum, cl = reference_report.replay(reference_report.demo11_setting(), s, 61)
    check_run(cum, cl.x_closed, reference_report.fixture(), reference_report.gif_demo11())
    assert cl.feas == True      # noqa: E712
