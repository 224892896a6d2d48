"""The oracle's interior-point method reaches the independent known answers of SURVEY.md Appendix C
(analytic demo8 optimum; SciPy-converged demo1/N=6 and demo9/N=5; demo1/N=5 infeasible) and its output
carries a KKT certificate of the ORIGINAL NLP."""
import numpy as np
import pytest

from oracle import ipm_dense
from tests.test_oracle_nlp import build

KNOWN = {
    "demo8_N5_mpc4_step0": dict(T=20.0, f=3600.238),
    "demo1_N6_mpc4_step0": dict(T=20.378864, f=4334.19729,
                                x=[[3, 4.222732, 5.378339, 6.533895, 7.68938, 8.844713, 10],
                                   [4, 4, 4.399556, 4.79926, 5.19917, 5.59952, 6],
                                   [0, .332888, .333015, .333194, .333574, .333687, 0]]),
    "demo9_N5_mpc4_step0": dict(T=30.451762, f=7392.01551,
                                x=[[1, 2.827106, 3.870164, 4.913362, 5.956658, 7],
                                   [5, 5, 6.500115, 8.000133, 9.500082, 11],
                                   [0, .963219, .963125, .96306, .96303, 1.570796]]),
}


def by_name(golden, name):
    return [c for c in golden if c["name"] == name][0]


@pytest.mark.parametrize("name", sorted(KNOWN))
def test_known_answers(nlp_golden, name):
    p = build(by_name(nlp_golden, name))
    r = ipm_dense.solve(p)
    k = KNOWN[name]
    assert r.status == ipm_dense.STATUS_OK and r.feas
    assert r.Ts_opt == pytest.approx(k["T"] * p.Ts, abs=2e-6)
    assert r.f == pytest.approx(k["f"], abs=2e-3)
    if "x" in k:
        np.testing.assert_allclose(r.xopt, np.array(k["x"]), atol=2e-4)
    assert np.allclose(r.uopt[0], 0.6, atol=1e-6)                     # rides the speed bound (Appendix C)
    cert = ipm_dense.kkt_certificate(p, r)
    assert cert["primal"] < 1e-8 and cert["stationarity"] < 1e-5 and cert["complementarity"] < 1e-5


def test_demo8_analytic_trajectory(nlp_golden):
    p = build(by_name(nlp_golden, "demo8_N5_mpc4_step0"))
    r = ipm_dense.solve(p)
    exp = np.array([[3 + 1.2 * k for k in range(6)], [4.0] * 6, [0.0] * 6])
    np.testing.assert_allclose(r.xopt, exp, atol=1e-6)


def test_demo1_N5_is_infeasible(nlp_golden):
    """terminal reference pose (9,5,pi/4) puts the obstacle corner inside the footprint"""
    p = build(by_name(nlp_golden, "demo1_N5_mpc4_step0"))
    r = ipm_dense.solve(p)
    assert r.status == ipm_dense.STATUS_INFEASIBLE and not r.feas
    assert 1e-3 < r.elastic < 0.1


@pytest.mark.parametrize("name", ["demo1_dyn_mpc8", "slanted_asym_mpc6", "slanted_asym_mpc8", "slanted_asym_mpc4"])
def test_fixed_time_and_slanted_certificates(nlp_golden, name):
    p = build(by_name(nlp_golden, name))
    r = ipm_dense.solve(p)
    assert r.feas
    cert = ipm_dense.kkt_certificate(p, r)
    assert cert["primal"] < 1e-7 and cert["stationarity"] < 1e-5 and cert["complementarity"] < 1e-5
