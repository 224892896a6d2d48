"""Pins the numpy NLP restatement (oracle/obca_nlp.py) to the reference's own model-building code.

tests/golden/nlp_eval.json holds, for several scenarios and random points, the objective value and
every constraint (lb, value, ub) the reference's obca_mpc4/6/8 hand to CasADi, in the reference's
order.  Here the restatement is evaluated at the same points and laid out in that order.
"""
import numpy as np
import pytest

from oracle.obca_nlp import Problem


def reference_order(p, z, pt):
    """(kind, lb, val, ub) scalars in the order the reference issues subject_to() calls."""
    c = p.eq(z)
    d = p.ineq(z)
    lb, ub = p.ineq_bounds()
    eqn = {row: c[i] for i, row in enumerate(p.eq_layout())}
    iqn = {row: (lb[i], d[i], ub[i]) for i, row in enumerate(p.ineq_layout())}
    N = p.N
    out = []
    for k in range(N):
        for j in range(3):
            out.append(("eq", 0.0, eqn[("dyn", k, j)], 0.0))
        if p.variant == 4:
            out.append(("eq", 0.0, 0.0, 0.0))          # Topt[k] == Topt[k+1], collapsed
    for j in range(2):
        for k in range(N + 1):
            out.append(("ineq",) + iqn[("xbnd", k, j)])
    for j in range(2):
        for k in range(N):
            out.append(("ineq",) + iqn[("ubnd", k, j)])
    for k in range(N):
        for j in range(2):
            out.append(("ineq",) + iqn[("acc", k, j)])
    for j in range(3):                                  # x[:,0] == x0 : value x, bounds x0
        out.append(("eq", p.x0[j], eqn[("init", 0, j)] + p.x0[j], p.x0[j]))
    if p.variant == 4:
        for j in range(3):
            r = p.xref[j, N]
            out.append(("eq", r, eqn[("term", N, j)] + r, r))
    if p.variant == 6:
        out.append(("ineq",) + iqn[("termx",)])
        out.append(("ineq",) + iqn[("termy",)])
    rowsA = len(pt["l"])
    for k in range(N + 1):
        for j in range(rowsA):
            j0 = 0 if p.variant == 4 else k * p.M      # mpc6/8 use rows k*M.. of column k (src/obca.py:1482)
            if j0 <= j < j0 + p.M:
                out.append(("ineq",) + iqn[("lam", k, j - j0)])
            else:                                       # dummy rows of l: only l >= 0, never used elsewhere
                out.append(("ineq", 0.0, pt["l"][j][k], np.inf))
        for j in range(4 * p.nObs):
            out.append(("ineq",) + iqn[("mu", k, j)])
        if p.variant == 4:
            out.append(("ineq",) + iqn[("Tpos", k)])
            out.append(("ineq",) + iqn[("Tbnd", k)])
    for k in range(N + 1):
        for i in range(p.nObs):
            out.append(("ineq",) + iqn[("norm", k, i)])
            out.append(("eq", 0.0, eqn[("rot", k, i, 0)], 0.0))
            out.append(("eq", 0.0, eqn[("rot", k, i, 1)], 0.0))
            out.append(("ineq",) + iqn[("dist", k, i)])
    return out


def build(case):
    a = case["inputs"]
    return Problem.from_reference_args(case["variant"], a["Ts"], a["P"], a["Q"], a["R"], a["N"], a["x0"], a["xL"],
                                       a["xU"], a["uL"], a["uU"], a["xref"], a["nObs"], a["vObs"], a["AObs"],
                                       a["bObs"], a["dmin"], a["ego"], a["u0"], a.get("uOpt"),
                                       a.get("terminal_set"))


def pack_point(p, pt):
    lam = np.asarray(pt["l"])
    if p.variant == 4:
        lam_eff = lam[:p.M]
    else:   # mpc6/8: l_ith = l[n:n+v-1, :] with n running across k  (src/obca.py:1482-1494)
        lam_eff = np.stack([lam[k * p.M:(k + 1) * p.M, k] for k in range(p.N + 1)], axis=1)
    T = pt["Topt"][0][0] if p.variant == 4 else None
    return p.pack(np.asarray(pt["x"]), np.asarray(pt["u"]), lam_eff, np.asarray(pt["mu"]), T)


def test_cases_present(nlp_golden):
    names = {c["name"] for c in nlp_golden}
    assert {"demo1_N6_mpc4_step0", "demo9_N5_mpc4_step0", "demo1_dyn_mpc6", "demo1_dyn_mpc8",
            "slanted_asym_mpc4", "slanted_asym_mpc6"} <= names


@pytest.mark.parametrize("idx", range(9))
def test_restatement_matches_reference_model(nlp_golden, idx):
    case = nlp_golden[idx]
    p = build(case)
    for pt in case["points"]:
        z = pack_point(p, pt["point"])
        assert p.objective(z) == pytest.approx(pt["objective"], rel=1e-12, abs=1e-12)
        flat = []
        for c in pt["cons"]:
            for lb, v, ub in zip(c["lb"], c["val"], c["ub"]):
                flat.append((c["kind"], lb, v, ub))
        mine = reference_order(p, z, pt["point"])
        assert len(mine) == len(flat)
        for i, (a, b) in enumerate(zip(mine, flat)):
            assert a[0] == b[0], (i, a, b)
            np.testing.assert_allclose(a[1:], b[1:], rtol=1e-12, atol=1e-12, err_msg="constraint %d %s" % (i, a))


@pytest.mark.parametrize("idx", [0, 4, 5, 7])
def test_derivatives_by_finite_differences(nlp_golden, idx):
    case = nlp_golden[idx]
    p = build(case)
    z = pack_point(p, case["points"][0]["point"])
    rng = np.random.default_rng(7)
    f, g, H = p.objective(z, grad=True, hess=True)
    c, Jc = p.eq(z, jac=True)
    d, Jd = p.ineq(z, jac=True)
    yc = rng.normal(size=c.size)
    yd = rng.normal(size=d.size)
    _, _, Hc = p.eq(z, jac=True, hess_y=yc)
    _, _, Hd = p.ineq(z, jac=True, hess_y=yd)
    eps = 1e-6
    gn = np.zeros_like(g)
    Jcn = np.zeros_like(Jc)
    Jdn = np.zeros_like(Jd)
    Hn = np.zeros_like(H)
    for i in range(p.n):
        e = np.zeros(p.n)
        e[i] = eps
        gn[i] = (p.objective(z + e) - p.objective(z - e)) / (2 * eps)
        Jcn[:, i] = (p.eq(z + e) - p.eq(z - e)) / (2 * eps)
        Jdn[:, i] = (p.ineq(z + e) - p.ineq(z - e)) / (2 * eps)
        gp = p.objective(z + e, grad=True)[1] + p.eq(z + e, jac=True)[1].T @ yc + p.ineq(z + e, jac=True)[1].T @ yd
        gm = p.objective(z - e, grad=True)[1] + p.eq(z - e, jac=True)[1].T @ yc + p.ineq(z - e, jac=True)[1].T @ yd
        Hn[:, i] = (gp - gm) / (2 * eps)
    np.testing.assert_allclose(g, gn, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(Jc, Jcn, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(Jd, Jdn, rtol=1e-6, atol=1e-6)
    Hl = H + Hc + Hd
    np.testing.assert_allclose(Hl, Hl.T, atol=1e-12)
    np.testing.assert_allclose(Hl, Hn, rtol=1e-5, atol=2e-5)
