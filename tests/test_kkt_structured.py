"""The structured Newton-step solve (local LDL^T + soft-dynamics Riccati) equals the dense augmented solve."""
import json

import numpy as np
import pytest

from oracle import ipm_dense
from oracle.kkt_structured import structured_step
from tests.test_oracle_nlp import build


def compare(p, d, split=0):
    n, N = p.n, p.N
    lay = [p.eq_layout()[i] for i in d["term"]]          # elastic equality rows, dense order
    Je, E, gh, iseq = d["Je"], d["E"], d["ghat"], d["iseq"]
    nt = len(lay)
    soft = np.array([r[0] in ("init", "dyn") for r in lay] + [False] * (Je.shape[0] - nt))
    cond = ~soft
    H = d["W"] + d["delta_w"] * np.eye(n)
    if p.variant == 4:
        H[n - 1, n - 1] += d["delta_w"] * N
    H = H + Je[cond].T @ (Je[cond] / E[cond][:, None])
    b = d["r_x"] + Je[cond].T @ (gh[cond] / E[cond]) - Je[soft].T @ (0 * gh[soft])
    # the soft rows' multipliers are already inside r_x (J^T y); nothing else to add
    init_rows = [i for i, r in enumerate(lay) if r[0] == "init"]
    dyn_rows = [[i for i, r in enumerate(lay) if r[0] == "dyn" and r[1] == k] for k in range(N)]
    A = [-Je[rows][:, p.ip(k):p.ip(k) + 3] for k, rows in enumerate(dyn_rows)]
    B = [-Je[rows][:, p.iu(k):p.iu(k) + 2] for k, rows in enumerate(dyn_rows)]
    free_T = p.variant == 4
    tcol = [(-Je[rows][:, p.iT()] if free_T else np.zeros(3)) for rows in dyn_rows]
    dx, dnu, dyi, dyd, ok = structured_step(p, H, b, d["Jh"], d["ch"], A, B, tcol,
                                            [E[rows] for rows in dyn_rows], [gh[rows] for rows in dyn_rows],
                                            E[init_rows], gh[init_rows], free_T, split=split)
    assert ok
    sc = max(1.0, np.max(np.abs(d["dx"])))
    np.testing.assert_allclose(dx, d["dx"], rtol=0, atol=2e-7 * sc)
    scy = max(1.0, np.max(np.abs(d["dye"])), np.max(np.abs(d["dyh"])))
    np.testing.assert_allclose(dnu, d["dyh"], rtol=0, atol=2e-6 * scy)
    np.testing.assert_allclose(dyi, d["dye"][init_rows], rtol=0, atol=2e-6 * scy)
    for k in range(N):
        np.testing.assert_allclose(dyd[k], d["dye"][dyn_rows[k]], rtol=0, atol=2e-6 * scy)


@pytest.mark.parametrize("idx", [0, 1, 4, 5, 6])
def test_structured_equals_dense(nlp_golden, idx):
    p = build(nlp_golden[idx])
    seen = []

    def probe(d):
        if d["it"] in (0, 1, 5, 12, 25, 40) :
            seen.append(d["it"])
            compare(p, d)

    ipm_dense.solve(p, dict(probe=probe, max_iter=41, max_soc=0))
    assert len(seen) >= 5


@pytest.mark.parametrize("idx", [0, 1, 4, 5, 6])
def test_two_sided_sweep_equals_dense(nlp_golden, idx):
    """blueprint of the parallel Riccati: backward recursion from N to m, forward recursion (cost-to-arrive) from 0 to m,
    one 6x6 solve where they meet -- same step as the dense augmented solve for every split stage"""
    p = build(nlp_golden[idx])
    seen = []

    def probe(d):
        if d["it"] in (0, 5, 25):
            for m in range(1, p.N):
                compare(p, d, split=m)
            seen.append(d["it"])

    ipm_dense.solve(p, dict(probe=probe, max_iter=26, max_soc=0))
    assert len(seen) == 3
