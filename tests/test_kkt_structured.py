"""The structured Newton-step solve (local LDL^T + soft-dynamics Riccati) equals the dense augmented solve."""
import json

import numpy as np
import pytest

from oracle import ipm_dense
from oracle.kkt_structured import structured_step
from tests.test_oracle_nlp import build


def compare(p, d, split=0, kernel_check=None):
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
                                            E[init_rows], gh[init_rows], free_T, split=split, kernel_check=kernel_check)
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

    ipm_dense.solve(p, dict(probe=probe, max_iter=41, max_soc=0, single_start=True, start_order="zeros", dodge=False))
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

    ipm_dense.solve(p, dict(probe=probe, max_iter=26, max_soc=0, single_start=True, start_order="zeros", dodge=False))
    assert len(seen) == 3


def _LS(a, b):
    return a * (a + 1) // 2 + b if a >= b else b * (b + 1) // 2 + a


def _forward_stage_as_the_kernel_does_it(k, Pi, pi, F, G, Lxx, Lxu, Luu, lx, lu, D, gh, Zref, Pinext, pinext, worst):
    """numpy transcription of `riccati_forward_half` (csrc/obca_kernel.hip), lane by lane: lane (a, b) of 36 builds the 5x5
    block S_vv and the two columns a, b of [S_vw c_v] from the packed stage block, the cost-to-arrive and the dynamics row's
    coefficients (-e_c for dp'_c, B e_c for du_c, the T column, -ghat for the gradient), and produces entry (a, b) of
    Pi_{k+1}; lanes 36..41 the gradient; the lanes with a = 0 keep their column of Z"""
    L8 = np.zeros((8, 8))
    L8[:6, :6], L8[:6, 6:], L8[6:, :6], L8[6:, 6:] = Lxx, Lxu, Lxu.T, Luu
    Lk = np.zeros(36)
    for a in range(8):
        for b in range(a + 1):
            Lk[_LS(a, b)] = L8[a, b]
    lk = np.concatenate([lx, lu])
    Pif = Pi.reshape(-1)
    a02, a12, tc, hcs, hsn, h = F[0, 2], F[1, 2], F[:3, 5], G[0, 0], G[1, 0], G[2, 1]
    outP, outp, Zk = np.zeros((6, 6)), np.zeros(6), np.zeros((5, 7))
    for lane in range(42):
        a = lane // 6 if lane < 36 else lane - 36
        b = lane % 6 if lane < 36 else 6
        s = np.zeros(15)
        for j in range(5):
            for l in range(j + 1):
                s[j * (j + 1) // 2 + l] = Pif[6 * j + l] + Lk[j * (j + 1) // 2 + l]
        s[0] += D[0]; s[2] += D[1]; s[3] += a02 * D[0]; s[4] += a12 * D[1]
        s[5] += a02 * a02 * D[0] + a12 * a12 * D[1] + D[2]
        if k == 0:
            s[6:9] = 0; s[9] = 1; s[10:14] = 0; s[14] = 1
        Sm = np.zeros((5, 5))
        for j in range(5):
            for l in range(j + 1):
                Sm[j, l] = Sm[l, j] = s[j * (j + 1) // 2 + l]

        def column(c):
            zi = 3 + c if c < 5 else 5
            f = np.zeros(3)
            if c < 3: f[c] = -1
            elif c == 3: f[:] = [hcs, hsn, 0]
            elif c == 4: f[:] = [0, 0, h]
            elif c == 5: f[:] = tc
            else: f[:] = -gh
            r = np.zeros(5)
            for j in range(5):
                v = 0.0
                if c >= 3: v = lk[j] if c == 6 else Lk[zi * (zi + 1) // 2 + j]
                if c >= 5: v += pi[j] if c == 6 else Pif[30 + j]
                r[j] = v
            g = D * f
            r[0] += g[0]; r[1] += g[1]; r[2] += a02 * g[0] + a12 * g[1] + g[2]
            if k == 0: r[3] = r[4] = 0
            return f, r

        fb, rb = column(b)
        zb = np.linalg.solve(Sm, rb)
        fa, ra = column(a)
        out = np.sum(fa * D * fb)
        if a >= 3 and b >= 3:
            za, zb_i = (3 + a if a < 5 else 5), (3 + b if b < 5 else 5)
            if b < 6:
                out += Lk[_LS(za, zb_i)] + (Pif[35] if a == 5 and b == 5 else 0.0)
            else:
                out += lk[za] + (pi[5] if a == 5 else 0.0)
        out -= ra @ zb
        if lane < 36: outP[a, b] = out
        else: outp[a] = out
        if a == 0: Zk[:, b] = zb
    nv = Zref.shape[0]
    worst.append(max(np.abs(outP - Pinext).max() / max(1.0, np.abs(Pinext).max()),
                     np.abs(outp - pinext).max() / max(1.0, np.abs(pinext).max()), np.abs(Zk[:nv] - Zref).max()))


@pytest.mark.parametrize("idx", [0, 5, 6])
def test_kernel_form_of_the_forward_stage(nlp_golden, idx):
    """the per-lane expressions the HIP forward half evaluates reproduce the blueprint's forward stage (cost-to-arrive,
    gradient, recovery map) at every stage -- free-time, fixed-time with terminal set, fixed-time"""
    from functools import partial
    p = build(nlp_golden[idx])
    worst = []

    def probe(d):
        if d["it"] in (0, 3, 20):
            compare(p, d, split=p.N - 1, kernel_check=partial(_forward_stage_as_the_kernel_does_it, worst=worst))

    ipm_dense.solve(p, dict(probe=probe, max_iter=21, max_soc=0, single_start=True, start_order="zeros", dodge=False))
    assert len(worst) == 3 * (p.N - 1) and max(worst) < 1e-10
