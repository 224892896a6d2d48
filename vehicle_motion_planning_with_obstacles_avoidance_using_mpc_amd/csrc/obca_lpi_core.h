// obca_lpi_core.h -- lane-per-instance ("throughput") form of the OBCA interior-point solver.
//
// Same algorithm, same formulas and the same two-level structured Newton solve as csrc/obca_kernel.hip
// (wave-per-instance, LDS resident), but written as ordinary serial code for ONE instance whose working set
// lives in a strided workspace: element i of array `a` of instance `inst` is ws[(a.off + i) * stride + inst].
// On the GPU a 64-lane wavefront runs 64 instances in lock-step (divergent iteration counts are masked by the
// hardware), every lane does useful work, and every workspace access is a coalesced 512-byte transaction across
// the batch.  The helper functions below are the wave kernel's, with lane-strided loops turned into plain loops.
//
// The file also compiles as plain C++ (no HIP) so that the core can be exercised on the CPU by tests/.
#ifndef OBCA_LPI_CORE_H
#define OBCA_LPI_CORE_H

#include <math.h>
#include <stdint.h>
#include "obca_device.h"

#if defined(__HIPCC__)
#define LPI_FN __device__ __forceinline__
#define LPI_MEM __device__ __forceinline__
#define LPI_HD __host__ __device__ inline
#else
#define LPI_FN static inline
#define LPI_MEM inline
#define LPI_HD static inline
#endif

#if defined(LPI_TRACE) && !defined(__HIPCC__)
#include <stdio.h>
#define LPI_TRACE_LINE(...) printf(__VA_ARGS__)          /* host debugging aid: one line per accepted iteration */
#else
#define LPI_TRACE_LINE(...)
#endif

namespace lpi {

constexpr int MW = OBCA_MAX_EDGES + 6;
constexpr int NW = OBCA_MAX_EDGES + 4;
constexpr int FILT_MAX = 128;

struct SP {                      // strided pointer into the workspace
    double* p;
    size_t st;
    LPI_MEM double& operator[](int i) const { return p[(size_t)i * st]; }
    LPI_MEM SP operator+(int k) const { return SP{p + (size_t)k * st, st}; }
};

struct Lay {
    int N, nO, M, NS, n, free_T, variant;
    int r_init, r_dyn, r_term, r_xb, r_ub, r_acc, r_T, r_tx, r_norm, r_dist, r_lam, r_mu, R;
    int npair, R_cap;          // R_cap: row count of the shape's largest variant (decides the filter capacity)
    LPI_MEM int ip(int k) const { return k * NS; }
    LPI_MEM int iu(int k) const { return k * NS + 3; }
    LPI_MEM int il(int k) const { return k * NS + (k < N ? 5 : 3); }
    LPI_MEM int imu(int k) const { return il(k) + M; }
    LPI_MEM int iT() const { return n - 1; }
};

struct Inst {
    double x0[3], u0[2], Ts, Tmax, term[3];
    double xL[2], xU[2], uL[2], uU[2], gego[4], off, dmin;
    double Q[9], P[9], R1[4], R2[4];
};

struct Sh {
    SP x, xt, dx, gf, bx;
    SP y, Einv, yhat, gh, dy, s, p, n, zL, zU, zp, zn, g;
    SP ct, st, cc, ctt, stt, cct;
    SP nu, dnu, crot;
    SP Aobs, bobs, xref;
    SP Lall, lall, Y;
    SP Pk, qk, Kk, kapk, Mik;
    SP filt;
    SP gsoc, rest, dyo, dxo, dnuo, crot_t;    // second-order correction: corrected residuals, trial residuals, saved original direction
    const int* offm;
};

LPI_FN double wave_sum(double v) { return v; }      // one instance per lane: nothing to reduce
LPI_FN void lpi_sincos(double x, double* s, double* c) { *s = sin(x); *c = cos(x); }

// number of workspace doubles per instance and the carve-up (host and device use the same function)
struct Carve {
    int x, xt, dx, gf, bx, y, Einv, yhat, gh, dy, s, p, n, zL, zU, zp, zn, g, ct, st, cc, ctt, stt, cct, nu, dnu, crot,
        Aobs, bobs, xref, Lall, lall, Y, Pk, qk, Kk, kapk, Mik, filt, gsoc, rest, dyo, dxo, dnuo, crot_t, total;
};
LPI_HD Carve carve(int N, int nO, int M, int n_max, int R_max) {
    Carve c;
    int t = 0;
    const int N1 = N + 1, np = N1 * nO;
#define TK(name, cnt) c.name = t; t += (cnt);
    TK(x, n_max) TK(xt, n_max) TK(dx, n_max) TK(gf, n_max) TK(bx, n_max)
    TK(y, R_max) TK(Einv, R_max) TK(yhat, R_max) TK(gh, R_max) TK(dy, R_max)
    TK(s, R_max) TK(p, R_max) TK(n, R_max) TK(zL, R_max) TK(zU, R_max) TK(zp, R_max) TK(zn, R_max) TK(g, R_max)
    TK(ct, N1) TK(st, N1) TK(cc, 2 * np) TK(ctt, N1) TK(stt, N1) TK(cct, 2 * np)
    TK(nu, 2 * np) TK(dnu, 2 * np) TK(crot, 2 * np)
    TK(Aobs, N1 * M * 2) TK(bobs, N1 * M) TK(xref, 3 * N1)
    TK(Lall, 64 * N1) TK(lall, 8 * N1) TK(Y, MW * 4 * np)
    TK(Pk, 36 * N1) TK(qk, 6 * N1) TK(Kk, 12 * N1) TK(kapk, 2 * N1) TK(Mik, 9 * (N1 + 1))
    TK(filt, 2 * FILT_MAX)
    TK(gsoc, R_max) TK(rest, R_max) TK(dyo, R_max) TK(dxo, n_max) TK(dnuo, 2 * np) TK(crot_t, 2 * np)
#undef TK
    c.total = t;
    return c;
}

LPI_FN double dmaxabs(double a, double b) { return fmax(a, fabs(b)); }

// weight of a row in sums (the N+1 tied Topt copies are one row with multiplicity N+1)
LPI_FN double row_w(const Lay& L, int r) {
    return (L.free_T && r >= L.r_T && r < L.r_T + 2) ? (double)(L.N + 1) : 1.0;
}
LPI_FN bool row_iseq(const Lay& L, int r) { return r < L.r_xb; }   // init, dyn, term
LPI_FN bool row_soft(const Lay& L, int r) { return r < L.r_term; } // init, dyn (Riccati)


// trig + c = A^T lambda for an iterate held in xv; results into ct/st/cc
LPI_FN void eval_geom(const Lay& L, const Sh& S, SP xv, SP ct, SP st, SP cc,
                          int lane) {
    for (int k = lane; k <= L.N; k += 1) {
        double sn, cs;
        lpi_sincos(xv[L.ip(k) + 2], &sn, &cs);
        ct[k] = cs;
        st[k] = sn;
    }
    for (int pr = lane; pr < L.npair; pr += 1) {
        const int k = pr / L.nO, i = pr - k * L.nO;
        const int o0 = S.offm[i], o1 = S.offm[i + 1];
        const SP lam = xv + L.il(k);
        const SP A = S.Aobs + k * L.M * 2;
        double c0 = 0.0, c1 = 0.0;
        for (int j = o0; j < o1; ++j) {
            c0 += A[2 * j] * lam[j];
            c1 += A[2 * j + 1] * lam[j];
        }
        cc[2 * pr] = c0;
        cc[2 * pr + 1] = c1;
    }
}

// value of elastic row r at iterate xv (geometry arrays must be current)
LPI_FN double row_value(const Lay& L, const Sh& S, const Inst& in, SP xv, SP ct,
                            SP st, SP cc, int r) {
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts;
    if (r < L.r_dyn) return xv[r] - in.x0[r];
    if (r < L.r_term) {
        const int q = r - L.r_dyn, k = q / 3, j = q - 3 * k;
        const SP pk = xv + L.ip(k);
        const SP pn = xv + L.ip(k + 1);
        const SP u = xv + L.iu(k);
        const double f = (j == 0) ? u[0] * ct[k] : (j == 1) ? u[0] * st[k] : u[1];
        return pn[j] - pk[j] - h * f;
    }
    if (r < L.r_xb) {
        const int j = r - L.r_term;
        return xv[L.ip(L.N) + j] - S.xref[j * (L.N + 1) + L.N];
    }
    if (r < L.r_ub) {
        const int q = r - L.r_xb;
        return xv[L.ip(q >> 1) + (q & 1)];
    }
    if (r < L.r_acc) {
        const int q = r - L.r_ub;
        return xv[L.iu(q >> 1) + (q & 1)];
    }
    if (r < L.r_T) {
        const int q = r - L.r_acc, k = q >> 1, c = q & 1;
        const double prev = (k == 0) ? in.u0[c] : xv[L.iu(k - 1) + c];
        return (prev - xv[L.iu(k) + c]) / h;
    }
    if (r < L.r_tx) return T;
    if (r < L.r_norm) return xv[L.ip(L.N) + (r - L.r_tx)];
    if (r < L.r_dist) {
        const int pr = r - L.r_norm;
        return cc[2 * pr] * cc[2 * pr] + cc[2 * pr + 1] * cc[2 * pr + 1];
    }
    if (r < L.r_lam) {
        const int pr = r - L.r_dist, k = pr / L.nO, i = pr - k * L.nO;
        const SP pk = xv + L.ip(k);
        const double tx = pk[0] + ct[k] * in.off, ty = pk[1] + st[k] * in.off;
        const SP mu = xv + L.imu(k) + 4 * i;
        const SP lam = xv + L.il(k);
        const SP b = S.bobs + k * L.M;
        double v = -(in.gego[0] * mu[0] + in.gego[1] * mu[1] + in.gego[2] * mu[2] + in.gego[3] * mu[3]) +
                   tx * cc[2 * pr] + ty * cc[2 * pr + 1];
        for (int j = S.offm[i]; j < S.offm[i + 1]; ++j) v -= b[j] * lam[j];
        return v;
    }
    if (r < L.r_mu) {
        const int q = r - L.r_lam, k = q / L.M, j = q - k * L.M;
        return xv[L.il(k) + j];
    }
    {
        const int q = r - L.r_mu, w4 = 4 * L.nO, k = q / w4, j = q - k * w4;
        return xv[L.imu(k) + j];
    }
}

LPI_FN void row_bounds(const Lay& L, const Inst& in, int r, double& lo, double& up) {
    const double INF = INFINITY;
    if (r < L.r_xb) { lo = 0.0; up = 0.0; return; }
    if (r < L.r_ub) { const int j = (r - L.r_xb) & 1; lo = in.xL[j]; up = in.xU[j]; return; }
    if (r < L.r_acc) { const int j = (r - L.r_ub) & 1; lo = in.uL[j]; up = in.uU[j]; return; }
    if (r < L.r_T) { const int c = (r - L.r_acc) & 1; const double a = c ? OBCA_ACC_MAX1 : OBCA_ACC_MAX0; lo = -a; up = a; return; }
    if (r < L.r_tx) { if (r == L.r_T) { lo = 0.0; up = INF; } else { lo = OBCA_T_MIN; up = in.Tmax; } return; }
    if (r < L.r_norm) { if (r == L.r_tx) { lo = in.term[0]; up = INF; } else { lo = in.term[1]; up = in.term[2]; } return; }
    if (r < L.r_dist) { lo = -INF; up = 1.0; return; }
    if (r < L.r_lam) { lo = in.dmin; up = INF; return; }
    lo = 0.0; up = INF;
}

// rotation-equality residuals (hard rows) for the pair pr
LPI_FN void rot_value(const Lay& L, SP xv, SP ct, SP st,
                                          SP cc, int pr, double& e1, double& e2) {
    const int k = pr / L.nO, i = pr - k * L.nO;
    const SP mu = xv + L.imu(k) + 4 * i;
    const double c0 = cc[2 * pr], c1 = cc[2 * pr + 1];
    e1 = mu[0] - mu[2] + ct[k] * c0 + st[k] * c1;
    e2 = mu[1] - mu[3] - st[k] * c0 + ct[k] * c1;
}

// scaled objective sf*f (all lanes return the same value); optionally its gradient into S.gf
template <bool GRAD>
LPI_FN double eval_objective(const Lay& L, const Sh& S, const Inst& in, SP xv, double sf, int lane) {
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts;
    double part = 0.0, gT = 0.0;
    for (int k = lane; k <= L.N; k += 1) {
        const SP pk = xv + L.ip(k);
        const double* W = (k < L.N) ? in.Q : in.P;
        double e[3], We[3];
        for (int j = 0; j < 3; ++j) e[j] = pk[j] - S.xref[j * (L.N + 1) + k];
        for (int a = 0; a < 3; ++a) We[a] = W[3 * a] * e[0] + W[3 * a + 1] * e[1] + W[3 * a + 2] * e[2];
        part += e[0] * We[0] + e[1] * We[1] + e[2] * We[2];
        if (GRAD) for (int a = 0; a < 3; ++a) S.gf[L.ip(k) + a] = sf * 2.0 * We[a];
        if (k < L.N) {
            const SP u = xv + L.iu(k);
            const double r0 = in.R1[0] * u[0] + in.R1[1] * u[1], r1 = in.R1[2] * u[0] + in.R1[3] * u[1];
            part += u[0] * r0 + u[1] * r1;
            double g0 = 2.0 * r0, g1 = 2.0 * r1;
            if (k + 1 < L.N) {          // (u_{k+1}-u_k)' R2 (.) / h^2
                const SP un = xv + L.iu(k + 1);
                const double q0 = un[0] - u[0], q1 = un[1] - u[1];
                const double s0 = in.R2[0] * q0 + in.R2[1] * q1, s1 = in.R2[2] * q0 + in.R2[3] * q1;
                const double qq = q0 * s0 + q1 * s1;
                part += qq / (h * h);
                g0 -= 2.0 * s0 / (h * h);
                g1 -= 2.0 * s1 / (h * h);
                gT += -2.0 * qq / (h * h * T);
            }
            if (k >= 1) {
                const SP um = xv + L.iu(k - 1);
                const double q0 = u[0] - um[0], q1 = u[1] - um[1];
                g0 += 2.0 * (in.R2[0] * q0 + in.R2[1] * q1) / (h * h);
                g1 += 2.0 * (in.R2[2] * q0 + in.R2[3] * q1) / (h * h);
            }
            if (GRAD) { S.gf[L.iu(k)] = sf * g0; S.gf[L.iu(k) + 1] = sf * g1; }
        }
    }
    double f = wave_sum(part);
    if (L.free_T) {
        f += (L.N + 1) * (10.0 * T + T * T);
        if (GRAD) {
            gT = wave_sum(gT) + (L.N + 1) * (10.0 + 2.0 * T);
            if (lane == 0) S.gf[L.iT()] = sf * gT;
        }
    }
    if (GRAD) {
        for (int k = 0; k <= L.N; ++k) {
            const int w = L.M + 4 * L.nO;
            for (int j = lane; j < w; j += 1) S.gf[L.il(k) + j] = 0.0;
        }
        }
    return sf * f;
}

// out = gf + J^T ymul (+ rotation rows with nu): gradient of the Lagrangian w.r.t. x.
// One target entry per lane (gather form, deterministic).
LPI_FN void gather_grad(const Lay& L, const Sh& S, const Inst& in, SP ym, SP out, int lane) {
    const SP xv = S.x;
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts;
    // poses and inputs
    for (int t = lane; t < (L.N + 1) * 5; t += 1) {
        const int k = t / 5, j = t - 5 * k;
        if (j >= 3 && k == L.N) continue;
        const double cs = S.ct[k], sn = S.st[k];
        double v;
        if (j < 3) {
            v = S.gf[L.ip(k) + j];
            if (k == 0) v += ym[L.r_init + j];
            if (k >= 1) v += ym[L.r_dyn + 3 * (k - 1) + j];
            if (k < L.N) {
                const SP yd = ym + L.r_dyn + 3 * k;
                v -= yd[j];
                if (j == 2) {
                    const double vel = xv[L.iu(k)];
                    v -= h * vel * (-sn * yd[0] + cs * yd[1]);
                }
            }
            if (k == L.N && L.variant == 4) v += ym[L.r_term + j];
            if (j < 2) {
                v += ym[L.r_xb + 2 * k + j];
                if (k == L.N && L.variant == 6) v += ym[L.r_tx + j];
            }
            for (int i = 0; i < L.nO; ++i) {
                const int pr = k * L.nO + i;
                const double c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
                const double yd = ym[L.r_dist + pr];
                if (j == 0) v += yd * c0;
                else if (j == 1) v += yd * c1;
                else {
                    const double dth = -sn * c0 + cs * c1;
                    v += yd * in.off * dth + S.nu[2 * pr] * dth + S.nu[2 * pr + 1] * (-cs * c0 - sn * c1);
                }
            }
            out[L.ip(k) + j] = v;
        } else {
            const int c = j - 3;
            v = S.gf[L.iu(k) + c];
            const SP yd = ym + L.r_dyn + 3 * k;
            v -= (c == 0) ? h * (cs * yd[0] + sn * yd[1]) : h * yd[2];
            v += ym[L.r_ub + 2 * k + c];
            v -= ym[L.r_acc + 2 * k + c] / h;
            if (k + 1 < L.N) v += ym[L.r_acc + 2 * (k + 1) + c] / h;
            out[L.iu(k) + c] = v;
        }
    }
    // lambda
    for (int t = lane; t < (L.N + 1) * L.M; t += 1) {
        const int k = t / L.M, j = t - k * L.M;
        int i = 0;
        while (j >= S.offm[i + 1]) ++i;
        const int pr = k * L.nO + i;
        const double a0 = S.Aobs[(k * L.M + j) * 2], a1 = S.Aobs[(k * L.M + j) * 2 + 1];
        const double cs = S.ct[k], sn = S.st[k], c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
        const SP pk = xv + L.ip(k);
        const double tx = pk[0] + cs * in.off, ty = pk[1] + sn * in.off;
        double v = S.nu[2 * pr] * (cs * a0 + sn * a1) + S.nu[2 * pr + 1] * (-sn * a0 + cs * a1);
        v += ym[L.r_norm + pr] * 2.0 * (a0 * c0 + a1 * c1);
        v += ym[L.r_dist + pr] * (tx * a0 + ty * a1 - S.bobs[k * L.M + j]);
        v += ym[L.r_lam + t];
        out[L.il(k) + j] = v;
    }
    // mu
    for (int t = lane; t < (L.N + 1) * 4 * L.nO; t += 1) {
        const int w4 = 4 * L.nO, k = t / w4, q = t - k * w4, i = q >> 2, j = q & 3;
        const int pr = k * L.nO + i;
        const double sgn = (j < 2) ? 1.0 : -1.0;
        double v = sgn * S.nu[2 * pr + (j & 1)];
        v -= in.gego[j] * ym[L.r_dist + pr];
        v += ym[L.r_mu + t];
        out[L.imu(k) + q] = v;
    }
    // time scale
    if (L.free_T) {
        double part = 0.0;
        for (int k = lane; k < L.N; k += 1) {
            const SP u = xv + L.iu(k);
            const SP yd = ym + L.r_dyn + 3 * k;
            part -= in.Ts * (u[0] * S.ct[k] * yd[0] + u[0] * S.st[k] * yd[1] + u[1] * yd[2]);
            for (int c = 0; c < 2; ++c) {
                const double prev = (k == 0) ? in.u0[c] : xv[L.iu(k - 1) + c];
                part -= ym[L.r_acc + 2 * k + c] * (prev - u[c]) / (T * h);
            }
        }
        part = wave_sum(part);
        if (lane == 0) out[L.iT()] = S.gf[L.iT()] + part + (L.N + 1) * (ym[L.r_T] + ym[L.r_T + 1]);
    }
}

// J_r dx for the condensed rows (soft rows get their dy from the Riccati sweep)
LPI_FN double row_jdx(const Lay& L, const Sh& S, const Inst& in, int r) {
    const SP xv = S.x;
    const SP d = S.dx;
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double dT = L.free_T ? d[L.iT()] : 0.0;
    const double h = T * in.Ts;
    if (r < L.r_xb) return d[L.ip(L.N) + (r - L.r_term)];
    if (r < L.r_ub) { const int q = r - L.r_xb; return d[L.ip(q >> 1) + (q & 1)]; }
    if (r < L.r_acc) { const int q = r - L.r_ub; return d[L.iu(q >> 1) + (q & 1)]; }
    if (r < L.r_T) {
        const int q = r - L.r_acc, k = q >> 1, c = q & 1;
        const double prev = (k == 0) ? in.u0[c] : xv[L.iu(k - 1) + c];
        const double dprev = (k == 0) ? 0.0 : d[L.iu(k - 1) + c];
        const double qq = prev - xv[L.iu(k) + c];
        return (dprev - d[L.iu(k) + c]) / h - qq / (T * h) * dT;
    }
    if (r < L.r_tx) return dT;
    if (r < L.r_norm) return d[L.ip(L.N) + (r - L.r_tx)];
    if (r < L.r_lam) {
        const bool isn = r < L.r_dist;
        const int pr = isn ? r - L.r_norm : r - L.r_dist;
        const int k = pr / L.nO, i = pr - k * L.nO;
        const double c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
        const SP A = S.Aobs + k * L.M * 2;
        const SP dl = d + L.il(k);
        double s0 = 0.0, s1 = 0.0, sb = 0.0;
        for (int j = S.offm[i]; j < S.offm[i + 1]; ++j) {
            s0 += A[2 * j] * dl[j];
            s1 += A[2 * j + 1] * dl[j];
            sb += S.bobs[k * L.M + j] * dl[j];
        }
        if (isn) return 2.0 * (c0 * s0 + c1 * s1);
        const double cs = S.ct[k], sn = S.st[k];
        const SP pk = xv + L.ip(k);
        const SP dp = d + L.ip(k);
        const SP dm = d + L.imu(k) + 4 * i;
        const double tx = pk[0] + cs * in.off, ty = pk[1] + sn * in.off;
        return -(in.gego[0] * dm[0] + in.gego[1] * dm[1] + in.gego[2] * dm[2] + in.gego[3] * dm[3]) + tx * s0 +
               ty * s1 - sb + c0 * dp[0] + c1 * dp[1] + in.off * (-sn * c0 + cs * c1) * dp[2];
    }
    if (r < L.r_mu) { const int q = r - L.r_lam, k = q / L.M; return d[L.il(k) + (q - k * L.M)]; }
    { const int q = r - L.r_mu, w4 = 4 * L.nO, k = q / w4; return d[L.imu(k) + (q - k * w4)]; }
}


// linearisation of one row at the current iterate: inverse D's and residuals (IPOPT's Sigma + delta_w)
struct Lin { double iDs, iDp, iDn, rs, rp, rn; };
LPI_FN Lin row_lin(double lo, double up, bool eq, double s, double p, double n, double y,
                                       double zL, double zU, double zp, double zn, double mu, double rho, double dw) {
    const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
    double sig = 0.0, gs = 0.0;
    if (hasL) { const double isl = 1.0 / (s - lo); sig += zL * isl; gs -= mu * isl; }
    if (hasU) { const double isu = 1.0 / (up - s); sig += zU * isu; gs += mu * isu; }
    if (hasL && !hasU) gs += OBCA_KAPPA_D * mu;
    if (hasU && !hasL) gs -= OBCA_KAPPA_D * mu;
    const double ip = 1.0 / p, inn = 1.0 / n;
    Lin q;
    q.iDs = eq ? 0.0 : 1.0 / (sig + dw);
    q.iDp = 1.0 / (zp * ip + dw);
    q.iDn = 1.0 / (zn * inn + dw);
    q.rs = eq ? 0.0 : (-y + gs);
    q.rp = rho - y - mu * ip;
    q.rn = rho + y - mu * inn;
    return q;
}

// barrier terms of one row at (s, p, n): -mu*log(product of its distances) + linear damping; ONE log per row
LPI_FN double row_barrier(double lo, double up, bool eq, double s, double p, double n, double mu,
                                              double rho) {
    const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
    double prod = p * n, lin = rho * (p + n);
    if (hasL) { prod *= (s - lo); if (!hasU) lin += OBCA_KAPPA_D * mu * (s - lo); }
    if (hasU) { prod *= (up - s); if (!hasL) lin += OBCA_KAPPA_D * mu * (up - s); }
    return lin - mu * log(prod);
}


// Lall[k] is the 8x8 symmetric stage matrix over (dp(0:3), du_prev(3:5), dT(5), du(6:8)); lall[k] its gradient.
LPI_FN void assemble_stages(const Lay& L, const Sh& S, const Inst& in, double sf, double dw, int lane) {
    const SP xv = S.x;
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts, ih2 = 1.0 / (h * h);
    double HTT = 0.0;
    for (int k = lane; k <= L.N; k += 1) {
        SP H = S.Lall + 64 * k;
        SP lv = S.lall + 8 * k;
        for (int a = 0; a < 64; ++a) H[a] = 0.0;
        const double cs = S.ct[k], sn = S.st[k];
        const double* W = (k < L.N) ? in.Q : in.P;
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) H[8 * a + b] = sf * 2.0 * W[3 * a + b];
            H[8 * a + a] += dw;
            lv[a] = S.bx[L.ip(k) + a];
        }
        lv[3] = lv[4] = lv[5] = 0.0;
        for (int j = 0; j < 2; ++j) {                               // position box rows
            H[8 * j + j] += S.Einv[L.r_xb + 2 * k + j];
        }
        if (k == L.N && L.variant == 4) for (int j = 0; j < 3; ++j) H[8 * j + j] += S.Einv[L.r_term + j];
        if (k == L.N && L.variant == 6) for (int j = 0; j < 2; ++j) H[8 * j + j] += S.Einv[L.r_tx + j];
        double hth = 0.0;                                           // theta-theta Lagrangian curvature
        for (int i = 0; i < L.nO; ++i) {
            const int pr = k * L.nO + i;
            const double c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
            const double yd = S.y[L.r_dist + pr], Ei = S.Einv[L.r_dist + pr];
            const double gp[3] = {c0, c1, in.off * (-sn * c0 + cs * c1)};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) H[8 * a + b] += Ei * gp[a] * gp[b];
            hth += yd * in.off * (-cs * c0 - sn * c1);
            hth += S.nu[2 * pr] * (-cs * c0 - sn * c1) + S.nu[2 * pr + 1] * (sn * c0 - cs * c1);
        }
        if (k < L.N) {
            const SP u = xv + L.iu(k);
            const SP yd = S.y + L.r_dyn + 3 * k;
            hth += h * u[0] * (yd[0] * cs + yd[1] * sn);
            const double hpu = h * (yd[0] * sn - yd[1] * cs);       // theta - v
            H[8 * 2 + 6] += hpu;
            H[8 * 6 + 2] += hpu;
            for (int a = 0; a < 2; ++a) {
                for (int b = 0; b < 2; ++b) H[8 * (6 + a) + 6 + b] = sf * 2.0 * in.R1[2 * a + b];
                H[8 * (6 + a) + 6 + a] += dw + S.Einv[L.r_ub + 2 * k + a];
                lv[6 + a] = S.bx[L.iu(k) + a];
            }
            // acceleration cost couples u_k with u_{k-1} (stage k) and u_{k+1} (stage k+1)
            const int ncost = (k + 1 < L.N ? 1 : 0) + (k >= 1 ? 1 : 0);
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) H[8 * (6 + a) + 6 + b] += sf * 2.0 * in.R2[2 * a + b] * ih2 * ncost;
            if (k >= 1)
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        const double v = -sf * 2.0 * in.R2[2 * a + b] * ih2;
                        H[8 * (3 + a) + 6 + b] += v;
                        H[8 * (6 + b) + 3 + a] += v;
                    }
            // acceleration rows k (u_{k-1}, u_k) and k+1 (u_k, u_{k+1})
            for (int c = 0; c < 2; ++c) {
                const double Ek = S.Einv[L.r_acc + 2 * k + c];
                H[8 * (6 + c) + 6 + c] += Ek * ih2;
                if (k >= 1) {
                    H[8 * (3 + c) + 6 + c] -= Ek * ih2;
                    H[8 * (6 + c) + 3 + c] -= Ek * ih2;
                }
                if (k + 1 < L.N) H[8 * (6 + c) + 6 + c] += S.Einv[L.r_acc + 2 * (k + 1) + c] * ih2;
            }
            if (L.free_T) {
                // (theta,T), (u,T) and (T,T) entries
                double hpT = -in.Ts * u[0] * (-yd[0] * sn + yd[1] * cs);
                double huT[2] = {-in.Ts * (yd[0] * cs + yd[1] * sn), -in.Ts * yd[2]};
                for (int c = 0; c < 2; ++c) {
                    const double prev = (k == 0) ? in.u0[c] : xv[L.iu(k - 1) + c];
                    const double q = prev - u[c];
                    const double ya = S.y[L.r_acc + 2 * k + c], Ek = S.Einv[L.r_acc + 2 * k + c];
                    huT[c] += ya / (T * h) + Ek * q / (T * h * h);
                    HTT += ya * 2.0 * q / (T * T * h) + Ek * q * q / (T * T * h * h);
                    if (k + 1 < L.N) {
                        const double qn = u[c] - xv[L.iu(k + 1) + c];
                        const double yn = S.y[L.r_acc + 2 * (k + 1) + c], En = S.Einv[L.r_acc + 2 * (k + 1) + c];
                        huT[c] += -yn / (T * h) - En * qn / (T * h * h);
                    }
                }
                // acceleration cost cross terms
                double qa[2] = {0, 0}, qb[2] = {0, 0};
                if (k + 1 < L.N) { qa[0] = xv[L.iu(k + 1)] - u[0]; qa[1] = xv[L.iu(k + 1) + 1] - u[1]; }
                if (k >= 1) { qb[0] = u[0] - xv[L.iu(k - 1)]; qb[1] = u[1] - xv[L.iu(k - 1) + 1]; }
                for (int a = 0; a < 2; ++a) {
                    const double Ra = in.R2[2 * a] * qa[0] + in.R2[2 * a + 1] * qa[1];
                    const double Rb = in.R2[2 * a] * qb[0] + in.R2[2 * a + 1] * qb[1];
                    huT[a] += sf * 4.0 * ih2 / T * (Ra - Rb);
                }
                if (k + 1 < L.N) {
                    const double qq = qa[0] * (in.R2[0] * qa[0] + in.R2[1] * qa[1]) + qa[1] * (in.R2[2] * qa[0] + in.R2[3] * qa[1]);
                    HTT += sf * 6.0 * qq * ih2 / (T * T);
                }
                H[8 * 2 + 5] += hpT; H[8 * 5 + 2] += hpT;
                for (int c = 0; c < 2; ++c) { H[8 * (6 + c) + 5] += huT[c]; H[8 * 5 + 6 + c] += huT[c]; }
            }
        } else {
            lv[6] = lv[7] = 0.0;
            H[8 * 6 + 6] = 1.0; H[8 * 7 + 7] = 1.0;     // no input at the last stage
        }
        H[8 * 2 + 2] += hth;
    }
    if (L.free_T) {
        HTT = wave_sum(HTT);
        if (lane == 0) {
            const double w = (double)(L.N + 1);
            HTT += sf * 2.0 * w + dw * w + w * (S.Einv[L.r_T] + S.Einv[L.r_T + 1]);
            S.Lall[8 * 5 + 5] += HTT;
            S.lall[5] = S.bx[L.iT()];
        }
    } else if (lane == 0) {
        S.Lall[8 * 5 + 5] = 1.0;                       // dT pinned to zero in the fixed-time variants
    }
}


// (I + Ppp E)^-1 by LU without pivoting; pivots equal those of I + E^1/2 Ppp E^1/2
LPI_FN int inv3_ipe(const double* P, int ld, const double* E, double Mi[9]) {
    double A[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) { A[3 * a + b] = P[ld * a + b] * E[b] + (a == b ? 1.0 : 0.0); Mi[3 * a + b] = (a == b ? 1.0 : 0.0); }
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (!(A[3 * j + j] > 0.0)) { bad = 1; LPI_TRACE_LINE("    (I+PE) pivot %d = %.3e\n", j, A[3 * j + j]); }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i > j) {
                const double f = A[3 * i + j] / A[3 * j + j];
#pragma unroll
                for (int c = 0; c < 3; ++c) { A[3 * i + c] -= f * A[3 * j + c]; Mi[3 * i + c] -= f * Mi[3 * j + c]; }
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int j = 2 - jj;
        const double inv = 1.0 / A[3 * j + j];
#pragma unroll
        for (int c = 0; c < 3; ++c) Mi[3 * j + c] *= inv;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < j) {
                const double f = A[3 * i + j];
#pragma unroll
                for (int c = 0; c < 3; ++c) Mi[3 * i + c] -= f * Mi[3 * j + c];
            }
    }
    return bad;
}

// soft-min of V(p', o) = 1/2 [p';o]'P[p';o] + q'[p';o] against 1/2 (p'-phat)' E^-1 (p'-phat), entirely in
// registers and redundantly in every lane (no LDS round trip):  X = P~ (symmetric 6x6), qt = q~, Mi = (I+Ppp E)^-1
LPI_FN int soft_min_regs(SP Pl, SP ql, const double E[3], double X[36],
                                             double qt[6], double Mi[9]) {
    double P[36], q[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) P[i] = Pl[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) q[i] = ql[i];
    const int bad = inv3_ipe(P, 6, E, Mi);
    double MP[18];                              // Mi [Ppp Ppo]  (3 x 6)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) MP[6 * a + c] = Mi[3 * a] * P[c] + Mi[3 * a + 1] * P[6 + c] + Mi[3 * a + 2] * P[12 + c];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c) X[6 * a + c] = 0.5 * (MP[6 * a + c] + MP[6 * c + a]);
#pragma unroll
        for (int c = 3; c < 6; ++c) { X[6 * a + c] = MP[6 * a + c]; X[6 * c + a] = MP[6 * a + c]; }
    }
#pragma unroll
    for (int a = 3; a < 6; ++a)
#pragma unroll
        for (int c = 3; c < 6; ++c) {           // Poo - Pop E (M Ppo)
            double v = P[6 * a + c];
#pragma unroll
            for (int e = 0; e < 3; ++e) v -= P[6 * a + e] * E[e] * MP[6 * e + c];
            X[6 * a + c] = v;
        }
    double Mq[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) Mq[a] = Mi[3 * a] * q[0] + Mi[3 * a + 1] * q[1] + Mi[3 * a + 2] * q[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) qt[a] = Mq[a];
#pragma unroll
    for (int a = 3; a < 6; ++a) qt[a] = q[a] - (P[6 * a] * E[0] * Mq[0] + P[6 * a + 1] * E[1] * Mq[1] + P[6 * a + 2] * E[2] * Mq[2]);
    return bad;
}


// ---------------------------------------------------------------- level 1: local blocks (serial over the pairs)
// Same 10x10 quasi-definite block per (stage, obstacle) as the wave kernel; here one instance handles its
// pairs one after the other, all four right-hand sides at once, and folds each 3x3 Schur complement straight
// into the stage block.
LPI_FN int local_blocks(const Lay& L, const Sh& S, const Inst& in, double dw) {
    int bad = 0;
    for (int pr = 0; pr < L.npair; ++pr) {
        const int k = pr / L.nO, i = pr - k * L.nO;
        const int o0 = S.offm[i], m = S.offm[i + 1] - o0;
        const double cs = S.ct[k], sn = S.st[k];
        const double c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
        const SP pk = S.x + L.ip(k);
        const double tx = pk[0] + cs * in.off, ty = pk[1] + sn * in.off;
        const double yn = S.y[L.r_norm + pr], En = S.Einv[L.r_norm + pr];
        const double yd = S.y[L.r_dist + pr], Ed = S.Einv[L.r_dist + pr];
        const double nu1 = S.nu[2 * pr], nu2 = S.nu[2 * pr + 1];
        double a0[OBCA_MAX_EDGES], a1[OBCA_MAX_EDGES], gn[OBCA_MAX_EDGES], gd[NW];
        double K[MW * (MW + 1) / 2], Yv[MW][4], Gm[MW][3];
#define KP(a, b) K[((a) * ((a) + 1)) / 2 + (b)]
        const double dth = -sn * c0 + cs * c1;
#pragma unroll
        for (int j = 0; j < OBCA_MAX_EDGES; ++j) {
            const bool on = j < m;
            a0[j] = on ? S.Aobs[(k * L.M + o0 + j) * 2] : 0.0;
            a1[j] = on ? S.Aobs[(k * L.M + o0 + j) * 2 + 1] : 0.0;
            const double bj = on ? S.bobs[k * L.M + o0 + j] : 0.0;
            gn[j] = 2.0 * (a0[j] * c0 + a1[j] * c1);
            gd[j] = on ? (tx * a0[j] + ty * a1[j] - bj) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) gd[OBCA_MAX_EDGES + j] = -in.gego[j];
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int b = 0; b < NW; ++b)
                if (b <= a) {
                    double v = Ed * gd[a] * gd[b];
                    if (a < OBCA_MAX_EDGES) v += En * gn[a] * gn[b] + yn * 2.0 * (a0[a] * a0[b] + a1[a] * a1[b]);
                    KP(a, b) = v;
                }
#pragma unroll
        for (int j = 0; j < OBCA_MAX_EDGES; ++j) {
            const bool on = j < m;
            KP(j, j) += on ? (dw + S.Einv[L.r_lam + k * L.M + (on ? o0 + j : 0)]) : 1.0;
            Yv[j][3] = on ? -S.bx[L.il(k) + o0 + j] : 0.0;
            Gm[j][0] = Ed * gd[j] * c0 + yd * a0[j];
            Gm[j][1] = Ed * gd[j] * c1 + yd * a1[j];
            Gm[j][2] = Ed * gd[j] * in.off * dth + yd * in.off * (-sn * a0[j] + cs * a1[j]) +
                       nu1 * (-sn * a0[j] + cs * a1[j]) + nu2 * (-cs * a0[j] - sn * a1[j]);
            KP(NW, j) = cs * a0[j] + sn * a1[j];
            KP(NW + 1, j) = -sn * a0[j] + cs * a1[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = OBCA_MAX_EDGES + j;
            KP(a, a) += dw + S.Einv[L.r_mu + k * 4 * L.nO + 4 * i + j];
            Yv[a][3] = -S.bx[L.imu(k) + 4 * i + j];
            Gm[a][0] = Ed * gd[a] * c0;
            Gm[a][1] = Ed * gd[a] * c1;
            Gm[a][2] = Ed * gd[a] * in.off * dth;
            KP(NW, a) = (j == 0) ? 1.0 : (j == 2) ? -1.0 : 0.0;
            KP(NW + 1, a) = (j == 1) ? 1.0 : (j == 3) ? -1.0 : 0.0;
        }
        KP(NW, NW) = 0.0; KP(NW + 1, NW) = 0.0; KP(NW + 1, NW + 1) = 0.0;
        Gm[NW][0] = 0.0; Gm[NW][1] = 0.0; Gm[NW][2] = dth; Yv[NW][3] = -S.crot[2 * pr];
        Gm[NW + 1][0] = 0.0; Gm[NW + 1][1] = 0.0; Gm[NW + 1][2] = -cs * c0 - sn * c1; Yv[NW + 1][3] = -S.crot[2 * pr + 1];
#pragma unroll
        for (int a = 0; a < MW; ++a) { Yv[a][0] = Gm[a][0]; Yv[a][1] = Gm[a][1]; Yv[a][2] = Gm[a][2]; }
        double dinv[MW];
        int nneg = 0;
#pragma unroll
        for (int j = 0; j < MW; ++j) {
            const double d = KP(j, j);
            // inertia by COUNT (exactly 2 negative pivots), not by position -- see csrc/obca_kernel.hip
            if (d < 0.0) ++nneg; else if (!(d > 0.0)) bad = 1;
            if (j == MW - 1 && nneg != 2) { bad = 1; LPI_TRACE_LINE("    pair %d: %d negative pivots\n", pr, nneg); }
            dinv[j] = 1.0 / d;
#pragma unroll
            for (int a = 0; a < MW; ++a) {
                if (a > j) {
                    const double la = KP(a, j) * dinv[j];
#pragma unroll
                    for (int b = 0; b < MW; ++b)
                        if (b > j && b <= a) KP(a, b) -= la * KP(b, j);
                }
            }
#pragma unroll
            for (int a = 0; a < MW; ++a) {
                if (a > j) {
                    KP(a, j) *= dinv[j];
#pragma unroll
                    for (int c = 0; c < 4; ++c) Yv[a][c] -= KP(a, j) * Yv[j][c];
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < MW; ++jj) {
            const int j = MW - 1 - jj;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double v = Yv[j][c] * dinv[j];
#pragma unroll
                for (int a = 0; a < MW; ++a)
                    if (a > j) v -= KP(a, j) * Yv[a][c];
                Yv[j][c] = v;
            }
        }
#undef KP
        const SP Yo = S.Y + pr * (MW * 4);
#pragma unroll
        for (int a = 0; a < MW; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) Yo[4 * a + c] = Yv[a][c];
        // Schur complement G'Y folded into the stage block
        double So[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double v = 0.0;
#pragma unroll
                for (int e = 0; e < MW; ++e) v += Gm[e][a] * Yv[e][c];
                So[a][c] = v;
            }
        const SP H = S.Lall + 64 * k;
        const SP lv = S.lall + 8 * k;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) H[8 * a + b] -= 0.5 * (So[a][b] + So[b][a]);
            lv[a] += So[a][3];
        }
    }
    return bad;
}

// ---------------------------------------------------------------- level 2: Riccati sweep + forward pass (serial)
LPI_FN int riccati(const Lay& L, const Sh& S, const Inst& in) {
    const SP xv = S.x;
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts;
    int bad = 0;
    {
        const SP PN = S.Pk + 36 * L.N;
        const SP qN = S.qk + 6 * L.N;
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) PN[6 * a + b] = (a < 3 && b < 3) ? S.Lall[64 * L.N + 8 * a + b] : 0.0;
            qN[a] = (a < 3) ? S.lall[8 * L.N + a] : 0.0;
        }
    }
    for (int k = L.N - 1; k >= 0; --k) {
        double E[3], gh[3], X[36], qt[6], Mi[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) { E[j] = 1.0 / S.Einv[L.r_dyn + 3 * k + j]; gh[j] = S.gh[L.r_dyn + 3 * k + j]; }
        bad |= soft_min_regs(S.Pk + 36 * (k + 1), S.qk + 6 * (k + 1), E, X, qt, Mi);
#pragma unroll
        for (int c = 0; c < 9; ++c) S.Mik[9 * k + c] = Mi[c];
        const double cs = S.ct[k], sn = S.st[k];
        const double u0 = xv[L.iu(k)], u1 = xv[L.iu(k) + 1];
        // [F G]: column 2 = e2 + (a0,a1,0), column 5 = e5 + (t0,t1,t2), column 6 = e3 + (b0,b1,0), column 7 = e4 + h e2
        const double a0 = -h * u0 * sn, a1 = h * u0 * cs;
        const double t0 = L.free_T ? in.Ts * u0 * cs : 0.0, t1 = L.free_T ? in.Ts * u0 * sn : 0.0, t2 = L.free_T ? in.Ts * u1 : 0.0;
        const double b0 = h * cs, b1 = h * sn;
        double Z[6][8], Mall[8][8], zv[6], mall[8];
#pragma unroll
        for (int r = 0; r < 6; ++r) {               // Z = X [F G]
            const double* Xr = X + 6 * r;
            Z[r][0] = Xr[0]; Z[r][1] = Xr[1];
            Z[r][2] = Xr[2] + a0 * Xr[0] + a1 * Xr[1];
            Z[r][3] = 0.0; Z[r][4] = 0.0;
            Z[r][5] = Xr[5] + t0 * Xr[0] + t1 * Xr[1] + t2 * Xr[2];
            Z[r][6] = Xr[3] + b0 * Xr[0] + b1 * Xr[1];
            Z[r][7] = Xr[4] + h * Xr[2];
            zv[r] = qt[r] - (Xr[0] * gh[0] + Xr[1] * gh[1] + Xr[2] * gh[2]);      // X f + qt, f = (-ghat, 0, 0, 0)
        }
        const SP Lk = S.Lall + 64 * k;
        const SP lk = S.lall + 8 * k;
#pragma unroll
        for (int b = 0; b < 8; ++b) {               // Mall = Lall + [F G]' Z
            Mall[0][b] = Lk[b] + Z[0][b];
            Mall[1][b] = Lk[8 + b] + Z[1][b];
            Mall[2][b] = Lk[16 + b] + Z[2][b] + a0 * Z[0][b] + a1 * Z[1][b];
            Mall[3][b] = Lk[24 + b];
            Mall[4][b] = Lk[32 + b];
            Mall[5][b] = Lk[40 + b] + Z[5][b] + t0 * Z[0][b] + t1 * Z[1][b] + t2 * Z[2][b];
            Mall[6][b] = Lk[48 + b] + Z[3][b] + b0 * Z[0][b] + b1 * Z[1][b];
            Mall[7][b] = Lk[56 + b] + Z[4][b] + h * Z[2][b];
        }
        mall[0] = lk[0] + zv[0]; mall[1] = lk[1] + zv[1];
        mall[2] = lk[2] + zv[2] + a0 * zv[0] + a1 * zv[1];
        mall[3] = lk[3]; mall[4] = lk[4];
        mall[5] = lk[5] + zv[5] + t0 * zv[0] + t1 * zv[1] + t2 * zv[2];
        mall[6] = lk[6] + zv[3] + b0 * zv[0] + b1 * zv[1];
        mall[7] = lk[7] + zv[4] + h * zv[2];
        const double m00 = Mall[6][6], m01 = 0.5 * (Mall[6][7] + Mall[7][6]), m11 = Mall[7][7];
        const double d1 = m11 - m01 * m01 / m00;
        if (!(m00 > 0.0) || !(d1 > 0.0)) { bad = 1; LPI_TRACE_LINE("    input block stage %d: m00 %.3e d1 %.3e\n", k, m00, d1); }
        const double idet = 1.0 / (m00 * d1);
        const double i00 = m11 * idet, i01 = -m01 * idet, i11 = m00 * idet;
        double xu[6][2], Kg[2][6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            xu[a][0] = 0.5 * (Mall[a][6] + Mall[6][a]);
            xu[a][1] = 0.5 * (Mall[a][7] + Mall[7][a]);
            Kg[0][a] = -(i00 * xu[a][0] + i01 * xu[a][1]);
            Kg[1][a] = -(i01 * xu[a][0] + i11 * xu[a][1]);
            S.Kk[12 * k + a] = Kg[0][a];
            S.Kk[12 * k + 6 + a] = Kg[1][a];
        }
        const double k0 = -(i00 * mall[6] + i01 * mall[7]), k1 = -(i01 * mall[6] + i11 * mall[7]);
        S.kapk[2 * k] = k0; S.kapk[2 * k + 1] = k1;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = 0; b < 6; ++b)
                S.Pk[36 * k + 6 * a + b] = 0.5 * (Mall[a][b] + Mall[b][a]) + xu[a][0] * Kg[0][b] + xu[a][1] * Kg[1][b];
            S.qk[6 * k + a] = mall[a] + xu[a][0] * k0 + xu[a][1] * k1;
        }
    }
    double E0[3], g0[3], X[36], qt[6], Mi0[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) { E0[j] = 1.0 / S.Einv[L.r_init + j]; g0[j] = S.gh[L.r_init + j]; }
    bad |= soft_min_regs(S.Pk, S.qk, E0, X, qt, Mi0);
    if (L.free_T && !(X[35] > 0.0)) { bad = 1; LPI_TRACE_LINE("    T pivot %.3e\n", X[35]); }
    if (bad) return 1;
    // forward pass
    double dT = 0.0;
    if (L.free_T) dT = -(qt[5] - (X[30] * g0[0] + X[31] * g0[1] + X[32] * g0[2])) / X[35];
    double dp[3], up[2] = {0.0, 0.0};
    {
        const SP P0 = S.Pk;
        const SP q0 = S.qk;
        double t[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a] = -g0[a] - E0[a] * (P0[6 * a + 5] * dT + q0[a]);
#pragma unroll
        for (int a = 0; a < 3; ++a) dp[a] = Mi0[a] * t[0] + Mi0[3 + a] * t[1] + Mi0[6 + a] * t[2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            S.dy[L.r_init + a] = -(P0[6 * a] * dp[0] + P0[6 * a + 1] * dp[1] + P0[6 * a + 2] * dp[2] + P0[6 * a + 5] * dT + q0[a]);
    }
    S.dx[0] = dp[0]; S.dx[1] = dp[1]; S.dx[2] = dp[2];
    if (L.free_T) S.dx[L.iT()] = dT;
    for (int k = 0; k < L.N; ++k) {
        const SP Kg = S.Kk + 12 * k;
        const double xi[6] = {dp[0], dp[1], dp[2], up[0], up[1], dT};
        double u[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double v = S.kapk[2 * k + c];
#pragma unroll
            for (int a = 0; a < 6; ++a) v += Kg[6 * c + a] * xi[a];
            u[c] = v;
        }
        const double cs = S.ct[k], sn = S.st[k];
        const SP uk = xv + L.iu(k);
        double ph[3];
        ph[0] = dp[0] - h * uk[0] * sn * dp[2] + h * cs * u[0] - S.gh[L.r_dyn + 3 * k];
        ph[1] = dp[1] + h * uk[0] * cs * dp[2] + h * sn * u[0] - S.gh[L.r_dyn + 3 * k + 1];
        ph[2] = dp[2] + h * u[1] - S.gh[L.r_dyn + 3 * k + 2];
        if (L.free_T) { ph[0] += in.Ts * uk[0] * cs * dT; ph[1] += in.Ts * uk[0] * sn * dT; ph[2] += in.Ts * uk[1] * dT; }
        const SP P1 = S.Pk + 36 * (k + 1);
        const SP q1 = S.qk + 6 * (k + 1);
        const SP Mi = S.Mik + 9 * k;
        double t[3], dn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double E = 1.0 / S.Einv[L.r_dyn + 3 * k + a];
            t[a] = ph[a] - E * (P1[6 * a + 3] * u[0] + P1[6 * a + 4] * u[1] + P1[6 * a + 5] * dT + q1[a]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) dn[a] = Mi[a] * t[0] + Mi[3 + a] * t[1] + Mi[6 + a] * t[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            S.dy[L.r_dyn + 3 * k + a] = -(P1[6 * a] * dn[0] + P1[6 * a + 1] * dn[1] + P1[6 * a + 2] * dn[2] +
                                          P1[6 * a + 3] * u[0] + P1[6 * a + 4] * u[1] + P1[6 * a + 5] * dT + q1[a]);
            S.dx[L.ip(k + 1) + a] = dn[a];
        }
        S.dx[L.iu(k)] = u[0];
        S.dx[L.iu(k) + 1] = u[1];
        dp[0] = dn[0]; dp[1] = dn[1]; dp[2] = dn[2];
        up[0] = u[0]; up[1] = u[1];
    }
    // local recovery: [dw; dnu] = Y_r - Y_G dp_k
    for (int pr = 0; pr < L.npair; ++pr) {
        const int k = pr / L.nO, i = pr - k * L.nO;
        const int o0 = S.offm[i], m = S.offm[i + 1] - o0;
        const SP Yo = S.Y + pr * (MW * 4);
        const SP d = S.dx + L.ip(k);
        const double d0 = d[0], d1 = d[1], d2 = d[2];
        for (int a = 0; a < MW; ++a) {
            const double v = Yo[4 * a + 3] - (Yo[4 * a] * d0 + Yo[4 * a + 1] * d1 + Yo[4 * a + 2] * d2);
            if (a < OBCA_MAX_EDGES) { if (a < m) S.dx[L.il(k) + o0 + a] = v; }
            else if (a < NW) S.dx[L.imu(k) + 4 * i + (a - OBCA_MAX_EDGES)] = v;
            else S.dnu[2 * pr + (a - NW)] = v;
        }
    }
    return 0;
}

// ---------------------------------------------------------------- optimality error (IPOPT eq. (5)/(6))
struct Err { double E, dual, prim, comp; };

LPI_FN Err ipm_errors(const Lay& L, const Sh& S, const Inst& in, double mu, double rho, double rxmax, double crotmax,
                      double nusum) {
    double dual = rxmax, prim = crotmax, comp = 0.0, ysum = nusum, zsum = 0.0, nz = 0.0, nrow = 2.0 * L.npair;
    for (int r = 0; r < L.R; ++r) {
        const double w = row_w(L, r);
        const bool eq = row_iseq(L, r);
        double lo, up;
        row_bounds(L, in, r, lo, up);
        const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
        const double s = S.s[r], y = S.y[r], p = S.p[r], n = S.n[r];
        const double zL = hasL ? S.zL[r] : 0.0, zU = hasU ? S.zU[r] : 0.0, zp = S.zp[r], zn = S.zn[r];
        if (!eq) dual = dmaxabs(dual, -y - zL + zU);
        dual = dmaxabs(dual, rho - y - zp);
        dual = dmaxabs(dual, rho + y - zn);
        prim = dmaxabs(prim, S.g[r] - (eq ? 0.0 : s) - p + n);
        comp = dmaxabs(comp, p * zp - mu);
        comp = dmaxabs(comp, n * zn - mu);
        if (hasL) comp = dmaxabs(comp, (s - lo) * zL - mu);
        if (hasU) comp = dmaxabs(comp, (up - s) * zU - mu);
        ysum += w * fabs(y);
        zsum += w * (zL + zU + zp + zn);
        nz += w * ((hasL ? 1.0 : 0.0) + (hasU ? 1.0 : 0.0) + 2.0);
        nrow += w;
    }
    const double sd = fmax(OBCA_S_MAX, (ysum + zsum) / (nrow + nz)) / OBCA_S_MAX;
    const double sc = fmax(OBCA_S_MAX, zsum / nz) / OBCA_S_MAX;
    Err e;
    e.dual = dual; e.prim = prim; e.comp = comp;
    e.E = fmax(fmax(dual / sd, prim), comp / sc);
    return e;
}

LPI_FN double row_barrier_lpi(double lo, double up, bool eq, double s, double p, double n, double mu, double rho) {
    const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
    double prod = p * n, lin = rho * (p + n);
    if (hasL) { prod *= (s - lo); if (!hasU) lin += OBCA_KAPPA_D * mu * (s - lo); }
    if (hasU) { prod *= (up - s); if (!hasL) lin += OBCA_KAPPA_D * mu * (up - s); }
    return lin - mu * log(prod);
}

// ---------------------------------------------------------------- one instance, start to finish
struct Out { int status, iters, nfact; double f, elastic, E0, ts_opt, sf; };

// The dodge starts (csrc/obca_device.h: OBCA_KIND_DODGE_R / _L; oracle/ipm_dense.py:dodge_start): the window moved sideways by
// side * OBCA_DODGE_OFFSET (side = -1: to the right of the direction of travel, +1: to the left), ramped in over the first
// OBCA_DODGE_RAMP stages; headings along the moved poses (continued from x0's, no jump by 2 pi), inputs by differences clipped to
// their box, the free-time scale as the window start's; lambda, mu of every (stage, obstacle) pair: the half-space row with the
// largest gap to the car at that pose, lambda = 1 / ||A_j|| on it, mu from the rotation equalities G'mu = -R A'lambda.
LPI_FN void dodge_start(const Lay& L, const Sh& S, const Inst& in, double side) {
    const int N1 = L.N + 1;
    for (int k = 0; k <= L.N; ++k) {
        double px = (k == 0) ? in.x0[0] : S.xref[0 * N1 + k], py = (k == 0) ? in.x0[1] : S.xref[1 * N1 + k];
        if (k > 0) {
            const int ka = k - 1, kb = k + 1 <= L.N ? k + 1 : L.N;
            const double ax = S.xref[0 * N1 + kb] - ((ka == 0) ? in.x0[0] : S.xref[0 * N1 + ka]);
            const double ay = S.xref[1 * N1 + kb] - ((ka == 0) ? in.x0[1] : S.xref[1 * N1 + ka]);
            const double len = sqrt(ax * ax + ay * ay);
            const double th = S.xref[2 * N1 + k];
            const double nx = len > 1e-9 ? -ay / len : -sin(th), ny = len > 1e-9 ? ax / len : cos(th);
            const double w = side * OBCA_DODGE_OFFSET * (k < OBCA_DODGE_RAMP ? (double)k / OBCA_DODGE_RAMP : 1.0);
            px += w * nx; py += w * ny;
        }
        S.x[L.ip(k)] = px; S.x[L.ip(k) + 1] = py;
    }
    S.x[L.ip(0) + 2] = in.x0[2];
    for (int k = 1; k <= L.N; ++k) {
        const double prev = S.x[L.ip(k - 1) + 2];
        double th = prev;
        if (k < L.N) {
            const double ddx = S.x[L.ip(k + 1)] - S.x[L.ip(k)], ddy = S.x[L.ip(k + 1) + 1] - S.x[L.ip(k) + 1];
            if (ddx * ddx + ddy * ddy > 1e-18) {
                double d = atan2(ddy, ddx) - prev;
                d -= 6.283185307179586 * floor(d / 6.283185307179586 + 0.5);      // nearest representative to the previous heading
                th = prev + d;
            }
        }
        S.x[L.ip(k) + 2] = th;
    }
    if (L.free_T) {
        double len = 0.0;
        for (int k = 0; k < L.N; ++k) {
            const double ddx = S.x[L.ip(k + 1)] - S.x[L.ip(k)], ddy = S.x[L.ip(k + 1) + 1] - S.x[L.ip(k) + 1];
            len += sqrt(ddx * ddx + ddy * ddy);
        }
        S.x[L.iT()] = fmin(fmax(1.0, len / (L.N * OBCA_WINDOW_SPEED_FRAC * in.uU[0] * in.Ts)), fmax(1.0, in.Tmax));
    }
    const double h = in.Ts * (L.free_T ? S.x[L.iT()] : 1.0);
    for (int k = 0; k < L.N; ++k) {
        const double ddx = S.x[L.ip(k + 1)] - S.x[L.ip(k)], ddy = S.x[L.ip(k + 1) + 1] - S.x[L.ip(k) + 1];
        const double dth = S.x[L.ip(k + 1) + 2] - S.x[L.ip(k) + 2];
        S.x[L.iu(k)] = fmin(fmax(sqrt(ddx * ddx + ddy * ddy) / h, in.uL[0]), in.uU[0]);
        S.x[L.iu(k) + 1] = fmin(fmax(dth / h, in.uL[1]), in.uU[1]);
    }
    for (int k = 0; k <= L.N; ++k) {
        const double th = S.x[L.ip(k) + 2], ct = cos(th), st = sin(th);
        const double tx = S.x[L.ip(k)] + ct * in.off, ty = S.x[L.ip(k) + 1] + st * in.off;
        for (int i = 0; i < L.nO; ++i) {
            const int o0 = S.offm[i], o1 = S.offm[i + 1];
            int jb = o0;
            double gb = -INFINITY, mub[4] = {0.0, 0.0, 0.0, 0.0}, lb = 0.0;
            for (int j = o0; j < o1; ++j) {
                const double a0 = S.Aobs[(k * L.M + j) * 2], a1 = S.Aobs[(k * L.M + j) * 2 + 1];
                const double nrm = sqrt(a0 * a0 + a1 * a1);
                if (!(nrm > 0.0)) continue;
                const double v0 = a0 / nrm, v1 = a1 / nrm;
                const double r0 = ct * v0 + st * v1, r1 = -st * v0 + ct * v1;
                const double m0 = fmax(-r0, 0.0), m1 = fmax(-r1, 0.0), m2 = fmax(r0, 0.0), m3 = fmax(r1, 0.0);
                const double gap = -(in.gego[0] * m0 + in.gego[1] * m1 + in.gego[2] * m2 + in.gego[3] * m3) + (a0 * tx + a1 * ty - S.bobs[k * L.M + j]) / nrm;
                if (gap > gb) { gb = gap; jb = j; lb = 1.0 / nrm; mub[0] = m0; mub[1] = m1; mub[2] = m2; mub[3] = m3; }
            }
            for (int j = o0; j < o1; ++j) S.x[L.il(k) + j] = (j == jb) ? lb : 0.0;
            for (int q = 0; q < 4; ++q) S.x[L.imu(k) + 4 * i + q] = mub[q];
        }
    }
}

// zwarm != nullptr: start from that primal vector moved one stage forward (last stage repeated; obca_set_warm_start); otherwise
// start `kind` of the ladder (csrc/obca_device.h: OBCA_KIND_*; the window as a trajectory: oracle/ipm_dense.py:window_start).
// mu0: the barrier parameter the pass begins with; iter_cap: its iteration limit beside max_iter_* (the caller's patience)
LPI_FN Out solve_instance(const Lay& L, const Sh& S, const Inst& in, const ObcaOptsDev& O,
                          const double* zwarm, double mu0, int kind, int iter_cap) {
    const int max_iter_v = L.free_T ? O.max_iter_free : O.max_iter_fixed;
    const int max_iter = max_iter_v < iter_cap ? max_iter_v : iter_cap;
    const double acc_tol = L.free_T ? 1e-6 : 1e-8;
    const double acc_objchg = L.free_T ? 1e20 : 1e-6;
    const bool from_window = kind == OBCA_KIND_WINDOW;
    if (zwarm) {
        const int blk = L.NS - 2;
        for (int k = 0; k <= L.N; ++k) {
            const int ks = k < L.N ? k + 1 : L.N;
            for (int q = 0; q < blk; ++q) {
                const int dst = (q < 3 ? L.ip(k) + q : L.il(k) + (q - 3));
                const int src = (q < 3 ? L.ip(ks) + q : L.il(ks) + (q - 3));
                S.x[dst] = zwarm[src];
            }
        }
        for (int k = 0; k < L.N; ++k)
            for (int j = 0; j < 2; ++j) S.x[L.iu(k) + j] = zwarm[L.iu(k + 1 < L.N ? k + 1 : L.N - 1) + j];
        if (L.free_T) S.x[L.iT()] = zwarm[L.iT()];
    } else {
        for (int t = 0; t < L.n; ++t) S.x[t] = 0.0;
        if (L.free_T) S.x[L.iT()] = 1.0;
        // x0 start: every pose at x0 -- where IPOPT's first full Newton step lands from the all-zero start (the dynamics
        // linearised at v = 0 read x_{k+1} = x_k, the initial condition x_0 = x0)
        if (kind == OBCA_KIND_X0) for (int k = 0; k <= L.N; ++k) for (int j = 0; j < 3; ++j) S.x[L.ip(k) + j] = in.x0[j];
        if (kind == OBCA_KIND_DODGE_R || kind == OBCA_KIND_DODGE_L) dodge_start(L, S, in, kind == OBCA_KIND_DODGE_R ? -1.0 : 1.0);
        if (from_window) {              // poses of the reference window, inputs by differences
            const int N1 = L.N + 1;
            for (int k = 0; k <= L.N; ++k)
                for (int j = 0; j < 3; ++j) S.x[L.ip(k) + j] = (k == 0) ? in.x0[j] : S.xref[j * N1 + k];
            if (L.free_T) {             // time scale at which the window is driven at OBCA_WINDOW_SPEED_FRAC of the speed bound
                double len = 0.0;
                for (int k = 0; k < L.N; ++k) {
                    const double ddx = S.x[L.ip(k + 1)] - S.x[L.ip(k)], ddy = S.x[L.ip(k + 1) + 1] - S.x[L.ip(k) + 1];
                    len += sqrt(ddx * ddx + ddy * ddy);
                }
                S.x[L.iT()] = fmin(fmax(1.0, len / (L.N * OBCA_WINDOW_SPEED_FRAC * in.uU[0] * in.Ts)), fmax(1.0, in.Tmax));
            }
            const double h = in.Ts * (L.free_T ? S.x[L.iT()] : 1.0);
            for (int k = 0; k < L.N; ++k) {
                const double ddx = S.x[L.ip(k + 1)] - S.x[L.ip(k)], ddy = S.x[L.ip(k + 1) + 1] - S.x[L.ip(k) + 1];
                const double dth = S.x[L.ip(k + 1) + 2] - S.x[L.ip(k) + 2];
                S.x[L.iu(k)] = fmin(fmax(sqrt(ddx * ddx + ddy * ddy) / h, in.uL[0]), in.uU[0]);
                S.x[L.iu(k) + 1] = fmin(fmax(dth / h, in.uL[1]), in.uU[1]);
            }
        }
    }
    for (int t = 0; t < 2 * L.npair; ++t) S.nu[t] = 0.0;
    Out o;
    o.status = OBCA_STATUS_MAXITER; o.iters = 0; o.nfact = 0; o.E0 = INFINITY; o.elastic = 0.0;
    double sf = 1.0, rho = O.rho;
    eval_geom(L, S, S.x, S.ct, S.st, S.cc, 0);
    double f = eval_objective<true>(L, S, in, S.x, 1.0, 0);
    {
        double gm = 0.0;
        for (int t = 0; t < L.n; ++t) gm = dmaxabs(gm, S.gf[t]);
        gm = fmax(gm, O.rho);
        sf = (gm > OBCA_MAX_GRADIENT) ? OBCA_MAX_GRADIENT / gm : 1.0;
        rho = O.rho * sf;
    }
    f = eval_objective<true>(L, S, in, S.x, sf, 0);
    double mu = mu0;
    bool bad_bounds = false;
    for (int r = 0; r < L.R; ++r) {
        double lo, up;
        row_bounds(L, in, r, lo, up);
        const bool eq = row_iseq(L, r);
        const double g = row_value(L, S, in, S.x, S.ct, S.st, S.cc, r);
        double s = g;
        const bool hasL = lo > -INFINITY, hasU = up < INFINITY;
        if (eq) s = 0.0;
        else if (hasL && hasU) {
            if (!(lo < up)) bad_bounds = true;
            const double pL = fmin(OBCA_BOUND_PUSH * fmax(1.0, fabs(lo)), OBCA_BOUND_FRAC * (up - lo));
            const double pU = fmin(OBCA_BOUND_PUSH * fmax(1.0, fabs(up)), OBCA_BOUND_FRAC * (up - lo));
            s = fmin(fmax(s, lo + pL), up - pU);
        } else if (hasL) s = fmax(s, lo + OBCA_BOUND_PUSH * fmax(1.0, fabs(lo)));
        else if (hasU) s = fmin(s, up - OBCA_BOUND_PUSH * fmax(1.0, fabs(up)));
        const double rr = g - s;
        const double a = (mu - rho * rr) / (2.0 * rho);
        const double en = a + sqrt(a * a + mu * rr / (2.0 * rho));
        const double ep = rr + en;
        S.g[r] = g; S.s[r] = s; S.p[r] = ep; S.n[r] = en;
        S.zp[r] = mu / ep; S.zn[r] = mu / en; S.y[r] = rho - mu / ep;
        S.zL[r] = (!eq && hasL) ? 1.0 : 0.0;
        S.zU[r] = (!eq && hasU) ? 1.0 : 0.0;
    }
    for (int pr = 0; pr < L.npair; ++pr) {
        double e1, e2;
        rot_value(L, S.x, S.ct, S.st, S.cc, pr, e1, e2);
        S.crot[2 * pr] = e1; S.crot[2 * pr + 1] = e2;
    }
    int nfilt = 0;
    const int filt_cap = OBCA_FILTER_CAP(L.R_cap);
    double theta_max = 0.0, theta_min = 0.0, delta_w_last = 0.0, tau = fmax(OBCA_TAU_MIN, 1.0 - mu);
    int acc_count = 0, it = 0;
    double fobj_prev = 0.0;
    bool have_prev = false;
    if (bad_bounds) o.status = OBCA_STATUS_BAD_BOUNDS;
    else
    for (it = 0; it <= max_iter; ++it) {
        gather_grad(L, S, in, S.y, S.bx, 0);
        double rxmax = 0.0, crotmax = 0.0, nusum = 0.0, th = 0.0, pnsum = 0.0, emax = 0.0;
        for (int t = 0; t < L.n; ++t) rxmax = dmaxabs(rxmax, S.bx[t]);
        for (int t = 0; t < 2 * L.npair; ++t) { crotmax = dmaxabs(crotmax, S.crot[t]); nusum += fabs(S.nu[t]); th += fabs(S.crot[t]); }
        for (int r = 0; r < L.R; ++r) {
            const double w = row_w(L, r), p = S.p[r], n = S.n[r];
            th += w * fabs(S.g[r] - (row_iseq(L, r) ? 0.0 : S.s[r]) - p + n);
            pnsum += w * (p + n);
            emax = fmax(emax, p + n);
        }
        o.elastic = emax;
        const Err e0 = ipm_errors(L, S, in, 0.0, rho, rxmax, crotmax, nusum);
        o.E0 = e0.E;
        if (it == 0) {
            theta_max = OBCA_THETA_MAX_FACT * fmax(1.0, th);
            theta_min = OBCA_THETA_MIN_FACT * fmax(1.0, th);
        }
        if (e0.E <= O.tol && e0.dual <= 1.0 && e0.prim <= 1e-4 && e0.comp <= 1e-4) { o.status = OBCA_STATUS_OK; break; }
        const double fobj = f + rho * pnsum;
        const double objchg = have_prev ? fabs(fobj - fobj_prev) / fmax(1.0, fabs(fobj)) : INFINITY;
        if (e0.E <= acc_tol && e0.dual <= 1e10 && e0.prim <= 1e-2 && e0.comp <= 1e-2 && objchg <= acc_objchg) {
            if (++acc_count >= OBCA_ACCEPTABLE_ITER) { o.status = OBCA_STATUS_ACCEPTABLE; break; }
        } else acc_count = 0;
        if (it == max_iter) break;
        {
            const double mu_floor = O.tol / (OBCA_KAPPA_EPS + 1.0);
            while (mu > mu_floor) {
                const Err em = ipm_errors(L, S, in, mu, rho, rxmax, crotmax, nusum);
                if (em.E > OBCA_KAPPA_EPS * mu) break;
                mu = fmax(mu_floor, fmin(OBCA_KAPPA_MU * mu, mu * sqrt(mu)));
                tau = fmax(OBCA_TAU_MIN, 1.0 - mu);
                nfilt = 0;
            }
        }
        double delta_w = 0.0;
        bool first_try = true;
        int fail = 0;
        for (;;) {
            for (int r = 0; r < L.R; ++r) {
                const bool eq = row_iseq(L, r);
                double lo, up;
                row_bounds(L, in, r, lo, up);
                const double y = S.y[r], s = S.s[r], p = S.p[r], n = S.n[r];
                const Lin q = row_lin(lo, up, eq, s, p, n, y, S.zL[r], S.zU[r], S.zp[r], S.zn[r], mu, rho, delta_w);
                const double rg = S.g[r] - (eq ? 0.0 : s) - p + n;
                const double gh = rg + q.rs * q.iDs + q.rp * q.iDp - q.rn * q.iDn;
                const double Ei = 1.0 / (q.iDs + q.iDp + q.iDn);
                S.Einv[r] = Ei;
                S.gh[r] = gh;
                S.yhat[r] = row_soft(L, r) ? y : (y + gh * Ei);
            }
            gather_grad(L, S, in, S.yhat, S.bx, 0);
            assemble_stages(L, S, in, sf, delta_w, 0);
            int bad = local_blocks(L, S, in, delta_w);
            if (!bad) bad = riccati(L, S, in);
            ++o.nfact;
            if (!bad) break;
            if (first_try) {
                delta_w = (delta_w_last == 0.0) ? OBCA_DELTA_W_0 : fmax(OBCA_DELTA_W_MIN, OBCA_KAPPA_W_MINUS * delta_w_last);
                first_try = false;
            } else {
                delta_w *= (delta_w_last == 0.0) ? OBCA_KAPPA_W_PLUS_BAR : OBCA_KAPPA_W_PLUS;
            }
            if (delta_w > OBCA_DELTA_W_MAX) { fail = 1; break; }
        }
        if (fail) { o.status = OBCA_STATUS_NUMERIC; break; }
        if (delta_w > 0.0) delta_w_last = delta_w;
        double a_max = 1.0, a_z = 1.0, dphi = 0.0, phi = f;
        for (int r = 0; r < L.R; ++r) {
            const bool eq = row_iseq(L, r);
            double lo, up;
            row_bounds(L, in, r, lo, up);
            const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
            const double dy = row_soft(L, r) ? S.dy[r] : (row_jdx(L, S, in, r) + S.gh[r]) * S.Einv[r];
            S.dy[r] = dy;
            const double w = row_w(L, r);
            const double s = S.s[r], p = S.p[r], n = S.n[r], y = S.y[r];
            const double zL = S.zL[r], zU = S.zU[r], zp = S.zp[r], zn = S.zn[r];
            const Lin q = row_lin(lo, up, eq, s, p, n, y, zL, zU, zp, zn, mu, rho, delta_w);
            const double ds = (dy - q.rs) * q.iDs, dp = (dy - q.rp) * q.iDp, dn = (-dy - q.rn) * q.iDn;
            const double gs = eq ? 0.0 : q.rs + y;
            if (hasL) {
                const double sl = s - lo;
                if (ds < 0.0) a_max = fmin(a_max, -tau * sl / ds);
                const double dz = (mu - zL * ds) / sl - zL;
                if (dz < 0.0) a_z = fmin(a_z, -tau * zL / dz);
            }
            if (hasU) {
                const double su = up - s;
                if (ds > 0.0) a_max = fmin(a_max, tau * su / ds);
                const double dz = (mu + zU * ds) / su - zU;
                if (dz < 0.0) a_z = fmin(a_z, -tau * zU / dz);
            }
            if (dp < 0.0) a_max = fmin(a_max, -tau * p / dp);
            if (dn < 0.0) a_max = fmin(a_max, -tau * n / dn);
            const double dzp = (mu - zp * dp) / p - zp, dzn = (mu - zn * dn) / n - zn;
            if (dzp < 0.0) a_z = fmin(a_z, -tau * zp / dzp);
            if (dzn < 0.0) a_z = fmin(a_z, -tau * zn / dzn);
            phi += w * row_barrier_lpi(lo, up, eq, s, p, n, mu, rho);
            dphi += w * (gs * ds + (rho - mu / p) * dp + (rho - mu / n) * dn);
        }
        for (int t = 0; t < L.n; ++t) dphi += S.gf[t] * S.dx[t];
        double alpha_min;
        if (dphi < 0.0) {
            double c = fmin(OBCA_GAMMA_THETA, OBCA_GAMMA_PHI * th / (-dphi));
            if (th <= theta_min) c = fmin(c, OBCA_DELTA * pow(th, OBCA_S_THETA) / pow(-dphi, OBCA_S_PHI));
            alpha_min = OBCA_GAMMA_ALPHA * c;
        } else alpha_min = OBCA_GAMMA_ALPHA * OBCA_GAMMA_THETA;
        // backtracking filter line search with IPOPT's second-order correction: when the FIRST trial step is rejected and
        // did not reduce the constraint violation, up to max_soc corrected steps are tried (same matrix, i.e. the same
        // delta_w; right-hand side from the accumulated residuals c_soc / g_soc; own fraction-to-boundary step a_soc;
        // acceptance tested with the ORIGINAL alpha) before the step length is halved.  a_try / S.dx / S.dy hold the
        // direction being tried -- the corrected one while use_soc -- and the original one waits in dxo / dyo / dnuo.
        double alpha = a_max, a_try = a_max, f_t = f, th_old = 0.0;
        bool accepted = false, aug = false, first_trial = true, use_soc = false;
        int soc_it = 0;
        for (;;) {
            for (int t = 0; t < L.n; ++t) S.xt[t] = S.x[t] + a_try * S.dx[t];
            eval_geom(L, S, S.xt, S.ctt, S.stt, S.cct, 0);
            f_t = eval_objective<false>(L, S, in, S.xt, sf, 0);
            double th_t = 0.0, phi_t = f_t;
            for (int r = 0; r < L.R; ++r) {
                const bool eq = row_iseq(L, r);
                double lo, up;
                row_bounds(L, in, r, lo, up);
                const double dy = S.dy[r], w = row_w(L, r);
                const double s = S.s[r], p = S.p[r], n = S.n[r];
                const Lin q = row_lin(lo, up, eq, s, p, n, S.y[r], S.zL[r], S.zU[r], S.zp[r], S.zn[r], mu, rho, delta_w);
                const double st = eq ? 0.0 : s + a_try * (dy - q.rs) * q.iDs;
                const double pt = p + a_try * (dy - q.rp) * q.iDp;
                const double nt = n + a_try * (-dy - q.rn) * q.iDn;
                const double gt = row_value(L, S, in, S.xt, S.ctt, S.stt, S.cct, r);
                const double res = gt - st - pt + nt;
                S.rest[r] = res;
                th_t += w * fabs(res);
                phi_t += w * row_barrier_lpi(lo, up, eq, st, pt, nt, mu, rho);
            }
            for (int pr = 0; pr < L.npair; ++pr) {
                double e1, e2;
                rot_value(L, S.xt, S.ctt, S.stt, S.cct, pr, e1, e2);
                th_t += fabs(e1) + fabs(e2);
                S.crot_t[2 * pr] = e1; S.crot_t[2 * pr + 1] = e2;
            }
            bool ok = false;
            aug = false;
            const bool finite = isfinite(th_t) && isfinite(f_t);
            bool blocked = !(th_t < theta_max) || !isfinite(phi_t);
            for (int i = 0; i < nfilt && !blocked; ++i)
                if (th_t >= S.filt[2 * i] && phi_t >= S.filt[2 * i + 1]) blocked = true;
            if (!blocked) {
                const bool switching = dphi < 0.0 && alpha * pow(-dphi, OBCA_S_PHI) > OBCA_DELTA * pow(th, OBCA_S_THETA);
                if (th <= theta_min && switching) {
                    ok = phi_t <= phi + OBCA_ETA_PHI * alpha * dphi + 10.0 * 2.220446049250313e-16 * fabs(phi);
                } else {
                    ok = (th_t <= (1.0 - OBCA_GAMMA_THETA) * th) || (phi_t <= phi - OBCA_GAMMA_PHI * th);
                    aug = ok;
                }
            }
            if (ok) { accepted = true; break; }
            bool solve_soc = false;
            if (use_soc) {
                if (!finite || th_t > OBCA_KAPPA_SOC * th_old || soc_it >= O.max_soc) {
                    for (int t = 0; t < L.n; ++t) S.dx[t] = S.dxo[t];          // back to the original direction
                    for (int t = 0; t < 2 * L.npair; ++t) S.dnu[t] = S.dnuo[t];
                    for (int r = 0; r < L.R; ++r) S.dy[r] = S.dyo[r];
                    use_soc = false;
                } else {
                    th_old = th_t;
                    for (int t = 0; t < 2 * L.npair; ++t) S.crot[t] = a_try * S.crot[t] + S.crot_t[t];
                    for (int r = 0; r < L.R; ++r) S.gsoc[r] = a_try * S.gsoc[r] + S.rest[r];
                    solve_soc = true;
                }
            } else if (first_trial && O.max_soc > 0 && finite && th_t >= th) {
                for (int t = 0; t < L.n; ++t) S.dxo[t] = S.dx[t];
                for (int t = 0; t < 2 * L.npair; ++t) { S.dnuo[t] = S.dnu[t]; S.crot[t] = alpha * S.crot[t] + S.crot_t[t]; }
                for (int r = 0; r < L.R; ++r) {
                    S.dyo[r] = S.dy[r];
                    S.gsoc[r] = alpha * (S.g[r] - (row_iseq(L, r) ? 0.0 : S.s[r]) - S.p[r] + S.n[r]) + S.rest[r];
                }
                th_old = th_t;
                use_soc = true;
                solve_soc = true;
            }
            first_trial = false;
            if (solve_soc) {
                ++soc_it;
                for (int r = 0; r < L.R; ++r) {
                    const bool eq = row_iseq(L, r);
                    double lo, up;
                    row_bounds(L, in, r, lo, up);
                    const double y = S.y[r];
                    const Lin q = row_lin(lo, up, eq, S.s[r], S.p[r], S.n[r], y, S.zL[r], S.zU[r], S.zp[r], S.zn[r], mu, rho, delta_w);
                    const double gh = S.gsoc[r] + q.rs * q.iDs + q.rp * q.iDp - q.rn * q.iDn;
                    S.gh[r] = gh;
                    S.yhat[r] = row_soft(L, r) ? y : (y + gh * S.Einv[r]);
                }
                gather_grad(L, S, in, S.yhat, S.bx, 0);
                assemble_stages(L, S, in, sf, delta_w, 0);
                (void)local_blocks(L, S, in, delta_w);            // same matrix as the accepted factorisation: same pivots
                (void)riccati(L, S, in);
                ++o.nfact;
                a_try = 1.0;
                for (int r = 0; r < L.R; ++r) {
                    const bool eq = row_iseq(L, r);
                    double lo, up;
                    row_bounds(L, in, r, lo, up);
                    const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
                    const double dy = row_soft(L, r) ? S.dy[r] : (row_jdx(L, S, in, r) + S.gh[r]) * S.Einv[r];
                    S.dy[r] = dy;
                    const double s = S.s[r], p = S.p[r], n = S.n[r];
                    const Lin q = row_lin(lo, up, eq, s, p, n, S.y[r], S.zL[r], S.zU[r], S.zp[r], S.zn[r], mu, rho, delta_w);
                    const double ds = (dy - q.rs) * q.iDs, dp = (dy - q.rp) * q.iDp, dn = (-dy - q.rn) * q.iDn;
                    if (hasL && ds < 0.0) a_try = fmin(a_try, -tau * (s - lo) / ds);
                    if (hasU && ds > 0.0) a_try = fmin(a_try, tau * (up - s) / ds);
                    if (dp < 0.0) a_try = fmin(a_try, -tau * p / dp);
                    if (dn < 0.0) a_try = fmin(a_try, -tau * n / dn);
                }
                continue;
            }
            alpha *= 0.5;
            a_try = alpha;
            if (alpha < alpha_min) break;
        }
        LPI_TRACE_LINE("it %3d th %.2e phi %.6e dphi %.2e mu %.1e dw %.1e a %.2e nfact %d nfilt %d\n", it, th, phi, dphi, mu,
                       delta_w, accepted ? a_try : -1.0, o.nfact, nfilt);
        if (!accepted) { o.status = OBCA_STATUS_LINESEARCH; break; }
        if (aug) {
            const double tn = (1.0 - OBCA_GAMMA_THETA) * th, pn = phi - OBCA_GAMMA_PHI * th;
            int w = 0;
            for (int i = 0; i < nfilt; ++i) {
                const double a = S.filt[2 * i], b = S.filt[2 * i + 1];
                if (!(a >= tn && b >= pn)) { S.filt[2 * w] = a; S.filt[2 * w + 1] = b; ++w; }
            }
            nfilt = w;
            if (nfilt >= filt_cap) { o.status = OBCA_STATUS_NUMERIC; break; }
            S.filt[2 * nfilt] = tn; S.filt[2 * nfilt + 1] = pn; ++nfilt;
        }
        for (int r = 0; r < L.R; ++r) {
            const bool eq = row_iseq(L, r);
            double lo, up;
            row_bounds(L, in, r, lo, up);
            const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
            // bound multipliers follow the ORIGINAL direction (step a_z); primal variables and y the accepted one
            const double dy = S.dy[r], dyz = use_soc ? S.dyo[r] : dy;
            const double s_old = S.s[r], p_old = S.p[r], n_old = S.n[r], y = S.y[r];
            const double zL0 = S.zL[r], zU0 = S.zU[r], zp0 = S.zp[r], zn0 = S.zn[r];
            const Lin q = row_lin(lo, up, eq, s_old, p_old, n_old, y, zL0, zU0, zp0, zn0, mu, rho, delta_w);
            const double ds = (dy - q.rs) * q.iDs, dp = (dy - q.rp) * q.iDp, dn = (-dy - q.rn) * q.iDn;
            const double dsz = (dyz - q.rs) * q.iDs, dpz = (dyz - q.rp) * q.iDp, dnz = (-dyz - q.rn) * q.iDn;
            const double s = eq ? 0.0 : s_old + a_try * ds, p = p_old + a_try * dp, n = n_old + a_try * dn;
            const double ks = OBCA_KAPPA_SIGMA;
            if (hasL) {
                const double zL = zL0 + a_z * ((mu - zL0 * dsz) / (s_old - lo) - zL0), sl = s - lo;
                S.zL[r] = fmax(fmin(zL, ks * mu / sl), mu / (ks * sl));
            }
            if (hasU) {
                const double zU = zU0 + a_z * ((mu + zU0 * dsz) / (up - s_old) - zU0), su = up - s;
                S.zU[r] = fmax(fmin(zU, ks * mu / su), mu / (ks * su));
            }
            const double zp = zp0 + a_z * ((mu - zp0 * dpz) / p_old - zp0);
            const double zn = zn0 + a_z * ((mu - zn0 * dnz) / n_old - zn0);
            S.zp[r] = fmax(fmin(zp, ks * mu / p), mu / (ks * p));
            S.zn[r] = fmax(fmin(zn, ks * mu / n), mu / (ks * n));
            S.s[r] = s; S.p[r] = p; S.n[r] = n;
            S.y[r] = y + a_try * dy;
        }
        for (int t = 0; t < 2 * L.npair; ++t) S.nu[t] += a_try * S.dnu[t];
        for (int t = 0; t < L.n; ++t) S.x[t] = S.xt[t];
        fobj_prev = fobj;
        have_prev = true;
        eval_geom(L, S, S.x, S.ct, S.st, S.cc, 0);
        f = eval_objective<true>(L, S, in, S.x, sf, 0);
        for (int r = 0; r < L.R; ++r) S.g[r] = row_value(L, S, in, S.x, S.ct, S.st, S.cc, r);
        for (int pr = 0; pr < L.npair; ++pr) {
            double e1, e2;
            rot_value(L, S.x, S.ct, S.st, S.cc, pr, e1, e2);
            S.crot[2 * pr] = e1; S.crot[2 * pr + 1] = e2;
        }
    }
    if ((o.status == OBCA_STATUS_OK || o.status == OBCA_STATUS_ACCEPTABLE) && o.elastic > O.feas_tol)
        o.status = OBCA_STATUS_INFEASIBLE;
    o.iters = it;
    o.f = f / sf;
    o.sf = sf;
    o.ts_opt = L.free_T ? S.x[L.iT()] * in.Ts : in.Ts;
    return o;
}

// bind the carve-up to a workspace and fill the layout / per-instance constants
LPI_FN void make_layout(Lay& L, int N, int nO, int M, int variant) {
    L.N = N; L.nO = nO; L.M = M; L.variant = variant;
    L.free_T = (variant == 4) ? 1 : 0;
    L.NS = 5 + M + 4 * nO;
    L.n = (N + 1) * (3 + M + 4 * nO) + 2 * N + L.free_T;
    L.npair = (N + 1) * nO;
    L.r_init = 0; L.r_dyn = 3; L.r_term = 3 + 3 * N;
    L.r_xb = L.r_term + (variant == 4 ? 3 : 0);
    L.r_ub = L.r_xb + 2 * (N + 1);
    L.r_acc = L.r_ub + 2 * N;
    L.r_T = L.r_acc + 2 * N;
    L.r_tx = L.r_T + (L.free_T ? 2 : 0);
    L.r_norm = L.r_tx + (variant == 6 ? 2 : 0);
    L.r_dist = L.r_norm + L.npair;
    L.r_lam = L.r_dist + L.npair;
    L.r_mu = L.r_lam + (N + 1) * M;
    L.R = L.r_mu + (N + 1) * 4 * nO;
}

LPI_FN void bind(Sh& S, const Carve& c, double* ws, size_t stride, size_t inst, const int* offm) {
#define B(name) S.name = SP{ws + (size_t)c.name * stride + inst, stride};
    B(x) B(xt) B(dx) B(gf) B(bx) B(y) B(Einv) B(yhat) B(gh) B(dy) B(s) B(p) B(n) B(zL) B(zU) B(zp) B(zn) B(g)
    B(ct) B(st) B(cc) B(ctt) B(stt) B(cct) B(nu) B(dnu) B(crot) B(Aobs) B(bobs) B(xref) B(Lall) B(lall) B(Y)
    B(Pk) B(qk) B(Kk) B(kapk) B(Mik) B(filt) B(gsoc) B(rest) B(dyo) B(dxo) B(dnuo) B(crot_t)
#undef B
    S.offm = offm;
}

// whole per-instance job: load inputs (instance-major C-ABI layout), solve, store outputs
// ws_col: which workspace column the instance uses (the device gives every instance its own: ws_col = inst; a host
// thread pool can reuse one contiguous column per thread with stride 1)
LPI_FN void run_instance(const ObcaLaunch& A, double* ws, size_t stride, size_t inst, const int* offm, size_t ws_col) {
    if (A.variant[inst] == 0) { A.status[inst] = OBCA_STATUS_SKIPPED; A.iters[inst] = 0; return; }
    {
        const int v = A.variant[inst];
        if ((v != 4 && v != 6 && v != 8) || (v == 6 && A.term == nullptr)) { A.status[inst] = OBCA_STATUS_BAD_VARIANT; A.iters[inst] = 0; return; }
    }
    Lay L;
    make_layout(L, A.N, A.nO, A.M, A.variant[inst]);
    L.R_cap = A.R_max;
    const Carve c = carve(A.N, A.nO, A.M, A.n_max, A.R_max);
    Sh S;
    bind(S, c, ws, stride, ws_col, offm);
    Inst in;
    const bool fr = L.free_T != 0;
    for (int j = 0; j < 9; ++j) { in.Q[j] = fr ? A.prm.free_time.Q[j] : A.prm.fixed_time.Q[j]; in.P[j] = fr ? A.prm.free_time.P[j] : A.prm.fixed_time.P[j]; }
    for (int j = 0; j < 4; ++j) { in.R1[j] = fr ? A.prm.free_time.R1[j] : A.prm.fixed_time.R1[j]; in.R2[j] = fr ? A.prm.free_time.R2[j] : A.prm.fixed_time.R2[j]; in.gego[j] = A.prm.gego[j]; }
    for (int j = 0; j < 2; ++j) { in.xL[j] = A.prm.xL[j]; in.xU[j] = A.prm.xU[j]; in.uL[j] = A.prm.uL[j]; in.uU[j] = A.prm.uU[j]; }
    in.off = A.prm.off; in.dmin = A.prm.dmin;
    for (int j = 0; j < 3; ++j) in.x0[j] = A.x0[inst * 3 + j];
    for (int j = 0; j < 2; ++j) in.u0[j] = A.u0[inst * 2 + j];
    in.Ts = A.Ts[inst];
    for (int j = 0; j < 3; ++j) in.term[j] = (L.variant == 6) ? A.term[inst * 3 + j] : 0.0;
    const int N1 = L.N + 1;
    const double* xr = A.xref + inst * 3 * N1;
    for (int t = 0; t < 3 * N1; ++t) S.xref[t] = xr[t];
    const double* Ag = A.A + inst * (size_t)N1 * L.M * 2;
    const double* bg = A.b + inst * (size_t)N1 * L.M;
    for (int k = 0; k < N1; ++k) {
        const int ks = (L.variant == 4) ? 0 : k;                  // q5: mpc4 reads step 0 only
        for (int q = 0; q < 2 * L.M; ++q) S.Aobs[k * 2 * L.M + q] = Ag[(size_t)ks * L.M * 2 + q];
        for (int q = 0; q < L.M; ++q) S.bobs[k * L.M + q] = bg[(size_t)ks * L.M + q];
    }
    const double dis = (S.xref[0 * N1 + L.N] - in.x0[0]) + (S.xref[1 * N1 + L.N] - in.x0[1]);
    in.Tmax = dis / (L.N * in.uU[0] * in.Ts) + 1.0;
    const bool warm = A.warm_z != nullptr && (A.warm_use == nullptr || A.warm_use[inst] != 0);
    // obca_mpc6 whose terminal set no trajectory can reach (csrc/obca_device.h: obca_terminal_shortfall) is not run
    double shortfall = -INFINITY;
    if (L.variant == 6) {
        const double sh = obca_terminal_shortfall(L.N, in.Ts, in.x0[0], cos(in.x0[2]), in.u0[0], in.uL[0], in.uU[0], in.xU[0], in.term[0], A.prm.opt.feas_tol);
        shortfall = sh;
        if (sh > 0.0 && A.prm.opt.screen) {
            double* xo = A.xopt + inst * 3 * N1;
            double* uo = A.uopt + inst * 2 * L.N;
            for (int j = 0; j < 3; ++j) for (int k = 0; k < N1; ++k) xo[j * N1 + k] = in.x0[j];
            for (int t = 0; t < 2 * L.N; ++t) uo[t] = 0.0;
            A.ts_opt[inst] = in.Ts; A.status[inst] = OBCA_STATUS_INFEASIBLE; A.iters[inst] = 0;
            if (A.info) { double* io = A.info + inst * 4; io[0] = 0.0; io[1] = sh; io[2] = 0.0; io[3] = 0.0; }
            if (A.cert_z != nullptr) {
                double* zc = A.cert_z + inst * (size_t)A.n_max;
                for (int t = 0; t < L.n; ++t) zc[t] = 0.0;
                for (int k = 0; k < N1; ++k) for (int j = 0; j < 3; ++j) zc[L.ip(k) + j] = in.x0[j];
            }
            if (A.cert_y != nullptr) {
                double* yc = A.cert_y + inst * (size_t)(A.R_max + 2 * L.npair);
                for (int q = 0; q < L.R + 2 * L.npair; ++q) yc[q] = 0.0;
            }
            return;
        }
    }
    // The start ladder (oracle/ipm_dense.py:solve; csrc/obca_kernel.hip runs the same passes): the starts of the order one after
    // the other until one ends at a feasible point; a free-time solve that converged with elastic variables left is first
    // repeated from the same start with rho x 100, then rho x 1000 (OBCA_RHO_ESCALATION; the next start begins at the base penalty again).  The caller's
    // optional warm start takes the place of the first cold start of the order.
    const ObcaOptsDev& O0 = A.prm.opt;
    const int max_iter_v = L.free_T ? O0.max_iter_free : O0.max_iter_fixed;
    Out o;
    o.status = OBCA_STATUS_MAXITER;
    int iters = 0, nfact = 0;
    // what a pass leaves behind goes to the caller's buffers
    auto store = [&](const Out& r) {
        if (A.warm_z != nullptr && (r.status == OBCA_STATUS_OK || r.status == OBCA_STATUS_ACCEPTABLE)) {
            double* zp = A.warm_z + inst * (size_t)A.n_max;
            for (int t = 0; t < L.n; ++t) zp[t] = S.x[t];
        }
        if (A.cert_z != nullptr) {
            double* zc = A.cert_z + inst * (size_t)A.n_max;
            for (int t = 0; t < L.n; ++t) zc[t] = S.x[t];
        }
        if (A.cert_y != nullptr) {
            double* yc = A.cert_y + inst * (size_t)(A.R_max + 2 * L.npair);
            const double isf = 1.0 / r.sf;
            for (int q = 0; q < L.R; ++q) yc[q] = S.y[q] * isf;
            for (int t = 0; t < 2 * L.npair; ++t) yc[L.R + t] = S.nu[t] * isf;
        }
        double* xo = A.xopt + inst * 3 * N1;
        double* uo = A.uopt + inst * 2 * L.N;
        for (int j = 0; j < 3; ++j) for (int k = 0; k < N1; ++k) xo[j * N1 + k] = S.x[L.ip(k) + j];
        for (int j = 0; j < 2; ++j) for (int k = 0; k < L.N; ++k) uo[j * L.N + k] = S.x[L.iu(k) + j];
        A.ts_opt[inst] = r.ts_opt;
        A.status[inst] = r.status;
        if (A.info) {
            double* io = A.info + inst * 4;
            io[0] = r.f; io[1] = r.elastic; io[2] = r.E0;
        }
    };
    // (which pass's answer stays there when no start ends at a feasible point: csrc/obca_device.h: OBCA_LADDER_REPLACES)
    bool have = false;
    int held_st = OBCA_STATUS_MAXITER, held_start = -1;
    for (int s = 0; s < O0.nstarts; ++s) {
        if (s > 0 && (o.status == OBCA_STATUS_OK || o.status == OBCA_STATUS_ACCEPTABLE || o.status == OBCA_STATUS_BAD_BOUNDS)) break;
        const int order = OBCA_EFFECTIVE_ORDER(O0.order, L.variant, warm, O0.nstarts == 1);
        const int kind = OBCA_START_KIND(order, s);
        const int cap = O0.nstarts == 1 ? max_iter_v : (s == 0 ? O0.patience : O0.retry_iter);
        const bool use_warm = warm && kind == OBCA_WARM_KIND(order);
        const double* z = use_warm ? A.warm_z + inst * (size_t)A.n_max : nullptr;
        const double mu0 = use_warm ? A.warm_mu : (kind == OBCA_KIND_WINDOW ? OBCA_RESTART_MU : OBCA_MU_INIT);
        for (int level = 0; level <= OBCA_N_ESCALATIONS; ++level) {
            // the l1 penalty is exact only while rho exceeds the multipliers: "rho too small" looks like "infeasible"
            if (level > 0 && !(o.status == OBCA_STATUS_INFEASIBLE && L.free_T)) break;
            ObcaOptsDev O = O0;
            if (level) O.rho *= OBCA_RHO_ESCALATION(level);
            o = solve_instance(L, S, in, O, z, mu0, kind, cap);
            iters += o.iters; nfact += o.nfact;
            if (OBCA_LADDER_REPLACES(o.status, s, have, held_st, held_start)) { store(o); have = true; held_st = o.status; held_start = s; }
        }
    }
    // The dodge rung (csrc/obca_device.h: OBCA_KIND_DODGE_*): fixed-time problems on which every start of the order ended
    // without a feasible point -- the window moved to the right, then to the left; both run, the feasible answer with the lower
    // objective is the one that stays in the caller's buffers (the first pass's outputs are overwritten only by a better one)
    if (O0.dodge && !L.free_T && shortfall < -OBCA_DODGE_MIN_SPARE && !(o.status == OBCA_STATUS_OK || o.status == OBCA_STATUS_ACCEPTABLE || o.status == OBCA_STATUS_BAD_BOUNDS)) {
        bool dodged = false;
        double fbest = 0.0;
    #ifndef OBCA_DODGE_LEVEL1_MU          /* study knob of the HOST build only (tools/dodge_mu_study.py): the first level's barrier parameter; the kernels use OBCA_RESTART_MU */
#define OBCA_DODGE_LEVEL1_MU OBCA_RESTART_MU
#endif
    // (second level, obca_mpc8 only -- csrc/obca_device.h: OBCA_DODGE_PASSES: the same two starts at OBCA_DODGE_LEVEL2_MU, if the first found nothing)
        for (int pass = 0; pass < OBCA_DODGE_PASSES(L.variant); ++pass) {
            if (pass == 2 && dodged) break;
            const int side = pass & 1;
            const Out r = solve_instance(L, S, in, O0, nullptr, pass < 2 ? OBCA_DODGE_LEVEL1_MU : OBCA_DODGE_LEVEL2_MU, side == 0 ? OBCA_KIND_DODGE_R : OBCA_KIND_DODGE_L, O0.retry_iter);
            iters += r.iters; nfact += r.nfact;
            const bool ok = r.status == OBCA_STATUS_OK || r.status == OBCA_STATUS_ACCEPTABLE;
            if (ok && (!dodged || r.f < fbest)) { store(r); dodged = true; fbest = r.f; }
        }
    }
    A.iters[inst] = iters;
    if (A.info) A.info[inst * 4 + 3] = (double)nfact;
}

}  // namespace lpi
#endif
