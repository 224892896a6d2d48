/*
 * ORACLE (test infrastructure, never on the product path): plain-C restatement of the OBCA NLP and of the
 * elastic interior-point method specified in oracle/ipm_dense.py, with a DENSE Bunch-Kaufman LDL^T of the
 * augmented KKT system (no structure exploited -- deliberately independent of the HIP kernel's two-level
 * solve).  Used by tests/ as the checker at batch sizes numpy is too slow for and by bench.py as the
 * `cpu_baseline` ("port").
 *
 * NLP: reference src/obca.py:828-1071 (obca_mpc4), :1361-1562 (obca_mpc6), :1564-1758 (obca_mpc8); the
 * function-by-function citations are in oracle/obca_nlp.py, which this file follows line by line and
 * against which it is tested (tests/test_c_oracle.py).  PARITY at the solver boundary: no IPOPT here; pinned on the ONE
 * IPOPT output the reference repository holds (its demo9 GIF: 69 consecutive closed-loop steps show the reference's digits with
 * the default start ladder, 42 with the reference's literal zero start first -- tests/test_reference_gif.py), UNPINNED beyond
 * that run; the NLP functions
 * are pinned through tests/golden/nlp_eval.json.
 *
 * Build: make -C oracle   ->  oracle/_build/libobca_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXO 8
#define ACC0 0.6
#define ACC1 0.5235987755982988
#define TMIN 1e-4
#define INF (1.0 / 0.0)

enum { ST_OK = 0, ST_ACCEPTABLE = 1, ST_INFEASIBLE = 2, ST_MAXITER = -1, ST_LINESEARCH = -2, ST_NUMERIC = -3, ST_BAD_BOUNDS = -4 };

typedef struct {
    int variant, N, nO, M, m[MAXO], offm[MAXO + 1];
    int NS, n, freeT;
    double x0[3], u0[2], Ts, Tmax, term[3];
    const double *xref, *A, *b;            /* xref[3][N+1], A[N+1][M][2], b[N+1][M] (per-step rows) */
    double Q[9], P[9], R1[4], R2[4], xL[2], xU[2], uL[2], uU[2], g[4], off, dmin;
    /* rows */
    int mh;                                 /* hard rows: 2 per (k,i) */
    int naug;                               /* elastic equality rows kept bordered: init, dyn, term */
    int nineq, me;                          /* me = naug + nineq elastic rows */
} Prob;

static int ip(const Prob* p, int k) { return k * p->NS; }
static int iu(const Prob* p, int k) { return k * p->NS + 3; }
static int il(const Prob* p, int k) { return k * p->NS + (k < p->N ? 5 : 3); }
static int imu(const Prob* p, int k) { return il(p, k) + p->M; }
static int iT(const Prob* p) { return p->n - 1; }
static double Tof(const Prob* p, const double* x) { return p->freeT ? x[iT(p)] : 1.0; }

static double objective(const Prob* p, const double* x, double* g, double* H) {
    const int N = p->N, n = p->n;
    const double T = Tof(p, x), h = T * p->Ts;
    double f = 0.0;
    if (g) memset(g, 0, sizeof(double) * n);
    for (int t = 0; t <= N; ++t) {
        const double* W = t < N ? p->Q : p->P;
        const int i = ip(p, t);
        double e[3], We[3];
        for (int j = 0; j < 3; ++j) e[j] = x[i + j] - p->xref[j * (N + 1) + t];
        for (int a = 0; a < 3; ++a) We[a] = W[3 * a] * e[0] + W[3 * a + 1] * e[1] + W[3 * a + 2] * e[2];
        f += e[0] * We[0] + e[1] * We[1] + e[2] * We[2];
        if (g) for (int a = 0; a < 3; ++a) g[i + a] += 2 * We[a];
        if (H) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) H[(i + a) * n + i + b] += 2 * W[3 * a + b];
        if (t == N) break;
        const int j = iu(p, t);
        const double u0 = x[j], u1 = x[j + 1];
        const double r0 = p->R1[0] * u0 + p->R1[1] * u1, r1 = p->R1[2] * u0 + p->R1[3] * u1;
        f += u0 * r0 + u1 * r1;
        if (g) { g[j] += 2 * r0; g[j + 1] += 2 * r1; }
        if (H) for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) H[(j + a) * n + j + b] += 2 * p->R1[2 * a + b];
        if (t < N - 1) {
            const int j2 = iu(p, t + 1);
            const double q0 = x[j2] - u0, q1 = x[j2 + 1] - u1;
            const double s0 = p->R2[0] * q0 + p->R2[1] * q1, s1 = p->R2[2] * q0 + p->R2[3] * q1;
            const double qq = q0 * s0 + q1 * s1, ih2 = 1.0 / (h * h);
            f += qq * ih2;
            if (g) {
                g[j2] += 2 * s0 * ih2; g[j2 + 1] += 2 * s1 * ih2; g[j] -= 2 * s0 * ih2; g[j + 1] -= 2 * s1 * ih2;
                if (p->freeT) g[iT(p)] += -2 * qq * ih2 / T;
            }
            if (H) {
                const double sv[2] = {s0, s1};
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
                    const double B = 2 * p->R2[2 * a + b] * ih2;
                    H[(j2 + a) * n + j2 + b] += B; H[(j + a) * n + j + b] += B;
                    H[(j2 + a) * n + j + b] -= B; H[(j + a) * n + j2 + b] -= B;
                }
                if (p->freeT) {
                    const int it = iT(p);
                    for (int a = 0; a < 2; ++a) {
                        const double c = 4 * sv[a] * ih2 / T;
                        H[(j2 + a) * n + it] -= c; H[it * n + j2 + a] -= c;
                        H[(j + a) * n + it] += c; H[it * n + j + a] += c;
                    }
                    H[it * n + it] += 6 * qq * ih2 / (T * T);
                }
            }
        }
    }
    if (p->freeT) {
        f += (N + 1) * (10 * T + T * T);
        if (g) g[iT(p)] += (N + 1) * (10 + 2 * T);
        if (H) H[iT(p) * n + iT(p)] += 2 * (N + 1);
    }
    return f;
}

/* c0,c1 of obstacle i at stage k */
static void cvec(const Prob* p, const double* x, int k, int i, double* c0, double* c1) {
    const double* lam = x + il(p, k);
    const double* A = p->A + (size_t)k * p->M * 2;
    double a = 0, b = 0;
    for (int j = p->offm[i]; j < p->offm[i + 1]; ++j) { a += A[2 * j] * lam[j]; b += A[2 * j + 1] * lam[j]; }
    *c0 = a; *c1 = b;
}

/* hard rows (rotation equalities): values ch[mh], optional Jacobian Jh[mh][n], optional Hessian += sum y*Hess */
static void hard_rows(const Prob* p, const double* x, double* ch, double* Jh, const double* y, double* H) {
    const int n = p->n;
    int r = 0;
    for (int k = 0; k <= p->N; ++k) {
        const int ipk = ip(p, k);
        const double ct = cos(x[ipk + 2]), st = sin(x[ipk + 2]);
        for (int i = 0; i < p->nO; ++i, r += 2) {
            double c0, c1;
            cvec(p, x, k, i, &c0, &c1);
            const double* mu = x + imu(p, k) + 4 * i;
            const double* A = p->A + (size_t)k * p->M * 2;
            if (ch) { ch[r] = mu[0] - mu[2] + ct * c0 + st * c1; ch[r + 1] = mu[1] - mu[3] - st * c0 + ct * c1; }
            for (int j = p->offm[i]; j < p->offm[i + 1]; ++j) {
                const double a0 = A[2 * j], a1 = A[2 * j + 1];
                const int col = il(p, k) + j;
                if (Jh) { Jh[r * n + col] = ct * a0 + st * a1; Jh[(r + 1) * n + col] = -st * a0 + ct * a1; }
                if (H) {
                    const double hl = y[r] * (-st * a0 + ct * a1) + y[r + 1] * (-ct * a0 - st * a1);
                    H[(ipk + 2) * n + col] += hl; H[col * n + ipk + 2] += hl;
                }
            }
            if (Jh) {
                const int im = imu(p, k) + 4 * i;
                Jh[r * n + im] = 1; Jh[r * n + im + 2] = -1; Jh[(r + 1) * n + im + 1] = 1; Jh[(r + 1) * n + im + 3] = -1;
                Jh[r * n + ipk + 2] = -st * c0 + ct * c1;
                Jh[(r + 1) * n + ipk + 2] = -ct * c0 - st * c1;
            }
            if (H) H[(ipk + 2) * n + ipk + 2] += y[r] * (-ct * c0 - st * c1) + y[r + 1] * (st * c0 - ct * c1);
        }
    }
}

/* elastic rows: [init(3) dyn(3N) term(3 if v4)] then inequalities in the layout of obca_nlp.ineq_layout().
 * ge values, lb/ub (lb==ub==0 for the equality rows), optional dense Jacobian Je[me][n], optional Hessian. */
static void elastic_rows(const Prob* p, const double* x, double* ge, double* lb, double* ub, double* Je,
                         const double* y, double* H) {
    const int N = p->N, n = p->n, fr = p->freeT;
    const double T = Tof(p, x), h = T * p->Ts;
    int r = 0;
#define SETB(lo, up) do { if (lb) { lb[r] = (lo); ub[r] = (up); } } while (0)
    for (int j = 0; j < 3; ++j, ++r) { if (ge) ge[r] = x[j] - p->x0[j]; SETB(0, 0); if (Je) Je[r * n + j] = 1; }
    for (int k = 0; k < N; ++k) {
        const int i = ip(p, k), j = iu(p, k), i2 = ip(p, k + 1);
        const double th = x[i + 2], v = x[j], w = x[j + 1], ct = cos(th), st = sin(th);
        if (ge) { ge[r] = x[i2] - x[i] - h * v * ct; ge[r + 1] = x[i2 + 1] - x[i + 1] - h * v * st; ge[r + 2] = x[i2 + 2] - x[i + 2] - h * w; }
        if (lb) for (int q = 0; q < 3; ++q) { lb[r + q] = 0; ub[r + q] = 0; }
        if (Je) {
            for (int q = 0; q < 3; ++q) { Je[(r + q) * n + i2 + q] += 1; Je[(r + q) * n + i + q] -= 1; }
            Je[r * n + i + 2] += h * v * st; Je[(r + 1) * n + i + 2] -= h * v * ct;
            Je[r * n + j] -= h * ct; Je[(r + 1) * n + j] -= h * st; Je[(r + 2) * n + j + 1] -= h;
            if (fr) { Je[r * n + iT(p)] -= p->Ts * v * ct; Je[(r + 1) * n + iT(p)] -= p->Ts * v * st; Je[(r + 2) * n + iT(p)] -= p->Ts * w; }
        }
        if (H) {
            const double px = y[r], py = y[r + 1], pt = y[r + 2];
            H[(i + 2) * n + i + 2] += h * v * (px * ct + py * st);
            const double bq = h * (px * st - py * ct);
            H[(i + 2) * n + j] += bq; H[j * n + i + 2] += bq;
            if (fr) {
                const int it = iT(p);
                const double d = -p->Ts * v * (-px * st + py * ct), e = -p->Ts * (px * ct + py * st);
                H[(i + 2) * n + it] += d; H[it * n + i + 2] += d;
                H[j * n + it] += e; H[it * n + j] += e;
                H[(j + 1) * n + it] += -p->Ts * pt; H[it * n + j + 1] += -p->Ts * pt;
            }
        }
        r += 3;
    }
    if (p->variant == 4)
        for (int j = 0; j < 3; ++j, ++r) {
            if (ge) ge[r] = x[ip(p, N) + j] - p->xref[j * (N + 1) + N];
            SETB(0, 0);
            if (Je) Je[r * n + ip(p, N) + j] = 1;
        }
    for (int k = 0; k <= N; ++k) for (int j = 0; j < 2; ++j, ++r) { if (ge) ge[r] = x[ip(p, k) + j]; SETB(p->xL[j], p->xU[j]); if (Je) Je[r * n + ip(p, k) + j] = 1; }
    for (int k = 0; k < N; ++k) for (int j = 0; j < 2; ++j, ++r) { if (ge) ge[r] = x[iu(p, k) + j]; SETB(p->uL[j], p->uU[j]); if (Je) Je[r * n + iu(p, k) + j] = 1; }
    for (int k = 0; k < N; ++k) for (int c = 0; c < 2; ++c, ++r) {
        const int cur = iu(p, k) + c;
        const double prev = k == 0 ? p->u0[c] : x[iu(p, k - 1) + c], q = prev - x[cur];
        if (ge) ge[r] = q / h;
        SETB(c ? -ACC1 : -ACC0, c ? ACC1 : ACC0);
        if (Je) { Je[r * n + cur] = -1 / h; if (k > 0) Je[r * n + iu(p, k - 1) + c] = 1 / h; if (fr) Je[r * n + iT(p)] = -q / (T * h); }
        if (H && fr) {
            const int it = iT(p);
            const double yy = y[r];
            H[cur * n + it] += yy / (T * h); H[it * n + cur] += yy / (T * h);
            if (k > 0) { const int pi = iu(p, k - 1) + c; H[pi * n + it] -= yy / (T * h); H[it * n + pi] -= yy / (T * h); }
            H[it * n + it] += yy * 2 * q / (T * T * h);
        }
    }
    if (p->variant == 4)
        for (int k = 0; k <= N; ++k) {
            if (ge) ge[r] = T; SETB(0, INF); if (Je) Je[r * n + iT(p)] = 1; ++r;
            if (ge) ge[r] = T; SETB(TMIN, p->Tmax); if (Je) Je[r * n + iT(p)] = 1; ++r;
        }
    if (p->variant == 6) {
        if (ge) ge[r] = x[ip(p, N)]; SETB(p->term[0], INF); if (Je) Je[r * n + ip(p, N)] = 1; ++r;
        if (ge) ge[r] = x[ip(p, N) + 1]; SETB(p->term[1], p->term[2]); if (Je) Je[r * n + ip(p, N) + 1] = 1; ++r;
    }
    for (int k = 0; k <= N; ++k) {
        const int ipk = ip(p, k);
        const double ct = cos(x[ipk + 2]), st = sin(x[ipk + 2]);
        for (int i = 0; i < p->nO; ++i) {
            double c0, c1;
            cvec(p, x, k, i, &c0, &c1);
            const double* A = p->A + (size_t)k * p->M * 2;
            const double* bb = p->b + (size_t)k * p->M;
            const int o0 = p->offm[i], o1 = p->offm[i + 1], ill = il(p, k), im = imu(p, k) + 4 * i;
            /* norm */
            if (ge) ge[r] = c0 * c0 + c1 * c1;
            SETB(-INF, 1.0);
            for (int j = o0; j < o1; ++j) {
                if (Je) Je[r * n + ill + j] = 2 * (A[2 * j] * c0 + A[2 * j + 1] * c1);
                if (H) for (int l = o0; l < o1; ++l) H[(ill + j) * n + ill + l] += y[r] * 2 * (A[2 * j] * A[2 * l] + A[2 * j + 1] * A[2 * l + 1]);
            }
            ++r;
            /* dist */
            const double tx = x[ipk] + ct * p->off, ty = x[ipk + 1] + st * p->off;
            if (ge) {
                double v = tx * c0 + ty * c1;
                for (int q = 0; q < 4; ++q) v -= p->g[q] * x[im + q];
                for (int j = o0; j < o1; ++j) v -= bb[j] * x[ill + j];
                ge[r] = v;
            }
            SETB(p->dmin, INF);
            if (Je) {
                for (int q = 0; q < 4; ++q) Je[r * n + im + q] = -p->g[q];
                for (int j = o0; j < o1; ++j) Je[r * n + ill + j] = tx * A[2 * j] + ty * A[2 * j + 1] - bb[j];
                Je[r * n + ipk] = c0; Je[r * n + ipk + 1] = c1; Je[r * n + ipk + 2] = p->off * (-st * c0 + ct * c1);
            }
            if (H) {
                const double yy = y[r];
                for (int j = o0; j < o1; ++j) {
                    const double a0 = A[2 * j], a1 = A[2 * j + 1], hl = yy * p->off * (-st * a0 + ct * a1);
                    H[ipk * n + ill + j] += yy * a0; H[(ill + j) * n + ipk] += yy * a0;
                    H[(ipk + 1) * n + ill + j] += yy * a1; H[(ill + j) * n + ipk + 1] += yy * a1;
                    H[(ipk + 2) * n + ill + j] += hl; H[(ill + j) * n + ipk + 2] += hl;
                }
                H[(ipk + 2) * n + ipk + 2] += yy * p->off * (-ct * c0 - st * c1);
            }
            ++r;
        }
    }
    for (int k = 0; k <= N; ++k) {
        for (int j = 0; j < p->M; ++j, ++r) { if (ge) ge[r] = x[il(p, k) + j]; SETB(0, INF); if (Je) Je[r * n + il(p, k) + j] = 1; }
        for (int j = 0; j < 4 * p->nO; ++j, ++r) { if (ge) ge[r] = x[imu(p, k) + j]; SETB(0, INF); if (Je) Je[r * n + imu(p, k) + j] = 1; }
    }
#undef SETB
}

/* ---- dense symmetric indefinite solve: Bunch-Kaufman LDL^T (lower), returns #negative eigenvalues ---- */
static int bk_factor(double* A, int n, int* piv) {
    const double alpha = (1.0 + sqrt(17.0)) / 8.0;
    int k = 0, neg = 0;
    while (k < n) {
        int kstep = 1, kp = k, imax = k;
        double absakk = fabs(A[k * n + k]), colmax = 0.0;
        for (int i = k + 1; i < n; ++i) if (fabs(A[i * n + k]) > colmax) { colmax = fabs(A[i * n + k]); imax = i; }
        if (fmax(absakk, colmax) == 0.0) { piv[k] = k; ++k; continue; }
        if (absakk < alpha * colmax) {
            double rowmax = 0.0;
            for (int j = k; j < imax; ++j) rowmax = fmax(rowmax, fabs(A[imax * n + j]));
            for (int i = imax + 1; i < n; ++i) rowmax = fmax(rowmax, fabs(A[i * n + imax]));
            if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
            else if (fabs(A[imax * n + imax]) >= alpha * rowmax) kp = imax;
            else { kp = imax; kstep = 2; }
        }
        const int kk = k + kstep - 1;
        if (kp != kk) {           /* symmetric interchange of rows/cols kk and kp (lower storage) */
            for (int i = kp + 1; i < n; ++i) { double t = A[i * n + kk]; A[i * n + kk] = A[i * n + kp]; A[i * n + kp] = t; }
            for (int j = kk + 1; j < kp; ++j) { double t = A[j * n + kk]; A[j * n + kk] = A[kp * n + j]; A[kp * n + j] = t; }
            { double t = A[kk * n + kk]; A[kk * n + kk] = A[kp * n + kp]; A[kp * n + kp] = t; }
            if (kstep == 2) { double t = A[(k + 1) * n + k]; A[(k + 1) * n + k] = A[kp * n + k]; A[kp * n + k] = t; }
        }
        if (kstep == 1) {
            const double d = A[k * n + k];
            if (d < 0) ++neg;
            const double r1 = 1.0 / d;
            for (int j = k + 1; j < n; ++j) {
                const double t = A[j * n + k] * r1;
                for (int i = j; i < n; ++i) A[i * n + j] -= A[i * n + k] * t;
            }
            for (int i = k + 1; i < n; ++i) A[i * n + k] *= r1;
            piv[k] = kp;
        } else {
            const double d21 = A[(k + 1) * n + k], d11 = A[(k + 1) * n + k + 1] / d21, d22 = A[k * n + k] / d21;
            const double t = 1.0 / (d11 * d22 - 1.0), dd = t / d21;
            const double det = A[k * n + k] * A[(k + 1) * n + k + 1] - d21 * d21;
            if (det < 0) ++neg; else if (A[k * n + k] + A[(k + 1) * n + k + 1] < 0) neg += 2;
            for (int j = k + 2; j < n; ++j) {
                const double wk = dd * (d11 * A[j * n + k] - A[j * n + k + 1]);
                const double wk1 = dd * (d22 * A[j * n + k + 1] - A[j * n + k]);
                for (int i = j; i < n; ++i) A[i * n + j] -= A[i * n + k] * wk + A[i * n + k + 1] * wk1;
                A[j * n + k] = wk; A[j * n + k + 1] = wk1;
            }
            piv[k] = -(kp + 1); piv[k + 1] = -(kp + 1);
        }
        k += kstep;
    }
    return neg;
}

static void bk_solve(const double* A, int n, const int* piv, double* b) {
    int k = 0;
    while (k < n) {
        if (piv[k] >= 0) {
            const int kp = piv[k];
            if (kp != k) { double t = b[k]; b[k] = b[kp]; b[kp] = t; }
            for (int i = k + 1; i < n; ++i) b[i] -= A[i * n + k] * b[k];
            b[k] /= A[k * n + k];
            ++k;
        } else {
            const int kp = -piv[k] - 1;
            if (kp != k + 1) { double t = b[k + 1]; b[k + 1] = b[kp]; b[kp] = t; }
            for (int i = k + 2; i < n; ++i) b[i] -= A[i * n + k] * b[k] + A[i * n + k + 1] * b[k + 1];
            const double akm1k = A[(k + 1) * n + k], akm1 = A[k * n + k] / akm1k, ak = A[(k + 1) * n + k + 1] / akm1k;
            const double denom = akm1 * ak - 1.0, bkm1 = b[k] / akm1k, bk = b[k + 1] / akm1k;
            b[k] = (ak * bkm1 - bk) / denom; b[k + 1] = (akm1 * bk - bkm1) / denom;
            k += 2;
        }
    }
    k = n - 1;
    while (k >= 0) {
        if (piv[k] >= 0) {
            for (int i = k + 1; i < n; ++i) b[k] -= A[i * n + k] * b[i];
            const int kp = piv[k];
            if (kp != k) { double t = b[k]; b[k] = b[kp]; b[kp] = t; }
            --k;
        } else {
            for (int i = k + 1; i < n; ++i) { b[k] -= A[i * n + k] * b[i]; b[k - 1] -= A[i * n + k - 1] * b[i]; }
            const int kp = -piv[k] - 1;
            if (kp != k) { double t = b[k]; b[k] = b[kp]; b[kp] = t; }
            k -= 2;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------ */
typedef struct {
    double tol, rho, feas_tol;
    int max_iter_free, max_iter_fixed, max_soc;
} Opts;

#define MU_INIT 0.1
#define RESTART_MU 1.0            /* csrc/obca_device.h: OBCA_RESTART_MU */
#define WINDOW_SPEED_FRAC 0.9
#define RETRY_ITER(N) (300 + 10 * (N)) /* csrc/obca_device.h: OBCA_RETRY_ITER */
#define PATIENCE(N) (500 + 10 * (N)) /* csrc/obca_device.h: OBCA_PATIENCE */
/* the three starts of the ladder and the three orders (include/obca_mpc.h: start_order; csrc/obca_device.h: OBCA_START_KIND) */
enum { KIND_X0 = 0, KIND_WINDOW = 1, KIND_ZEROS = 2, KIND_DODGE_R = 3, KIND_DODGE_L = 4 };   /* the last two: oracle/ipm_dense.py:dodge_start */
#define DODGE_OFFSET 3.0          /* csrc/obca_device.h: OBCA_DODGE_OFFSET */
#define DODGE_RAMP 3
#define DODGE_MIN_SPARE 0.1       /* csrc/obca_device.h: OBCA_DODGE_MIN_SPARE */
/* index: the EFFECTIVE order 1, 2, 3 (csrc/obca_device.h: OBCA_EFFECTIVE_ORDER; 0 = default: 1 -- the window first -- for every variant, 3 for a single-start call) */
static const int START_ORDERS[4][3] = {{KIND_X0, KIND_WINDOW, KIND_ZEROS}, {KIND_WINDOW, KIND_X0, KIND_ZEROS}, {KIND_ZEROS, KIND_WINDOW, KIND_X0}, {KIND_X0, KIND_WINDOW, KIND_ZEROS}};
#define KAPPA_MU 0.2
#define THETA_MU 1.5
#define KAPPA_EPS 10.0
#define TAU_MIN 0.99
#define BPUSH 1e-2
#define KAPPA_D 1e-5
#define KAPPA_SIGMA 1e10
#define KAPPA_SOC 0.99
#define S_MAX 100.0
#define GAMMA_THETA 1e-5
#define GAMMA_PHI 1e-8
#define S_THETA 1.1
#define S_PHI 2.3
#define ETA_PHI 1e-8
#define GAMMA_ALPHA 0.05
#define DW_MIN 1e-20
#define DW_0 1e-4
#define DW_MAX 1e40
#define KW_PLUS 8.0
#define KW_PLUS_BAR 100.0
#define KW_MINUS (1.0 / 3.0)

typedef struct { double *s, *p, *n, *y, *zL, *zU, *zp, *zn, *lb, *ub, *w; int* eq; } Rows;

static double maxabs(const double* v, int n) { double m = 0; for (int i = 0; i < n; ++i) m = fmax(m, fabs(v[i])); return m; }

/* one pass of the ladder: start `kind` (oracle/ipm_dense.py: x0_start, window_start, or the reference's all-zero start) with barrier
   parameter mu0 and at most iter_cap iterations (beside max_iter_*) */
static int solve_one(Prob* p, const Opts* o, double* xout, double* uout, double* ts, int* iters, double* info, int kind, double mu0, int iter_cap) {
    const int from_window = kind == KIND_WINDOW;
    const int n = p->n, mh = p->mh, me = p->me, na = p->naug, N = p->N;
    const int nk = n + mh + na;
    /* T rows carry multiplicity N+1 in the dense layout by being repeated (like oracle/obca_nlp.py) */
    size_t dbl = (size_t)n * 10 + (size_t)me * 29 + (size_t)mh * 7 + (size_t)me * n + (size_t)mh * n + (size_t)n * n +
                 (size_t)nk * nk + (size_t)nk * 2 + 64;
    double* mem = (double*)calloc(dbl, sizeof(double));
    int* piv = (int*)malloc(sizeof(int) * nk);
    int* eq = (int*)malloc(sizeof(int) * me);
    if (!mem || !piv || !eq) { free(mem); free(piv); free(eq); return ST_NUMERIC; }
    double* q = mem;
#define TAKE(c) (q += (c), q - (c))
    double *x = TAKE(n), *xt = TAKE(n), *dx = TAKE(n), *g = TAKE(n), *rx = TAKE(n), *bx = TAKE(n), *tmpn = TAKE(n), *gt_ = TAKE(n);
    double *s = TAKE(me), *ep = TAKE(me), *en = TAKE(me), *y = TAKE(me), *zL = TAKE(me), *zU = TAKE(me), *zp = TAKE(me), *zn = TAKE(me);
    double *ge = TAKE(me), *lb = TAKE(me), *ub = TAKE(me), *E = TAKE(me), *gh = TAKE(me), *dy = TAKE(me), *get = TAKE(me);
    double *Ds = TAKE(me), *Dp = TAKE(me), *Dn = TAKE(me), *rs = TAKE(me), *rp = TAKE(me), *rn = TAKE(me), *gs = TAKE(me);
    double *st_ = TAKE(me), *pt_ = TAKE(me);
    double *nt_ = TAKE(me), *gsoc = TAKE(me), *ghs = TAKE(me), *dys = TAKE(me), *get2 = TAKE(me);
    double *csoc = TAKE(mh), *dyhs = TAKE(mh), *cht2 = TAKE(mh), *dxs = TAKE(n), *xt2 = TAKE(n);
    double *yh = TAKE(mh), *dyh = TAKE(mh), *ch = TAKE(mh), *cht = TAKE(mh);
    double *Je = TAKE((size_t)me * n), *Jh = TAKE((size_t)mh * n), *W = TAKE((size_t)n * n), *K = TAKE((size_t)nk * nk), *rhs = TAKE(nk), *sol = TAKE(nk);
    (void)gt_; (void)tmpn;
    /* the line-search filter is unbounded, as IPOPT's (the kernels hold 64 / 128 entries: csrc/obca_device.h OBCA_FILTER_CAP) */
    int filt_room = 256;
    double* filt_t = (double*)malloc(sizeof(double) * 2 * filt_room);
    double* filt_p = filt_t + filt_room;
    int nfilt = 0, status = ST_MAXITER, it = 0, nfact = 0;
    if (!filt_t) { free(mem); free(piv); free(eq); return ST_NUMERIC; }
    for (int i = 0; i < n; ++i) x[i] = 0;
    if (p->freeT) x[iT(p)] = 1.0;
    if (kind == KIND_X0) for (int k = 0; k <= N; ++k) for (int j = 0; j < 3; ++j) x[ip(p, k) + j] = p->x0[j];
    if (from_window) {
        const int N1 = N + 1;
        for (int k = 0; k <= N; ++k) for (int j = 0; j < 3; ++j) x[ip(p, k) + j] = k == 0 ? p->x0[j] : p->xref[j * N1 + k];
        if (p->freeT) {
            double len = 0.0;
            for (int k = 0; k < N; ++k) {
                const double ddx = x[ip(p, k + 1)] - x[ip(p, k)], ddy = x[ip(p, k + 1) + 1] - x[ip(p, k) + 1];
                len += sqrt(ddx * ddx + ddy * ddy);
            }
            x[iT(p)] = fmin(fmax(1.0, len / (N * WINDOW_SPEED_FRAC * p->uU[0] * p->Ts)), fmax(1.0, p->Tmax));
        }
        const double h = p->Ts * (p->freeT ? x[iT(p)] : 1.0);
        for (int k = 0; k < N; ++k) {
            const double ddx = x[ip(p, k + 1)] - x[ip(p, k)], ddy = x[ip(p, k + 1) + 1] - x[ip(p, k) + 1];
            const double dth = x[ip(p, k + 1) + 2] - x[ip(p, k) + 2];
            x[iu(p, k)] = fmin(fmax(sqrt(ddx * ddx + ddy * ddy) / h, p->uL[0]), p->uU[0]);
            x[iu(p, k) + 1] = fmin(fmax(dth / h, p->uL[1]), p->uU[1]);
        }
    }
    if (kind == KIND_DODGE_R || kind == KIND_DODGE_L) {      /* oracle/ipm_dense.py:dodge_start */
        const int N1 = N + 1;
        const double side = kind == KIND_DODGE_R ? -1.0 : 1.0;
        for (int k = 0; k <= N; ++k) {
            double px = k == 0 ? p->x0[0] : p->xref[0 * N1 + k], py = k == 0 ? p->x0[1] : p->xref[1 * N1 + k];
            if (k > 0) {
                const int ka = k - 1, kb = k + 1 <= N ? k + 1 : N;
                const double ax = p->xref[0 * N1 + kb] - (ka == 0 ? p->x0[0] : p->xref[0 * N1 + ka]);
                const double ay = p->xref[1 * N1 + kb] - (ka == 0 ? p->x0[1] : p->xref[1 * N1 + ka]);
                const double len = sqrt(ax * ax + ay * ay), th = p->xref[2 * N1 + k];
                const double nx = len > 1e-9 ? -ay / len : -sin(th), ny = len > 1e-9 ? ax / len : cos(th);
                const double w = side * DODGE_OFFSET * (k < DODGE_RAMP ? (double)k / DODGE_RAMP : 1.0);
                px += w * nx; py += w * ny;
            }
            x[ip(p, k)] = px; x[ip(p, k) + 1] = py;
        }
        x[ip(p, 0) + 2] = p->x0[2];
        for (int k = 1; k <= N; ++k) {
            const double prev = x[ip(p, k - 1) + 2];
            double th = prev;
            if (k < N) {
                const double ddx = x[ip(p, k + 1)] - x[ip(p, k)], ddy = x[ip(p, k + 1) + 1] - x[ip(p, k) + 1];
                if (ddx * ddx + ddy * ddy > 1e-18) {
                    double d = atan2(ddy, ddx) - prev;
                    d -= 6.283185307179586 * floor(d / 6.283185307179586 + 0.5);
                    th = prev + d;
                }
            }
            x[ip(p, k) + 2] = th;
        }
        if (p->freeT) {
            double len = 0.0;
            for (int k = 0; k < N; ++k) {
                const double ddx = x[ip(p, k + 1)] - x[ip(p, k)], ddy = x[ip(p, k + 1) + 1] - x[ip(p, k) + 1];
                len += sqrt(ddx * ddx + ddy * ddy);
            }
            x[iT(p)] = fmin(fmax(1.0, len / (N * WINDOW_SPEED_FRAC * p->uU[0] * p->Ts)), fmax(1.0, p->Tmax));
        }
        const double h = p->Ts * (p->freeT ? x[iT(p)] : 1.0);
        for (int k = 0; k < N; ++k) {
            const double ddx = x[ip(p, k + 1)] - x[ip(p, k)], ddy = x[ip(p, k + 1) + 1] - x[ip(p, k) + 1];
            const double dth = x[ip(p, k + 1) + 2] - x[ip(p, k) + 2];
            x[iu(p, k)] = fmin(fmax(sqrt(ddx * ddx + ddy * ddy) / h, p->uL[0]), p->uU[0]);
            x[iu(p, k) + 1] = fmin(fmax(dth / h, p->uL[1]), p->uU[1]);
        }
        for (int k = 0; k <= N; ++k) {
            const double th = x[ip(p, k) + 2], ct = cos(th), st = sin(th);
            const double tx = x[ip(p, k)] + ct * p->off, ty = x[ip(p, k) + 1] + st * p->off;
            for (int i = 0; i < p->nO; ++i) {
                const int o0 = p->offm[i], o1 = p->offm[i + 1];
                int jb = o0;
                double gb = -INF, mub[4] = {0, 0, 0, 0}, lb_ = 0.0;
                for (int j = o0; j < o1; ++j) {
                    const double a0 = p->A[((size_t)k * p->M + j) * 2], a1 = p->A[((size_t)k * p->M + j) * 2 + 1];
                    const double nrm = sqrt(a0 * a0 + a1 * a1);
                    if (!(nrm > 0.0)) continue;
                    const double v0 = a0 / nrm, v1 = a1 / nrm, r0 = ct * v0 + st * v1, r1 = -st * v0 + ct * v1;
                    const double m0 = fmax(-r0, 0.0), m1 = fmax(-r1, 0.0), m2 = fmax(r0, 0.0), m3 = fmax(r1, 0.0);
                    const double gap = -(p->g[0] * m0 + p->g[1] * m1 + p->g[2] * m2 + p->g[3] * m3) + (a0 * tx + a1 * ty - p->b[(size_t)k * p->M + j]) / nrm;
                    if (gap > gb) { gb = gap; jb = j; lb_ = 1.0 / nrm; mub[0] = m0; mub[1] = m1; mub[2] = m2; mub[3] = m3; }
                }
                for (int j = o0; j < o1; ++j) x[il(p, k) + j] = j == jb ? lb_ : 0.0;
                for (int q = 0; q < 4; ++q) x[imu(p, k) + 4 * i + q] = mub[q];
            }
        }
    }
    /* scaling */
    objective(p, x, g, NULL);
    double gmax = fmax(maxabs(g, n), o->rho);
    const double sf = gmax > 100.0 ? 100.0 / gmax : 1.0, rho = o->rho * sf;
    elastic_rows(p, x, ge, lb, ub, NULL, NULL, NULL);
    int bad_bounds = 0;
    double mu = mu0;
    for (int r = 0; r < me; ++r) {
        eq[r] = r < na;
        const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
        double sv = ge[r];
        if (eq[r]) sv = 0;
        else if (hasL && hasU) {
            if (!(lb[r] < ub[r])) bad_bounds = 1;
            const double pL = fmin(BPUSH * fmax(1, fabs(lb[r])), BPUSH * (ub[r] - lb[r]));
            const double pU = fmin(BPUSH * fmax(1, fabs(ub[r])), BPUSH * (ub[r] - lb[r]));
            sv = fmin(fmax(sv, lb[r] + pL), ub[r] - pU);
        } else if (hasL) sv = fmax(sv, lb[r] + BPUSH * fmax(1, fabs(lb[r])));
        else if (hasU) sv = fmin(sv, ub[r] - BPUSH * fmax(1, fabs(ub[r])));
        const double rr = ge[r] - sv, a = (mu - rho * rr) / (2 * rho);
        en[r] = a + sqrt(a * a + mu * rr / (2 * rho));
        ep[r] = rr + en[r];
        s[r] = sv; zp[r] = mu / ep[r]; zn[r] = mu / en[r]; y[r] = rho - zp[r];
        zL[r] = hasL ? 1 : 0; zU[r] = hasU ? 1 : 0;
    }
    double elastic = 0, E0 = INF, f = 0;
    if (bad_bounds) status = ST_BAD_BOUNDS;
    else {
        double theta_max = 0, theta_min = 0, dw_last = 0, tau = fmax(TAU_MIN, 1 - mu), fprev = 0;
        int have_prev = 0, acc = 0;
        const int max_iter_v = p->freeT ? o->max_iter_free : o->max_iter_fixed;
        const int max_iter = max_iter_v < iter_cap ? max_iter_v : iter_cap;
        const double acc_tol = p->freeT ? 1e-6 : 1e-8, acc_obj = p->freeT ? 1e20 : 1e-6;
        for (it = 0; it <= max_iter; ++it) {
            memset(Je, 0, sizeof(double) * (size_t)me * n);
            memset(Jh, 0, sizeof(double) * (size_t)mh * n);
            f = sf * objective(p, x, g, NULL);
            for (int i = 0; i < n; ++i) g[i] *= sf;
            elastic_rows(p, x, ge, NULL, NULL, Je, NULL, NULL);
            hard_rows(p, x, ch, Jh, NULL, NULL);
            /* errors */
            for (int i = 0; i < n; ++i) {
                double v = g[i];
                for (int r = 0; r < me; ++r) v += Je[(size_t)r * n + i] * y[r];
                for (int r = 0; r < mh; ++r) v += Jh[(size_t)r * n + i] * yh[r];
                rx[i] = v;
            }
            double th = 0, pnsum = 0;
            elastic = 0;
            for (int r = 0; r < mh; ++r) th += fabs(ch[r]);
            for (int r = 0; r < me; ++r) { th += fabs(ge[r] - s[r] - ep[r] + en[r]); pnsum += ep[r] + en[r]; elastic = fmax(elastic, ep[r] + en[r]); }
            double errs[2][4];
            for (int pass = 0; pass < 2; ++pass) {
                const double m_ = pass ? mu : 0.0;
                double dual = maxabs(rx, n), prim = maxabs(ch, mh), comp = 0, ysum = 0, zsum = 0, nz = 0;
                for (int r = 0; r < mh; ++r) ysum += fabs(yh[r]);
                for (int r = 0; r < me; ++r) {
                    const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
                    const double zl = hasL ? zL[r] : 0, zu = hasU ? zU[r] : 0;
                    if (!eq[r]) dual = fmax(dual, fabs(-y[r] - zl + zu));
                    dual = fmax(dual, fabs(rho - y[r] - zp[r])); dual = fmax(dual, fabs(rho + y[r] - zn[r]));
                    prim = fmax(prim, fabs(ge[r] - s[r] - ep[r] + en[r]));
                    comp = fmax(comp, fabs(ep[r] * zp[r] - m_)); comp = fmax(comp, fabs(en[r] * zn[r] - m_));
                    if (hasL) comp = fmax(comp, fabs((s[r] - lb[r]) * zl - m_));
                    if (hasU) comp = fmax(comp, fabs((ub[r] - s[r]) * zu - m_));
                    ysum += fabs(y[r]); zsum += zl + zu + zp[r] + zn[r]; nz += hasL + hasU + 2;
                }
                const double sd = fmax(S_MAX, (ysum + zsum) / (mh + me + nz)) / S_MAX, sc = fmax(S_MAX, zsum / nz) / S_MAX;
                errs[pass][0] = fmax(fmax(dual / sd, prim), comp / sc); errs[pass][1] = dual; errs[pass][2] = prim; errs[pass][3] = comp;
                if (pass == 1) {   /* monotone mu update loop needs comp recomputed per mu: handled below */ }
            }
            E0 = errs[0][0];
            if (it == 0) { theta_max = 1e4 * fmax(1, th); theta_min = 1e-4 * fmax(1, th); }
            if (E0 <= o->tol && errs[0][1] <= 1.0 && errs[0][2] <= 1e-4 && errs[0][3] <= 1e-4) { status = ST_OK; break; }
            const double fobj = f + rho * pnsum, objchg = have_prev ? fabs(fobj - fprev) / fmax(1, fabs(fobj)) : INF;
            if (E0 <= acc_tol && errs[0][1] <= 1e10 && errs[0][2] <= 1e-2 && errs[0][3] <= 1e-2 && objchg <= acc_obj) {
                if (++acc >= 15) { status = ST_ACCEPTABLE; break; }
            } else acc = 0;
            if (it == max_iter) break;
            /* barrier update */
            const double mu_floor = o->tol / (KAPPA_EPS + 1);
            while (mu > mu_floor) {
                double dual = errs[0][1], prim = errs[0][2], comp = 0, ysum = 0, zsum = 0, nz = 0;
                for (int r = 0; r < mh; ++r) ysum += fabs(yh[r]);
                for (int r = 0; r < me; ++r) {
                    const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
                    const double zl = hasL ? zL[r] : 0, zu = hasU ? zU[r] : 0;
                    comp = fmax(comp, fabs(ep[r] * zp[r] - mu)); comp = fmax(comp, fabs(en[r] * zn[r] - mu));
                    if (hasL) comp = fmax(comp, fabs((s[r] - lb[r]) * zl - mu));
                    if (hasU) comp = fmax(comp, fabs((ub[r] - s[r]) * zu - mu));
                    ysum += fabs(y[r]); zsum += zl + zu + zp[r] + zn[r]; nz += hasL + hasU + 2;
                }
                const double sd = fmax(S_MAX, (ysum + zsum) / (mh + me + nz)) / S_MAX, sc = fmax(S_MAX, zsum / nz) / S_MAX;
                if (fmax(fmax(dual / sd, prim), comp / sc) > KAPPA_EPS * mu) break;
                mu = fmax(mu_floor, fmin(KAPPA_MU * mu, pow(mu, THETA_MU)));
                tau = fmax(TAU_MIN, 1 - mu);
                nfilt = 0;
            }
            /* Lagrangian Hessian */
            memset(W, 0, sizeof(double) * (size_t)n * n);
            objective(p, x, NULL, W);
            for (size_t i = 0; i < (size_t)n * n; ++i) W[i] *= sf;
            elastic_rows(p, x, NULL, NULL, NULL, NULL, y, W);
            hard_rows(p, x, NULL, NULL, yh, W);
            /* inertia-corrected factorisation */
            double dw = 0;
            int first = 1, fail = 0;
            for (;;) {
                for (int r = 0; r < me; ++r) {
                    const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
                    double sig = 0, gsv = 0;
                    if (hasL) { sig += zL[r] / (s[r] - lb[r]); gsv -= mu / (s[r] - lb[r]); }
                    if (hasU) { sig += zU[r] / (ub[r] - s[r]); gsv += mu / (ub[r] - s[r]); }
                    if (hasL && !hasU) gsv += KAPPA_D * mu;
                    if (hasU && !hasL) gsv -= KAPPA_D * mu;
                    gs[r] = gsv; Ds[r] = sig + dw; Dp[r] = zp[r] / ep[r] + dw; Dn[r] = zn[r] / en[r] + dw;
                    rs[r] = eq[r] ? 0 : -y[r] + gsv; rp[r] = rho - y[r] - mu / ep[r]; rn[r] = rho + y[r] - mu / en[r];
                    E[r] = (eq[r] ? 0 : 1 / Ds[r]) + 1 / Dp[r] + 1 / Dn[r];
                    gh[r] = (ge[r] - s[r] - ep[r] + en[r]) + (eq[r] ? 0 : rs[r] / Ds[r]) + rp[r] / Dp[r] - rn[r] / Dn[r];
                }
                memset(K, 0, sizeof(double) * (size_t)nk * nk);
                for (int i = 0; i < n; ++i) {
                    for (int j = 0; j <= i; ++j) K[(size_t)i * nk + j] = W[(size_t)i * n + j];
                    K[(size_t)i * nk + i] += dw;
                }
                if (p->freeT) K[(size_t)iT(p) * nk + iT(p)] += dw * N;
                for (int i = 0; i < n; ++i) rhs[i] = -rx[i];
                for (int r = na; r < me; ++r) {             /* condensed rows */
                    const double* Jr = Je + (size_t)r * n;
                    const double ei = 1 / E[r];
                    for (int i = 0; i < n; ++i) {
                        if (Jr[i] == 0) continue;
                        const double v = Jr[i] * ei;
                        rhs[i] -= v * gh[r];
                        for (int j = 0; j <= i; ++j) if (Jr[j] != 0) K[(size_t)i * nk + j] += v * Jr[j];
                    }
                }
                for (int r = 0; r < mh; ++r) { for (int j = 0; j < n; ++j) K[(size_t)(n + r) * nk + j] = Jh[(size_t)r * n + j]; rhs[n + r] = -ch[r]; }
                for (int r = 0; r < na; ++r) {
                    for (int j = 0; j < n; ++j) K[(size_t)(n + mh + r) * nk + j] = Je[(size_t)r * n + j];
                    K[(size_t)(n + mh + r) * nk + n + mh + r] = -E[r];
                    rhs[n + mh + r] = -gh[r];
                }
                const int neg = bk_factor(K, nk, piv);
                ++nfact;
                int zero = 0;
                for (int i = 0; i < nk; ++i) if (piv[i] >= 0 && K[(size_t)i * nk + i] == 0.0) zero = 1;
                if (neg == mh + na && !zero) break;
                if (first) { dw = dw_last == 0 ? DW_0 : fmax(DW_MIN, KW_MINUS * dw_last); first = 0; }
                else dw *= dw_last == 0 ? KW_PLUS_BAR : KW_PLUS;
                if (dw > DW_MAX) { fail = 1; break; }
            }
            if (fail) { status = ST_NUMERIC; break; }
            if (dw > 0) dw_last = dw;
            memcpy(sol, rhs, sizeof(double) * nk);
            bk_solve(K, nk, piv, sol);
            memcpy(dx, sol, sizeof(double) * n);
            memcpy(dyh, sol + n, sizeof(double) * mh);
            double a_max = 1, a_z = 1, dphi = 0, phi = f;
            for (int i = 0; i < n; ++i) dphi += g[i] * dx[i];
            for (int r = 0; r < me; ++r) {
                if (r < na) dy[r] = sol[n + mh + r];
                else { double v = gh[r]; const double* Jr = Je + (size_t)r * n; for (int i = 0; i < n; ++i) v += Jr[i] * dx[i]; dy[r] = v / E[r]; }
                const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
                const double ds = eq[r] ? 0 : (dy[r] - rs[r]) / Ds[r], dp = (dy[r] - rp[r]) / Dp[r], dn = (-dy[r] - rn[r]) / Dn[r];
                if (hasL) {
                    const double sl = s[r] - lb[r];
                    if (ds < 0) a_max = fmin(a_max, -tau * sl / ds);
                    const double dz = (mu - zL[r] * ds) / sl - zL[r];
                    if (dz < 0) a_z = fmin(a_z, -tau * zL[r] / dz);
                    phi -= mu * log(sl); if (!hasU) phi += KAPPA_D * mu * sl;
                }
                if (hasU) {
                    const double su = ub[r] - s[r];
                    if (ds > 0) a_max = fmin(a_max, tau * su / ds);
                    const double dz = (mu + zU[r] * ds) / su - zU[r];
                    if (dz < 0) a_z = fmin(a_z, -tau * zU[r] / dz);
                    phi -= mu * log(su); if (!hasL) phi += KAPPA_D * mu * su;
                }
                if (dp < 0) a_max = fmin(a_max, -tau * ep[r] / dp);
                if (dn < 0) a_max = fmin(a_max, -tau * en[r] / dn);
                const double dzp = (mu - zp[r] * dp) / ep[r] - zp[r], dzn = (mu - zn[r] * dn) / en[r] - zn[r];
                if (dzp < 0) a_z = fmin(a_z, -tau * zp[r] / dzp);
                if (dzn < 0) a_z = fmin(a_z, -tau * zn[r] / dzn);
                phi += rho * (ep[r] + en[r]) - mu * (log(ep[r]) + log(en[r]));
                dphi += gs[r] * ds + (rho - mu / ep[r]) * dp + (rho - mu / en[r]) * dn;
            }
            double alpha_min;
            if (dphi < 0) {
                double c = fmin(GAMMA_THETA, GAMMA_PHI * th / (-dphi));
                if (th <= theta_min) c = fmin(c, pow(th, S_THETA) / pow(-dphi, S_PHI));
                alpha_min = GAMMA_ALPHA * c;
            } else alpha_min = GAMMA_ALPHA * GAMMA_THETA;
            double alpha = a_max;
            int accepted = 0, aug = 0, first_trial = 1, use_soc = 0;
            double a_used = a_max;
            for (;;) {
                for (int i = 0; i < n; ++i) xt[i] = x[i] + alpha * dx[i];
                const double ft = sf * objective(p, xt, NULL, NULL);
                elastic_rows(p, xt, get, NULL, NULL, NULL, NULL, NULL);
                hard_rows(p, xt, cht, NULL, NULL, NULL);
                double th_t = 0, phi_t = ft;
                for (int r = 0; r < mh; ++r) th_t += fabs(cht[r]);
                for (int r = 0; r < me; ++r) {
                    const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
                    const double sv = eq[r] ? 0 : s[r] + alpha * (dy[r] - rs[r]) / Ds[r];
                    const double pv = ep[r] + alpha * (dy[r] - rp[r]) / Dp[r], nv = en[r] + alpha * (-dy[r] - rn[r]) / Dn[r];
                    st_[r] = sv; pt_[r] = pv; nt_[r] = nv;
                    th_t += fabs(get[r] - sv - pv + nv);
                    phi_t += rho * (pv + nv) - mu * (log(pv) + log(nv));
                    if (hasL) { phi_t -= mu * log(sv - lb[r]); if (!hasU) phi_t += KAPPA_D * mu * (sv - lb[r]); }
                    if (hasU) { phi_t -= mu * log(ub[r] - sv); if (!hasL) phi_t += KAPPA_D * mu * (ub[r] - sv); }
                }
                int ok = 0;
                aug = 0;
                int blocked = th_t >= theta_max;
                for (int i = 0; i < nfilt && !blocked; ++i) if (th_t >= filt_t[i] && phi_t >= filt_p[i]) blocked = 1;
                if (isfinite(phi_t) && isfinite(th_t) && !blocked) {
                    const int sw = dphi < 0 && alpha * pow(-dphi, S_PHI) > pow(th, S_THETA);
                    if (th <= theta_min && sw) ok = phi_t <= phi + ETA_PHI * alpha * dphi + 10 * 2.220446049250313e-16 * fabs(phi);
                    else { ok = (th_t <= (1 - GAMMA_THETA) * th) || (phi_t <= phi - GAMMA_PHI * th); aug = ok; }
                }
                if (ok) { accepted = 1; a_used = alpha; break; }
                /* second-order correction (IPOPT max_soc = 4, kappa_soc = 0.99): only after the FIRST trial step, and only
                   when it did not reduce the constraint violation; same matrix (same delta_w), corrected right-hand side */
                int evals_finite = isfinite(ft);
                for (int r = 0; r < me && evals_finite; ++r) if (!isfinite(get[r])) evals_finite = 0;
                for (int r = 0; r < mh && evals_finite; ++r) if (!isfinite(cht[r])) evals_finite = 0;
                if (first_trial && o->max_soc > 0 && th_t >= th && evals_finite) {
                    for (int r = 0; r < mh; ++r) csoc[r] = alpha * ch[r] + cht[r];
                    for (int r = 0; r < me; ++r) gsoc[r] = alpha * (ge[r] - s[r] - ep[r] + en[r]) + (get[r] - st_[r] - pt_[r] + nt_[r]);
                    double th_old = th_t;
                    for (int ps = 0; ps < o->max_soc; ++ps) {
                        for (int r = 0; r < me; ++r)
                            ghs[r] = gsoc[r] + (eq[r] ? 0 : rs[r] / Ds[r]) + rp[r] / Dp[r] - rn[r] / Dn[r];
                        for (int i = 0; i < n; ++i) sol[i] = -rx[i];
                        for (int r = na; r < me; ++r) {
                            const double* Jr = Je + (size_t)r * n;
                            const double ei = 1 / E[r];
                            for (int i = 0; i < n; ++i) {
                                if (Jr[i] == 0) continue;
                                const double v = Jr[i] * ei;
                                sol[i] -= v * ghs[r];
                            }
                        }
                        for (int r = 0; r < mh; ++r) sol[n + r] = -csoc[r];
                        for (int r = 0; r < na; ++r) sol[n + mh + r] = -ghs[r];
                        bk_solve(K, nk, piv, sol);
                        memcpy(dxs, sol, sizeof(double) * n);
                        memcpy(dyhs, sol + n, sizeof(double) * mh);
                        double a_soc = 1;
                        for (int r = 0; r < me; ++r) {
                            if (r < na) dys[r] = sol[n + mh + r];
                            else { double v = ghs[r]; const double* Jr = Je + (size_t)r * n; for (int i = 0; i < n; ++i) v += Jr[i] * dxs[i]; dys[r] = v / E[r]; }
                            const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
                            const double ds = eq[r] ? 0 : (dys[r] - rs[r]) / Ds[r], dp = (dys[r] - rp[r]) / Dp[r], dn = (-dys[r] - rn[r]) / Dn[r];
                            if (hasL && ds < 0) a_soc = fmin(a_soc, -tau * (s[r] - lb[r]) / ds);
                            if (hasU && ds > 0) a_soc = fmin(a_soc, tau * (ub[r] - s[r]) / ds);
                            if (dp < 0) a_soc = fmin(a_soc, -tau * ep[r] / dp);
                            if (dn < 0) a_soc = fmin(a_soc, -tau * en[r] / dn);
                        }
                        for (int i = 0; i < n; ++i) xt2[i] = x[i] + a_soc * dxs[i];
                        const double fs = sf * objective(p, xt2, NULL, NULL);
                        elastic_rows(p, xt2, get2, NULL, NULL, NULL, NULL, NULL);
                        hard_rows(p, xt2, cht2, NULL, NULL, NULL);
                        int fin = isfinite(fs);
                        for (int r = 0; r < me && fin; ++r) if (!isfinite(get2[r])) fin = 0;
                        for (int r = 0; r < mh && fin; ++r) if (!isfinite(cht2[r])) fin = 0;
                        double th_s = 0, phi_s = fs;
                        for (int r = 0; r < mh; ++r) th_s += fabs(cht2[r]);
                        for (int r = 0; r < me; ++r) {
                            const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
                            const double sv = eq[r] ? 0 : s[r] + a_soc * (dys[r] - rs[r]) / Ds[r];
                            const double pv = ep[r] + a_soc * (dys[r] - rp[r]) / Dp[r], nv = en[r] + a_soc * (-dys[r] - rn[r]) / Dn[r];
                            st_[r] = sv; pt_[r] = pv; nt_[r] = nv;
                            th_s += fabs(get2[r] - sv - pv + nv);
                            phi_s += rho * (pv + nv) - mu * (log(pv) + log(nv));
                            if (hasL) { phi_s -= mu * log(sv - lb[r]); if (!hasU) phi_s += KAPPA_D * mu * (sv - lb[r]); }
                            if (hasU) { phi_s -= mu * log(ub[r] - sv); if (!hasL) phi_s += KAPPA_D * mu * (ub[r] - sv); }
                        }
                        if (!fin) { th_s = INF; phi_s = INF; }
                        int ok_s = 0, aug_s = 0;
                        int blk = th_s >= theta_max;
                        for (int i = 0; i < nfilt && !blk; ++i) if (th_s >= filt_t[i] && phi_s >= filt_p[i]) blk = 1;
                        if (isfinite(phi_s) && isfinite(th_s) && !blk) {
                            const int sw = dphi < 0 && alpha * pow(-dphi, S_PHI) > pow(th, S_THETA);
                            if (th <= theta_min && sw) ok_s = phi_s <= phi + ETA_PHI * alpha * dphi + 10 * 2.220446049250313e-16 * fabs(phi);
                            else { ok_s = (th_s <= (1 - GAMMA_THETA) * th) || (phi_s <= phi - GAMMA_PHI * th); aug_s = ok_s; }
                        }
                        if (ok_s) { accepted = 1; use_soc = 1; aug = aug_s; a_used = a_soc; break; }
                        if (!fin || th_s > KAPPA_SOC * th_old) break;
                        th_old = th_s;
                        for (int r = 0; r < mh; ++r) csoc[r] = a_soc * csoc[r] + cht2[r];
                        for (int r = 0; r < me; ++r) gsoc[r] = a_soc * gsoc[r] + (get2[r] - st_[r] - pt_[r] + nt_[r]);
                    }
                    if (accepted) break;
                }
                first_trial = 0;
                alpha *= 0.5;
                if (alpha < alpha_min) break;
            }
            if (!accepted) { status = ST_LINESEARCH; break; }
            if (aug) {
                const double tn = (1 - GAMMA_THETA) * th, pn = phi - GAMMA_PHI * th;
                int w = 0;
                for (int i = 0; i < nfilt; ++i) if (!(filt_t[i] >= tn && filt_p[i] >= pn)) { filt_t[w] = filt_t[i]; filt_p[w] = filt_p[i]; ++w; }
                nfilt = w;
                if (nfilt >= filt_room) {
                    double* nf = (double*)malloc(sizeof(double) * 4 * filt_room);
                    if (!nf) { status = ST_NUMERIC; break; }
                    memcpy(nf, filt_t, sizeof(double) * nfilt); memcpy(nf + 2 * filt_room, filt_p, sizeof(double) * nfilt);
                    free(filt_t);
                    filt_t = nf; filt_p = nf + 2 * filt_room; filt_room *= 2;
                }
                filt_t[nfilt] = tn; filt_p[nfilt] = pn; ++nfilt;
            }
            for (int r = 0; r < me; ++r) {
                const int hasL = !eq[r] && lb[r] > -INF, hasU = !eq[r] && ub[r] < INF;
                /* the bound multipliers always follow the ORIGINAL direction (step a_z); the primal variables and y the
                   accepted one -- the second-order-corrected direction with its own step length when that was taken */
                const double ds = eq[r] ? 0 : (dy[r] - rs[r]) / Ds[r], dp = (dy[r] - rp[r]) / Dp[r], dn = (-dy[r] - rn[r]) / Dn[r];
                const double dyu = use_soc ? dys[r] : dy[r];
                const double dsu = eq[r] ? 0 : (dyu - rs[r]) / Ds[r], dpu = (dyu - rp[r]) / Dp[r], dnu = (-dyu - rn[r]) / Dn[r];
                const double so = s[r], po = ep[r], no = en[r];
                s[r] = eq[r] ? 0 : so + a_used * dsu; ep[r] = po + a_used * dpu; en[r] = no + a_used * dnu;
                if (hasL) { const double z = zL[r] + a_z * ((mu - zL[r] * ds) / (so - lb[r]) - zL[r]), sl = s[r] - lb[r]; zL[r] = fmax(fmin(z, KAPPA_SIGMA * mu / sl), mu / (KAPPA_SIGMA * sl)); }
                if (hasU) { const double z = zU[r] + a_z * ((mu + zU[r] * ds) / (ub[r] - so) - zU[r]), su = ub[r] - s[r]; zU[r] = fmax(fmin(z, KAPPA_SIGMA * mu / su), mu / (KAPPA_SIGMA * su)); }
                const double z1 = zp[r] + a_z * ((mu - zp[r] * dp) / po - zp[r]), z2 = zn[r] + a_z * ((mu - zn[r] * dn) / no - zn[r]);
                zp[r] = fmax(fmin(z1, KAPPA_SIGMA * mu / ep[r]), mu / (KAPPA_SIGMA * ep[r]));
                zn[r] = fmax(fmin(z2, KAPPA_SIGMA * mu / en[r]), mu / (KAPPA_SIGMA * en[r]));
                y[r] += a_used * dyu;
            }
            for (int r = 0; r < mh; ++r) yh[r] += a_used * (use_soc ? dyhs[r] : dyh[r]);
            memcpy(x, use_soc ? xt2 : xt, sizeof(double) * n);
            fprev = fobj; have_prev = 1;
        }
    }
    if ((status == ST_OK || status == ST_ACCEPTABLE) && elastic > o->feas_tol) status = ST_INFEASIBLE;
    for (int j = 0; j < 3; ++j) for (int k = 0; k <= N; ++k) xout[j * (N + 1) + k] = x[ip(p, k) + j];
    for (int j = 0; j < 2; ++j) for (int k = 0; k < N; ++k) uout[j * N + k] = x[iu(p, k) + j];
    *ts = p->freeT ? x[iT(p)] * p->Ts : p->Ts;
    *iters = it;
    if (info) { info[0] = objective(p, x, NULL, NULL); info[1] = elastic; info[2] = E0; info[3] = nfact; }
    (void)bx;
    free(mem); free(piv); free(eq); free(filt_t);
    return status;
}

typedef struct {
    double Qf[9], Pf[9], R1f[4], R2f[4], Qx[9], Px[9], R1x[4], R2x[4];
    double xL[2], xU[2], uL[2], uU[2], ego[4], dmin, tol, rho, feas_tol;
    int max_iter_free, max_iter_fixed;
    int max_soc;                            /* 0 = IPOPT's default (4), negative = off */
    int start_order, single_start, patience, retry_iter;     /* as obca_params (include/obca_mpc.h) */
    int dodge, terminal_screen;                              /* likewise: 0 = the default (on), negative = off */
} OracleParams;

static void sym(double* d, const double* s, int k) { for (int a = 0; a < k; ++a) for (int b = 0; b < k; ++b) d[k * a + b] = 0.5 * (s[k * a + b] + s[k * b + a]); }

/* same array layout as obca_solve_batch in include/obca_mpc.h, HOST pointers */
int obca_oracle_solve_batch(int N, int n_obs, const int* m, const int* variant, int B,
                            const double* x0, const double* u0, const double* xref, const double* A, const double* b,
                            const double* Ts, const double* term, const OracleParams* prm,
                            double* xopt, double* uopt, double* ts_opt, int* status, int* iters, double* info,
                            int threads) {
    int M = 0;
    for (int i = 0; i < n_obs; ++i) M += m[i];
    (void)threads;
    /* as csrc/obca_device.h: obca_resolve_starts -- what obca_solve_batch answers with OBCA_E_INVAL */
    if (prm->start_order < 0 || prm->start_order > 3 || prm->single_start < 0 || prm->single_start > 1) return -22;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int q = 0; q < B; ++q) {
        Prob p;
        memset(&p, 0, sizeof(p));
        p.variant = variant[q]; p.N = N; p.nO = n_obs; p.M = M;
        p.offm[0] = 0;
        for (int i = 0; i < n_obs; ++i) { p.m[i] = m[i]; p.offm[i + 1] = p.offm[i] + m[i]; }
        p.freeT = p.variant == 4;
        p.NS = 5 + M + 4 * n_obs;
        p.n = (N + 1) * (3 + M + 4 * n_obs) + 2 * N + p.freeT;
        memcpy(p.x0, x0 + (size_t)q * 3, 24); memcpy(p.u0, u0 + (size_t)q * 2, 16);
        p.Ts = Ts[q];
        if (p.variant == 6) memcpy(p.term, term + (size_t)q * 3, 24);
        p.xref = xref + (size_t)q * 3 * (N + 1);
        double* Arep = NULL; double* brep = NULL;
        if (p.variant == 4) {          /* q5: mpc4 reads the step-0 rows at every k */
            Arep = (double*)malloc(sizeof(double) * (N + 1) * M * 2); brep = (double*)malloc(sizeof(double) * (N + 1) * M);
            for (int k = 0; k <= N; ++k) { memcpy(Arep + (size_t)k * M * 2, A + (size_t)q * (N + 1) * M * 2, sizeof(double) * M * 2); memcpy(brep + (size_t)k * M, b + (size_t)q * (N + 1) * M, sizeof(double) * M); }
            p.A = Arep; p.b = brep;
        } else { p.A = A + (size_t)q * (N + 1) * M * 2; p.b = b + (size_t)q * (N + 1) * M; }
        sym(p.Q, p.freeT ? prm->Qf : prm->Qx, 3); sym(p.P, p.freeT ? prm->Pf : prm->Px, 3);
        sym(p.R1, p.freeT ? prm->R1f : prm->R1x, 2); sym(p.R2, p.freeT ? prm->R2f : prm->R2x, 2);
        for (int j = 0; j < 2; ++j) { p.xL[j] = prm->xL[j]; p.xU[j] = prm->xU[j]; p.uL[j] = prm->uL[j]; p.uU[j] = prm->uU[j]; }
        const double Lc = prm->ego[0] + prm->ego[2], Wc = prm->ego[1] + prm->ego[3];
        p.g[0] = Lc / 2; p.g[1] = Wc / 2; p.g[2] = Lc / 2; p.g[3] = Wc / 2; p.off = Lc / 2 - prm->ego[2]; p.dmin = prm->dmin;
        if (p.freeT) {
            const double dis = (p.xref[0 * (N + 1) + N] - p.x0[0]) + (p.xref[1 * (N + 1) + N] - p.x0[1]);
            p.Tmax = dis / (N * p.uU[0] * p.Ts) + 1.0;
        }
        p.mh = 2 * (N + 1) * n_obs;
        p.naug = 3 + 3 * N + (p.variant == 4 ? 3 : 0);
        p.nineq = 2 * (N + 1) + 2 * N + 2 * N + (p.variant == 4 ? 2 * (N + 1) : 0) + (p.variant == 6 ? 2 : 0) +
                  2 * (N + 1) * n_obs + (N + 1) * (M + 4 * n_obs);
        p.me = p.naug + p.nineq;
        Opts o;
        o.tol = prm->tol > 0 ? prm->tol : 1e-8; o.rho = prm->rho > 0 ? prm->rho : 1e4; o.feas_tol = prm->feas_tol > 0 ? prm->feas_tol : 1e-6;
        o.max_iter_free = prm->max_iter_free > 0 ? prm->max_iter_free : 3000; o.max_iter_fixed = prm->max_iter_fixed > 0 ? prm->max_iter_fixed : 1000;
        o.max_soc = prm->max_soc == 0 ? 4 : (prm->max_soc < 0 ? 0 : prm->max_soc);
        /* the start ladder (oracle/ipm_dense.py:solve): the starts of the order until one ends at a feasible point; obca_mpc4 that
           converged with elastic variables left repeats the same start with rho x 100 and, if elastic variables still remain, with rho x 1000
           (the next start begins at the base penalty) */
        const int order0 = prm->start_order;
        const int order = order0 != 0 ? order0 : (prm->single_start ? 3 : 1);      /* csrc/obca_device.h: OBCA_EFFECTIVE_ORDER */
        const int nstarts = prm->single_start ? 1 : 3;
        const int max_iter_v = p.freeT ? o.max_iter_free : o.max_iter_fixed;
        const int pat = prm->patience > 0 ? prm->patience : PATIENCE(N), ret = prm->retry_iter > 0 ? prm->retry_iter : RETRY_ITER(N);
        double* xo = xopt + (size_t)q * 3 * (N + 1); double* uo = uopt + (size_t)q * 2 * N;
        double* io = info ? info + (size_t)q * 4 : NULL;
        int it_sum = 0;
        double nf_sum = 0.0;
        status[q] = ST_MAXITER;
        double shortfall = -INF;                                  /* oracle/ipm_dense.py:terminal_set_shortfall (-inf unless obca_mpc6) */
        if (p.variant == 6) {
            double vhi = p.u0[0], vlo = p.u0[0], reach = 0.0;
            const double c0 = cos(p.x0[2]);
            for (int k = 0; k < N; ++k) {
                vhi = fmin(p.uU[0], vhi + 0.6 * p.Ts);
                vlo = fmax(p.uL[0], vlo - 0.6 * p.Ts);
                reach += p.Ts * (k == 0 ? fmax(vhi * c0, vlo * c0) : fmax(fabs(vhi), fabs(vlo)));
            }
            const double xN = fmin(p.x0[0] + reach, p.xU[0]);
            const double sh = p.term[0] - xN - 2.0 * o.feas_tol * (N + 2 + N * p.Ts + p.Ts * p.Ts * N * (N + 1) / 2.0);
            shortfall = sh;
            if (sh > 0.0 && prm->terminal_screen >= 0) {
                for (int j = 0; j < 3; ++j) for (int k = 0; k <= N; ++k) xo[j * (N + 1) + k] = p.x0[j];
                for (int t = 0; t < 2 * N; ++t) uo[t] = 0.0;
                ts_opt[q] = p.Ts; status[q] = ST_INFEASIBLE; iters[q] = 0;
                if (io) { io[0] = 0.0; io[1] = sh; io[2] = 0.0; io[3] = 0.0; }
                free(Arep); free(brep);
                continue;
            }
        }
        /* which pass's answer an exhausted ladder returns (oracle/ipm_dense.py: _replaces): a feasible one always; otherwise the
           first, replaced only by a later pass that converged ("infeasible") where the held one did not, or by the same start's
           repetition with a raised penalty */
        {
            double* xp_ = (double*)malloc(sizeof(double) * (3 * (N + 1) + 2 * N));
            double* up_ = xp_ + 3 * (N + 1);
            int last = ST_MAXITER, held = 0, held_start = -1;
            for (int s = 0; s < nstarts && xp_; ++s) {
                if (s > 0 && (last == ST_OK || last == ST_ACCEPTABLE || last == ST_BAD_BOUNDS)) break;
                const int kind = START_ORDERS[order][s];
                const int cap = nstarts == 1 ? max_iter_v : (s == 0 ? pat : ret);
                const double mu0 = kind == KIND_WINDOW ? RESTART_MU : MU_INIT;
                for (int level = 0; level <= 2; ++level) {
                    if (level > 0 && !(last == ST_INFEASIBLE && p.variant == 4)) break;
                    Opts o2 = o;
                    if (level) o2.rho = o.rho * (level == 1 ? 100.0 : 1000.0);         /* csrc/obca_device.h: OBCA_RHO_ESCALATION */
                    double tst, inf_[4];
                    int itr;
                    last = solve_one(&p, &o2, xp_, up_, &tst, &itr, inf_, kind, mu0, cap);
                    it_sum += itr; nf_sum += inf_[3];
                    const int good = last == ST_OK || last == ST_ACCEPTABLE || last == ST_BAD_BOUNDS;
                    if (!held || good || (last == ST_INFEASIBLE && (status[q] != ST_INFEASIBLE || held_start == s))) {
                        memcpy(xo, xp_, sizeof(double) * 3 * (N + 1)); memcpy(uo, up_, sizeof(double) * 2 * N);
                        ts_opt[q] = tst; status[q] = last;
                        if (io) { io[0] = inf_[0]; io[1] = inf_[1]; io[2] = inf_[2]; }
                        held = 1; held_start = s;
                    }
                }
            }
            free(xp_);
        }
        /* the dodge rung (oracle/ipm_dense.py:solve): fixed-time problems after the order is exhausted; both sides run, the
           feasible answer with the lower objective stays */
        if (p.variant != 4 && prm->dodge >= 0 && shortfall < -DODGE_MIN_SPARE && !(status[q] == ST_OK || status[q] == ST_ACCEPTABLE || status[q] == ST_BAD_BOUNDS)) {
            double* xt_ = (double*)malloc(sizeof(double) * (3 * (N + 1) + 2 * N));
            double* ut_ = xt_ + 3 * (N + 1);
            int have = 0;
            double fbest = 0.0;
            /* second level (obca_mpc8 only, oracle/ipm_dense.py: DODGE_LEVEL2_MU): the same two starts at IPOPT's own mu_init, if the first level found nothing */
            for (int pass = 0; pass < (p.variant == 8 ? 4 : 2) && xt_; ++pass) {
                if (pass == 2 && have) break;
                const int side = pass & 1;
                double tst, inf_[4];
                int itr;
                const int st = solve_one(&p, &o, xt_, ut_, &tst, &itr, inf_, side == 0 ? KIND_DODGE_R : KIND_DODGE_L, pass < 2 ? RESTART_MU : MU_INIT, ret);
                it_sum += itr; nf_sum += inf_[3];
                if ((st == ST_OK || st == ST_ACCEPTABLE) && (!have || inf_[0] < fbest)) {
                    memcpy(xo, xt_, sizeof(double) * 3 * (N + 1)); memcpy(uo, ut_, sizeof(double) * 2 * N);
                    ts_opt[q] = tst; status[q] = st;
                    if (io) { io[0] = inf_[0]; io[1] = inf_[1]; io[2] = inf_[2]; }
                    have = 1; fbest = inf_[0];
                }
            }
            free(xt_);
        }
        iters[q] = it_sum;
        if (io) io[3] = nf_sum;
        free(Arep); free(brep);
    }
    return 0;
}

/* model evaluation hooks for tests: objective, rows and Jacobians at a given point (single instance, variant set in prob) */
int obca_oracle_sizes(int N, int n_obs, const int* m, int variant, int* n, int* mh, int* me) {
    int M = 0;
    for (int i = 0; i < n_obs; ++i) M += m[i];
    const int fr = variant == 4;
    *n = (N + 1) * (3 + M + 4 * n_obs) + 2 * N + fr;
    *mh = 2 * (N + 1) * n_obs;
    *me = 3 + 3 * N + (fr ? 3 : 0) + 2 * (N + 1) + 6 * N - 2 * N + (fr ? 2 * (N + 1) : 0) + (variant == 6 ? 2 : 0) + 2 * (N + 1) * n_obs + (N + 1) * (M + 4 * n_obs);
    return 0;
}
