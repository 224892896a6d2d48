// obca_kernel.hip -- batched OBCA-MPC interior-point solver for MI355X (gfx950, CDNA4).
//
// One 64-lane wavefront (= one workgroup) solves one NLP instance; every iterate, row state and KKT block
// of that instance lives in the CU's LDS.  The algorithm is the elastic primal-dual interior-point method
// specified in oracle/ipm_dense.py (IPOPT's filter line-search algorithm applied to the l1-elastic form of
// the NLP that the reference builds in src/obca.py:828-1071 / 1361-1562 / 1564-1758); the Newton step is
// the two-level structured solve of oracle/kkt_structured.py:
//   level 1  per (stage, obstacle): LDL^T of the (lambda_i, mu_i, nu_i) block, one lane per pair, fully in
//            registers, 3x3 Schur complement onto the pose;
//   level 2  backward Riccati sweep over xi_k = (dp_k, du_{k-1}, dT) with the elastic dynamics rows folded in
//            through (I + P E)^-1 P, lane-parallel over the 6x6 / 8x8 stage blocks held in LDS.
// Nothing here calls into oracle/; the oracle is the checker (tests/, bench.py cpu_baseline).
//
// fp64 throughout (reference arithmetic: IPOPT/MUMPS double).  No MFMA: the largest dense block is 8x8.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "obca_device.h"

#define SYNC() __syncthreads()

// Threads per instance.  64 = one wavefront per instance (this file compiled as is).  csrc/obca_kernel_mw.hip includes
// this file with OBCA_NT = 256: four wavefronts (one per SIMD of a CU) share one instance whose working set is too
// large for one wavefront's registers / needs most of the CU's LDS (long horizons).  Every loop below strides by NT,
// every reduction goes through red_* (wave-level for NT = 64, wave + LDS exchange otherwise), and the few places
// where exactly 64 lanes play fixed roles (8x8 stage blocks, the filter) are confined to the first wavefront.
#ifndef OBCA_NT
#define OBCA_NT 64
#endif

#ifdef OBCA_PROFILE
#define PROF_DECL long long prof_last = wall_clock64();
#define PROF(i) { const long long t_ = wall_clock64(); prof_t[i] += t_ - prof_last; prof_last = t_; }
#define RPROF(i) { const long long t_ = wall_clock64(); prof_t[i] += t_ - rlast; rlast = t_; }
__device__ long long* prof_dummy;
#else
#define PROF_DECL
#define PROF(i)
#define RPROF(i)
#endif

namespace {

constexpr int NT = OBCA_NT;
constexpr int MW = OBCA_MAX_EDGES + 6;      // local block width: lambda (<=4) + mu (4) + nu (2)
constexpr int NW = OBCA_MAX_EDGES + 4;      // primal part of the local block

// ---------------------------------------------------------------- out-of-line math
// fp64 log/pow/sincos expand to 150-400 instructions each; inlined at every call site they pushed the hot loop
// past the 64 KiB instruction cache.  One shared copy each.
__device__ __noinline__ double dlog(double x) { return log(x); }
__device__ __noinline__ double dpow(double x, double y) { return exp(y * log(x)); }
struct SinCos { double s, c; };
__device__ __noinline__ SinCos dsincos(double x) { SinCos r; r.s = sin(x); r.c = cos(x); return r; }

// ---------------------------------------------------------------- wave reductions (64 lanes)
// DPP within each row of 16 lanes (no LDS crossbar: __shfl_xor lowers to ds_bpermute, ~12 dependent LDS
// round trips per fp64 reduction), then the four row results are read with v_readlane and combined as
// scalars, so the result is uniform by construction.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    // (every lane of these patterns has a source lane: no `old` value is needed -- passing one cost two v_mov per move)
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_read(double v, int l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)u, l);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
#define OBCA_ROW_REDUCE(OP)                                                          \
    v = OP(v, dpp_move<0xB1>(v));  /* quad_perm [1,0,3,2] */                         \
    v = OP(v, dpp_move<0x4E>(v));  /* quad_perm [2,3,0,1] */                         \
    v = OP(v, dpp_move<0x141>(v)); /* row_half_mirror */                             \
    v = OP(v, dpp_move<0x140>(v)); /* row_mirror: every lane holds its row's result */ \
    const double r0 = lane_read(v, 0), r1 = lane_read(v, 16), r2 = lane_read(v, 32), r3 = lane_read(v, 48);
__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
__device__ __forceinline__ double wave_sum(double v) { OBCA_ROW_REDUCE(op_add) return (r0 + r1) + (r2 + r3); }
__device__ __forceinline__ double wave_max(double v) { OBCA_ROW_REDUCE(fmax) return fmax(fmax(r0, r1), fmax(r2, r3)); }
__device__ __forceinline__ double wave_min(double v) { OBCA_ROW_REDUCE(fmin) return fmin(fmin(r0, r1), fmin(r2, r3)); }
__device__ __forceinline__ int wave_or(int v) { return __any(v) ? 1 : 0; }

// reductions over all NT threads of the instance; the result is identical in every thread
#if OBCA_NT == 64
__device__ __forceinline__ double red_sum(double v) { return wave_sum(v); }
__device__ __forceinline__ double red_max(double v) { return wave_max(v); }
__device__ __forceinline__ double red_min(double v) { return wave_min(v); }
__device__ __forceinline__ int red_or(int v) { return wave_or(v); }
#else
__shared__ double g_red[OBCA_NT / 64];
#define OBCA_BLOCK_REDUCE(WAVE_OP, COMBINE)                                \
    const double w = WAVE_OP(v);                                           \
    if ((threadIdx.x & 63) == 0) g_red[threadIdx.x >> 6] = w;              \
    __syncthreads();                                                       \
    double r = g_red[0];                                                   \
    _Pragma("unroll") for (int i = 1; i < OBCA_NT / 64; ++i) r = COMBINE(r, g_red[i]); /* fixed order */ \
    __syncthreads();                                                       \
    return r;
__device__ __forceinline__ double red_sum(double v) { OBCA_BLOCK_REDUCE(wave_sum, op_add) }
__device__ __forceinline__ double red_max(double v) { OBCA_BLOCK_REDUCE(wave_max, fmax) }
__device__ __forceinline__ double red_min(double v) { OBCA_BLOCK_REDUCE(wave_min, fmin) }
__device__ __forceinline__ int red_or(int v) { return red_max(v ? 1.0 : 0.0) > 0.0 ? 1 : 0; }
#endif

// ---------------------------------------------------------------- instance layout
// Reciprocal for the hot divisions (row linearisation, pivots of the local LDL^T and of the 3 x 3 / 2 x 2 Riccati blocks, step
// lengths): v_rcp_f64 + two Newton steps, <= 1 ulp, five dependent instructions instead of the ~11 of the correctly rounded
// quotient (v_div_scale x 2, rcp, four fma, mul, fma, v_div_fmas, v_div_fixup).  Arguments are finite, normal and non-zero
// wherever the result is used (a zero or infinite argument gives NaN, not inf / 0: such pivots are rejected by their sign
// tests before the reciprocal matters, and trial points are checked with isfinite).  -DOBCA_EXACT_DIV restores 1.0 / d.
#ifdef OBCA_EXACT_DIV
__device__ __forceinline__ double rcp64(double d) { return 1.0 / d; }
#else
__device__ __forceinline__ double rcp64(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}
#endif

struct Lay {
    int N, nO, M, NS, n, free_T, variant;
    int r_init, r_dyn, r_term, r_xb, r_ub, r_acc, r_T, r_tx, r_norm, r_dist, r_lam, r_mu, R;
    int npair;
    int uvs, uvo;             // inputs as the Riccati sweep reads them: u_k = S.uv[uvs * k + uvo .. + 1]
    __device__ __forceinline__ int ip(int k) const { return k * NS; }
    __device__ __forceinline__ int iu(int k) const { return k * NS + 3; }
    __device__ __forceinline__ int il(int k) const { return k * NS + (k < N ? 5 : 3); }
    __device__ __forceinline__ int imu(int k) const { return il(k) + M; }
    __device__ __forceinline__ int iT() const { return n - 1; }
    // compact storage of the objective gradient (it is zero on lambda, mu): stage k -> 5 entries (pose, input), then T
    __device__ __forceinline__ int ig(int k) const { return 5 * k; }
    __device__ __forceinline__ int igT() const { return 5 * (N + 1); }
};
// packed lower triangle of a symmetric 8 x 8 stage block
__device__ __forceinline__ int LS(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

struct Inst {             // per-instance constants (registers, wave-uniform)
    double x0[3], u0[2], Ts, Tmax, term[3];
    double xL[2], xU[2], uL[2], uU[2], gego[4], off, dmin;
    double Q[9], P[9], R1[4], R2[4];
};

static_assert(sizeof(Inst) <= OBCA_INST_DOUBLES * sizeof(double), "Inst does not fit its LDS slot");

// LDS carve-up (all doubles)
struct Sh {
    double *x, *xt, *dx, *gf, *bx;
    double *y, *Einv, *gh, *dy, *tmp; // row data other lanes read (dy: soft rows only); tmp: staging
    double *Lb, *Ub;                         // row bounds (read-only after initialisation)
    double *ct, *st, *cc, *ctt, *stt, *cct;
    double *nu, *dnu, *crot;
    double *Aobs, *bobs, *xref;
    double *Lall, *lall, *Sloc, *Y;
    double *Pk, *qk, *Kk, *kapk, *Mik;
    double *FG, *Mall, *mall;
    double *Zk, *Pi0;                        // forward half of the two-sided sweep (four-wavefront kernels)
    double* lsv;                             // line-search scalars parked in LDS across a corrected (second-order) solve
    // what the serial Riccati sweep reads of the iterate and of the row data: multipliers-side data of the soft rows (init, dyn)
    // and the inputs u_k / the time scale.  Kernels with everything in LDS alias them to Einv / gh / x; the kernel for shapes
    // beyond the LDS (obca_ipm_kernel_gm) keeps LDS mirrors, so that the stage-serial sweep never waits for HBM.
    const double *Es, *gs, *uv, *Tv;
    int* offm;
};

__device__ __forceinline__ double dmaxabs(double a, double b) { return fmax(a, fabs(b)); }

// weight of a row in sums (the N+1 tied Topt copies are one row with multiplicity N+1)
__device__ __forceinline__ double row_w(const Lay& L, int r) {
    return (L.free_T && r >= L.r_T && r < L.r_T + 2) ? (double)(L.N + 1) : 1.0;
}
__device__ __forceinline__ bool row_iseq(const Lay& L, int r) { return r < L.r_xb; }   // init, dyn, term
__device__ __forceinline__ bool row_soft(const Lay& L, int r) { return r < L.r_term; } // init, dyn (Riccati)

// ---------------------------------------------------------------- model evaluation
// trig + c = A^T lambda for an iterate held in xv; results into ct/st/cc
__device__ __forceinline__ void eval_geom(const Lay& L, const Sh& S, const double* xv, double* ct, double* st, double* cc,
                          int lane) {
    for (int k = lane; k <= L.N; k += NT) {
        const SinCos sc = dsincos(xv[L.ip(k) + 2]);
        ct[k] = sc.c;
        st[k] = sc.s;
    }
    for (int pr = lane; pr < L.npair; pr += NT) {
        const int k = pr / L.nO, i = pr - k * L.nO;
        const int o0 = S.offm[i], o1 = S.offm[i + 1];
        const double* lam = xv + L.il(k);
        const double* A = S.Aobs + (size_t)k * L.M * 2;
        double c0 = 0.0, c1 = 0.0;
        for (int j = o0; j < o1; ++j) {
            c0 += A[2 * j] * lam[j];
            c1 += A[2 * j + 1] * lam[j];
        }
        cc[2 * pr] = c0;
        cc[2 * pr + 1] = c1;
    }
    SYNC();
}

// value of elastic row r at iterate xv (geometry arrays must be current)
__device__ __forceinline__ double row_value(const Lay& L, const Sh& S, const Inst& in, const double* xv, const double* ct,
                            const double* st, const double* cc, int r) {
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts, ih = rcp64(h);
    if (r < L.r_dyn) return xv[r] - in.x0[r];
    if (r < L.r_term) {
        const int q = r - L.r_dyn, k = q / 3, j = q - 3 * k;
        const double* pk = xv + L.ip(k);
        const double* pn = xv + L.ip(k + 1);
        const double* u = xv + L.iu(k);
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
        return (prev - xv[L.iu(k) + c]) * ih;
    }
    if (r < L.r_tx) return T;
    if (r < L.r_norm) return xv[L.ip(L.N) + (r - L.r_tx)];
    if (r < L.r_dist) {
        const int pr = r - L.r_norm;
        return cc[2 * pr] * cc[2 * pr] + cc[2 * pr + 1] * cc[2 * pr + 1];
    }
    if (r < L.r_lam) {
        const int pr = r - L.r_dist, k = pr / L.nO, i = pr - k * L.nO;
        const double* pk = xv + L.ip(k);
        const double tx = pk[0] + ct[k] * in.off, ty = pk[1] + st[k] * in.off;
        const double* mu = xv + L.imu(k) + 4 * i;
        const double* lam = xv + L.il(k);
        const double* b = S.bobs + (size_t)k * L.M;
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

__device__ __forceinline__ void row_bounds(const Lay& L, const Inst& in, int r, double& lo, double& up) {
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
__device__ __forceinline__ void rot_value(const Lay& L, const double* xv, const double* ct, const double* st,
                                          const double* cc, int pr, double& e1, double& e2) {
    const int k = pr / L.nO, i = pr - k * L.nO;
    const double* mu = xv + L.imu(k) + 4 * i;
    const double c0 = cc[2 * pr], c1 = cc[2 * pr + 1];
    e1 = mu[0] - mu[2] + ct[k] * c0 + st[k] * c1;
    e2 = mu[1] - mu[3] - st[k] * c0 + ct[k] * c1;
}

// scaled objective sf*f (all lanes return the same value); optionally its gradient into S.gf
template <bool GRAD>
__device__ __forceinline__ double eval_objective(const Lay& L, const Sh& S, const Inst& in, const double* xv, double sf, int lane) {
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts, ih2 = rcp64(h * h), ih2T = ih2 * rcp64(T);
    double part = 0.0, gT = 0.0;
    for (int k = lane; k <= L.N; k += NT) {
        const double* pk = xv + L.ip(k);
        const double* W = (k < L.N) ? in.Q : in.P;
        double e[3], We[3];
        for (int j = 0; j < 3; ++j) e[j] = pk[j] - S.xref[j * (L.N + 1) + k];
        for (int a = 0; a < 3; ++a) We[a] = W[3 * a] * e[0] + W[3 * a + 1] * e[1] + W[3 * a + 2] * e[2];
        part += e[0] * We[0] + e[1] * We[1] + e[2] * We[2];
        if (GRAD) for (int a = 0; a < 3; ++a) S.gf[L.ig(k) + a] = sf * 2.0 * We[a];
        if (k < L.N) {
            const double* u = xv + L.iu(k);
            const double r0 = in.R1[0] * u[0] + in.R1[1] * u[1], r1 = in.R1[2] * u[0] + in.R1[3] * u[1];
            part += u[0] * r0 + u[1] * r1;
            double g0 = 2.0 * r0, g1 = 2.0 * r1;
            if (k + 1 < L.N) {          // (u_{k+1}-u_k)' R2 (.) / h^2
                const double* un = xv + L.iu(k + 1);
                const double q0 = un[0] - u[0], q1 = un[1] - u[1];
                const double s0 = in.R2[0] * q0 + in.R2[1] * q1, s1 = in.R2[2] * q0 + in.R2[3] * q1;
                const double qq = q0 * s0 + q1 * s1;
                part += qq * ih2;
                g0 -= 2.0 * s0 * ih2;
                g1 -= 2.0 * s1 * ih2;
                gT += -2.0 * qq * ih2T;
            }
            if (k >= 1) {
                const double* um = xv + L.iu(k - 1);
                const double q0 = u[0] - um[0], q1 = u[1] - um[1];
                g0 += 2.0 * (in.R2[0] * q0 + in.R2[1] * q1) * ih2;
                g1 += 2.0 * (in.R2[2] * q0 + in.R2[3] * q1) * ih2;
            }
            if (GRAD) { S.gf[L.ig(k) + 3] = sf * g0; S.gf[L.ig(k) + 4] = sf * g1; }
        }
    }
    double f = red_sum(part);
    if (L.free_T) {
        f += (L.N + 1) * (10.0 * T + T * T);
        if (GRAD) {
            gT = red_sum(gT) + (L.N + 1) * (10.0 + 2.0 * T);
            if (lane == 0) S.gf[L.igT()] = sf * gT;
        }
    }
    if (GRAD) SYNC();             // the lambda / mu entries of gf are zero for good (set once at the start)
    return sf * f;
}

// out = gf + J^T ymul (+ rotation rows with nu): gradient of the Lagrangian w.r.t. x.
// One target entry per lane (gather form, deterministic).
// HAT: multipliers yhat = y + ghat / E of the condensed rows (the soft rows keep y) evaluated on the fly -- storing
// them cost one more row array in LDS
template <bool HAT>
__device__ __forceinline__ void gather_grad(const Lay& L, const Sh& S, const Inst& in, double* out, int lane) {
    auto YM = [&](int r) -> double { return HAT ? S.y[r] + S.gh[r] * S.Einv[r] : S.y[r]; };     // condensed rows
    auto YS = [&](int r) -> double { return S.y[r]; };                                             // init, dyn (soft)
    const double* xv = S.x;
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts, ih = rcp64(h), iTh = ih * rcp64(T);
    // poses and inputs
    for (int t = lane; t < (L.N + 1) * 5; t += NT) {
        const int k = t / 5, j = t - 5 * k;
        if (j >= 3 && k == L.N) continue;
        const double cs = S.ct[k], sn = S.st[k];
        double v;
        if (j < 3) {
            v = S.gf[L.ig(k) + j];
            if (k == 0) v += YS(L.r_init + j);
            if (k >= 1) v += YS(L.r_dyn + 3 * (k - 1) + j);
            if (k < L.N) {
                const double* yd = S.y + L.r_dyn + 3 * k;
                v -= yd[j];
                if (j == 2) {
                    const double vel = xv[L.iu(k)];
                    v -= h * vel * (-sn * yd[0] + cs * yd[1]);
                }
            }
            if (k == L.N && L.variant == 4) v += YM(L.r_term + j);
            if (j < 2) {
                v += YM(L.r_xb + 2 * k + j);
                if (k == L.N && L.variant == 6) v += YM(L.r_tx + j);
            }
            for (int i = 0; i < L.nO; ++i) {
                const int pr = k * L.nO + i;
                const double c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
                const double yd = YM(L.r_dist + pr);
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
            v = S.gf[L.ig(k) + 3 + c];
            const double* yd = S.y + L.r_dyn + 3 * k;
            v -= (c == 0) ? h * (cs * yd[0] + sn * yd[1]) : h * yd[2];
            v += YM(L.r_ub + 2 * k + c);
            v -= YM(L.r_acc + 2 * k + c) * ih;
            if (k + 1 < L.N) v += YM(L.r_acc + 2 * (k + 1) + c) * ih;
            out[L.iu(k) + c] = v;
        }
    }
    // lambda
    for (int t = lane; t < (L.N + 1) * L.M; t += NT) {
        const int k = t / L.M, j = t - k * L.M;
        int i = 0;
        while (j >= S.offm[i + 1]) ++i;
        const int pr = k * L.nO + i;
        const double a0 = S.Aobs[((size_t)k * L.M + j) * 2], a1 = S.Aobs[((size_t)k * L.M + j) * 2 + 1];
        const double cs = S.ct[k], sn = S.st[k], c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
        const double* pk = xv + L.ip(k);
        const double tx = pk[0] + cs * in.off, ty = pk[1] + sn * in.off;
        double v = S.nu[2 * pr] * (cs * a0 + sn * a1) + S.nu[2 * pr + 1] * (-sn * a0 + cs * a1);
        v += YM(L.r_norm + pr) * 2.0 * (a0 * c0 + a1 * c1);
        v += YM(L.r_dist + pr) * (tx * a0 + ty * a1 - S.bobs[(size_t)k * L.M + j]);
        v += YM(L.r_lam + t);
        out[L.il(k) + j] = v;
    }
    // mu
    for (int t = lane; t < (L.N + 1) * 4 * L.nO; t += NT) {
        const int w4 = 4 * L.nO, k = t / w4, q = t - k * w4, i = q >> 2, j = q & 3;
        const int pr = k * L.nO + i;
        const double sgn = (j < 2) ? 1.0 : -1.0;
        double v = sgn * S.nu[2 * pr + (j & 1)];
        v -= in.gego[j] * YM(L.r_dist + pr);
        v += YM(L.r_mu + t);
        out[L.imu(k) + q] = v;
    }
    // time scale
    if (L.free_T) {
        double part = 0.0;
        for (int k = lane; k < L.N; k += NT) {
            const double* u = xv + L.iu(k);
            const double* yd = S.y + L.r_dyn + 3 * k;
            part -= in.Ts * (u[0] * S.ct[k] * yd[0] + u[0] * S.st[k] * yd[1] + u[1] * yd[2]);
            for (int c = 0; c < 2; ++c) {
                const double prev = (k == 0) ? in.u0[c] : xv[L.iu(k - 1) + c];
                part -= YM(L.r_acc + 2 * k + c) * (prev - u[c]) * iTh;
            }
        }
        part = red_sum(part);
        if (lane == 0) out[L.iT()] = S.gf[L.igT()] + part + (L.N + 1) * (YM(L.r_T) + YM(L.r_T + 1));
    }
    SYNC();
}

// J_r dx for the condensed rows (soft rows get their dy from the Riccati sweep)
__device__ __forceinline__ double row_jdx(const Lay& L, const Sh& S, const Inst& in, int r) {
    const double* xv = S.x;
    const double* d = S.dx;
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double dT = L.free_T ? d[L.iT()] : 0.0;
    const double h = T * in.Ts, ih = rcp64(h), iTh = ih * rcp64(T);
    if (r < L.r_xb) return d[L.ip(L.N) + (r - L.r_term)];
    if (r < L.r_ub) { const int q = r - L.r_xb; return d[L.ip(q >> 1) + (q & 1)]; }
    if (r < L.r_acc) { const int q = r - L.r_ub; return d[L.iu(q >> 1) + (q & 1)]; }
    if (r < L.r_T) {
        const int q = r - L.r_acc, k = q >> 1, c = q & 1;
        const double prev = (k == 0) ? in.u0[c] : xv[L.iu(k - 1) + c];
        const double dprev = (k == 0) ? 0.0 : d[L.iu(k - 1) + c];
        const double qq = prev - xv[L.iu(k) + c];
        return (dprev - d[L.iu(k) + c]) * ih - qq * iTh * dT;
    }
    if (r < L.r_tx) return dT;
    if (r < L.r_norm) return d[L.ip(L.N) + (r - L.r_tx)];
    if (r < L.r_lam) {
        const bool isn = r < L.r_dist;
        const int pr = isn ? r - L.r_norm : r - L.r_dist;
        const int k = pr / L.nO, i = pr - k * L.nO;
        const double c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
        const double* A = S.Aobs + (size_t)k * L.M * 2;
        const double* dl = d + L.il(k);
        double s0 = 0.0, s1 = 0.0, sb = 0.0;
        for (int j = S.offm[i]; j < S.offm[i + 1]; ++j) {
            s0 += A[2 * j] * dl[j];
            s1 += A[2 * j + 1] * dl[j];
            sb += S.bobs[(size_t)k * L.M + j] * dl[j];
        }
        if (isn) return 2.0 * (c0 * s0 + c1 * s1);
        const double cs = S.ct[k], sn = S.st[k];
        const double* pk = xv + L.ip(k);
        const double* dp = d + L.ip(k);
        const double* dm = d + L.imu(k) + 4 * i;
        const double tx = pk[0] + cs * in.off, ty = pk[1] + sn * in.off;
        return -(in.gego[0] * dm[0] + in.gego[1] * dm[1] + in.gego[2] * dm[2] + in.gego[3] * dm[3]) + tx * s0 +
               ty * s1 - sb + c0 * dp[0] + c1 * dp[1] + in.off * (-sn * c0 + cs * c1) * dp[2];
    }
    if (r < L.r_mu) { const int q = r - L.r_lam, k = q / L.M; return d[L.il(k) + (q - k * L.M)]; }
    { const int q = r - L.r_mu, w4 = 4 * L.nO, k = q / w4; return d[L.imu(k) + (q - k * w4)]; }
}

// ---------------------------------------------------------------- lane-owned rows (registers)
// Row r = lane + 64*j lives in slot j of its owner lane: bounds, slack, elastic pair, multipliers, the row value
// and the per-iteration linearisation (inverse D's and residuals) never leave registers.  Only what OTHER lanes
// need goes to LDS: y (gather / curvature), yhat, Einv, ghat (assembly, Riccati).
#define OBCA_ROW_FIELDS(X) X(s) X(p) X(n) X(zL) X(zU) X(zp) X(zn) X(g) X(dy) X(iDs) X(iDp) X(iDn) X(rs) X(rp) X(rn)
template <int RPL>
struct Rows {                    // row state in registers: slot j of this thread is row r = thread + NT * j
#define X(f) double f##_[RPL];
    OBCA_ROW_FIELDS(X)
#undef X
    __device__ __forceinline__ static constexpr int slots() { return RPL; }
#define X(f) __device__ __forceinline__ double& f(int j) { return f##_[j]; } \
             __device__ __forceinline__ double f(int j) const { return f##_[j]; }
    OBCA_ROW_FIELDS(X)
#undef X
};
// RPL = 0: row state in MEMORY (the HBM workspace of the kernel for shapes beyond the LDS, obca_ipm_kernel_gm): same
// thread <-> row assignment, any number of slots
template <>
struct Rows<0> {
#define X(f) double* f##_;
    OBCA_ROW_FIELDS(X)
#undef X
    int lane_, nslots_;
    __device__ __forceinline__ int slots() const { return nslots_; }
#define X(f) __device__ __forceinline__ double& f(int j) const { return f##_[lane_ + NT * j]; }
    OBCA_ROW_FIELDS(X)
#undef X
};

// RPL = -5 (four-wavefront kernel for up to 1280 rows): five slots, the first four in registers, the FIFTH IN LDS.  With 256
// threads the fifth slot holds rows 1024 .. R-1 only -- 90 of C3's 1114 rows -- but as registers it cost every thread 30 VGPRs
// and the kernel spilled (144 B of scratch per lane, twelve doubles of row state written and re-read every iteration).  Element
// i < nl = R_max - 1024 of the block is 15 consecutive doubles (odd stride: consecutive threads fall on different LDS banks);
// the threads without a fifth row share one dummy element (only the unconditional initialisations ever touch it).
template <>
struct Rows<-5> {
#define X(f) double f##_[4];
    OBCA_ROW_FIELDS(X)
#undef X
    double* m_;
    double* gl_;        // the row VALUES of the four register slots live in LDS too: gl_[NT j] is row thread + NT j
    enum {
#define X(f) K_##f,
        OBCA_ROW_FIELDS(X)
#undef X
        K_N };
    __device__ __forceinline__ static constexpr int slots() { return 5; }
#define X(f) __device__ __forceinline__ double& f##R(int j) { return j < 4 ? f##_[j] : m_[K_##f]; } \
             __device__ __forceinline__ double f##R(int j) const { return j < 4 ? f##_[j] : m_[K_##f]; }
    OBCA_ROW_FIELDS(X)
#undef X
#define X(f) __device__ __forceinline__ double& f(int j) { return f##R(j); } \
             __device__ __forceinline__ double f(int j) const { return f##R(j); }
    X(s) X(p) X(n) X(zL) X(zU) X(zp) X(zn) X(dy) X(iDs) X(iDp) X(iDn) X(rs) X(rp) X(rn)
#undef X
    __device__ __forceinline__ double& g(int j) { return j < 4 ? gl_[NT * j] : m_[K_g]; }
    __device__ __forceinline__ double g(int j) const { return j < 4 ? gl_[NT * j] : m_[K_g]; }
};

struct Err { double E, dual, prim, comp; };

template <int RPL>
__device__ __forceinline__ Err ipm_errors(const Lay& L, const Sh& S, const Rows<RPL>& W, double mu, double rho, double rxmax,
                          double crotmax, double nusum, int lane) {
    double dual = 0.0, prim = 0.0, comp = 0.0, ysum = 0.0, zsum = 0.0, nz = 0.0, nrow = 0.0;
#pragma unroll
    for (int j = 0; j < W.slots(); ++j) {
        const int r = lane + NT * j;
        if (r < L.R) {
            const double w = row_w(L, r);
            const bool eq = row_iseq(L, r);
            const double lo_ = S.Lb[r], up_ = S.Ub[r];
            const bool hasL = !eq && lo_ > -INFINITY, hasU = !eq && up_ < INFINITY;
            const double s = W.s(j), y = S.y[r], p = W.p(j), n = W.n(j);
            const double zL = hasL ? W.zL(j) : 0.0, zU = hasU ? W.zU(j) : 0.0, zp = W.zp(j), zn = W.zn(j);
            if (!eq) dual = dmaxabs(dual, -y - zL + zU);
            dual = dmaxabs(dual, rho - y - zp);
            dual = dmaxabs(dual, rho + y - zn);
            prim = dmaxabs(prim, W.g(j) - (eq ? 0.0 : s) - p + n);
            comp = dmaxabs(comp, p * zp - mu);
            comp = dmaxabs(comp, n * zn - mu);
            if (hasL) comp = dmaxabs(comp, (s - lo_) * zL - mu);
            if (hasU) comp = dmaxabs(comp, (up_ - s) * zU - mu);
            ysum += w * fabs(y);
            zsum += w * (zL + zU + zp + zn);
            nz += w * ((hasL ? 1.0 : 0.0) + (hasU ? 1.0 : 0.0) + 2.0);
            nrow += w;
        }
    }
    dual = fmax(red_max(dual), rxmax);
    prim = fmax(red_max(prim), crotmax);
    comp = red_max(comp);
    ysum = red_sum(ysum) + nusum;
    zsum = red_sum(zsum);
    nz = red_sum(nz);
    nrow = red_sum(nrow) + 2.0 * L.npair;
    const double sd = fmax(OBCA_S_MAX, (ysum + zsum) / (nrow + nz)) / OBCA_S_MAX;
    const double sc = fmax(OBCA_S_MAX, zsum / nz) / OBCA_S_MAX;
    Err e;
    e.dual = dual; e.prim = prim; e.comp = comp;
    e.E = fmax(fmax(dual / sd, prim), comp / sc);
    return e;
}

// First evaluation of an iteration: the optimality error at mu = 0 (termination tests) and at the current mu (barrier
// update) differ only in the complementarity term, and the constraint violation theta, the elastic sum and the largest
// elastic variable read the same registers -- one pass over the rows instead of three.  nz / nrow (numbers of bound
// multipliers and of rows) never change and are passed in.  th_lane carries the lane's rotation-row part of theta in.
struct ErrFirst { Err e0, em; double th, pnsum, emax; };
template <int RPL>
__device__ __forceinline__ ErrFirst ipm_errors_first(const Lay& L, const Sh& S, const Rows<RPL>& W, double mu, double rho, double rxmax,
                                     double crotmax, double nusum, double th_lane, double nz, double nrow, int lane) {
    double dual = 0.0, prim = 0.0, comp0 = 0.0, compm = 0.0, ysum = 0.0, zsum = 0.0, th = th_lane, pnsum = 0.0, emax = 0.0;
#pragma unroll
    for (int j = 0; j < W.slots(); ++j) {
        const int r = lane + NT * j;
        if (r < L.R) {
            const double w = row_w(L, r);
            const bool eq = row_iseq(L, r);
            const double lo_ = S.Lb[r], up_ = S.Ub[r];
            const bool hasL = !eq && lo_ > -INFINITY, hasU = !eq && up_ < INFINITY;
            const double s = W.s(j), y = S.y[r], p = W.p(j), n = W.n(j);
            const double zL = hasL ? W.zL(j) : 0.0, zU = hasU ? W.zU(j) : 0.0, zp = W.zp(j), zn = W.zn(j);
            if (!eq) dual = dmaxabs(dual, -y - zL + zU);
            dual = dmaxabs(dual, rho - y - zp);
            dual = dmaxabs(dual, rho + y - zn);
            const double res = W.g(j) - (eq ? 0.0 : s) - p + n;
            prim = dmaxabs(prim, res);
            th += w * fabs(res);
            pnsum += w * (p + n);
            emax = fmax(emax, p + n);
            comp0 = dmaxabs(comp0, p * zp); compm = dmaxabs(compm, p * zp - mu);
            comp0 = dmaxabs(comp0, n * zn); compm = dmaxabs(compm, n * zn - mu);
            if (hasL) { comp0 = dmaxabs(comp0, (s - lo_) * zL); compm = dmaxabs(compm, (s - lo_) * zL - mu); }
            if (hasU) { comp0 = dmaxabs(comp0, (up_ - s) * zU); compm = dmaxabs(compm, (up_ - s) * zU - mu); }
            ysum += w * fabs(y);
            zsum += w * (zL + zU + zp + zn);
        }
    }
    ErrFirst o;
    dual = fmax(red_max(dual), rxmax);
    prim = fmax(red_max(prim), crotmax);
    comp0 = red_max(comp0);
    compm = red_max(compm);
    ysum = red_sum(ysum) + nusum;
    zsum = red_sum(zsum);
    o.th = red_sum(th); o.pnsum = red_sum(pnsum); o.emax = red_max(emax);
    const double sd = fmax(OBCA_S_MAX, (ysum + zsum) / (nrow + nz)) / OBCA_S_MAX;
    const double sc = fmax(OBCA_S_MAX, zsum / nz) / OBCA_S_MAX;
    o.e0.dual = dual; o.e0.prim = prim; o.e0.comp = comp0; o.e0.E = fmax(fmax(dual / sd, prim), comp0 / sc);
    o.em.dual = dual; o.em.prim = prim; o.em.comp = compm; o.em.E = fmax(fmax(dual / sd, prim), compm / sc);
    return o;
}

// linearisation of one row at the current iterate: inverse D's and residuals (IPOPT's Sigma + delta_w)
struct Lin { double iDs, iDp, iDn, rs, rp, rn; };
__device__ __forceinline__ Lin row_lin(double lo, double up, bool eq, double s, double p, double n, double y,
                                       double zL, double zU, double zp, double zn, double mu, double rho, double dw) {
    const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
    double sig = 0.0, gs = 0.0;
    if (hasL) { const double isl = rcp64(s - lo); sig += zL * isl; gs -= mu * isl; }
    if (hasU) { const double isu = rcp64(up - s); sig += zU * isu; gs += mu * isu; }
    if (hasL && !hasU) gs += OBCA_KAPPA_D * mu;
    if (hasU && !hasL) gs -= OBCA_KAPPA_D * mu;
    const double ip = rcp64(p), inn = rcp64(n);
    Lin q;
    q.iDs = eq ? 0.0 : rcp64(sig + dw);
    q.iDp = rcp64(zp * ip + dw);
    q.iDn = rcp64(zn * inn + dw);
    q.rs = eq ? 0.0 : (-y + gs);
    q.rp = rho - y - mu * ip;
    q.rn = rho + y - mu * inn;
    return q;
}

// barrier terms of one row at (s, p, n): -mu*log(product of its distances) + linear damping; ONE log per row
__device__ __forceinline__ double row_barrier(double lo, double up, bool eq, double s, double p, double n, double mu,
                                              double rho) {
    const bool hasL = !eq && lo > -INFINITY, hasU = !eq && up < INFINITY;
    double prod = p * n, lin = rho * (p + n);
    if (hasL) { prod *= (s - lo); if (!hasU) lin += OBCA_KAPPA_D * mu * (s - lo); }
    if (hasU) { prod *= (up - s); if (!hasL) lin += OBCA_KAPPA_D * mu * (up - s); }
    return lin - mu * dlog(prod);
}

// ---------------------------------------------------------------- stage-cost assembly (one lane per stage)
// Lall[k] is the 8x8 symmetric stage matrix over (dp(0:3), du_prev(3:5), dT(5), du(6:8)); lall[k] its gradient.
__device__ __forceinline__ void assemble_stages(const Lay& L, const Sh& S, const Inst& in, double sf, double dw, int lane) {
    const double* xv = S.x;
    const double T = L.free_T ? xv[L.iT()] : 1.0;
    const double h = T * in.Ts, ih = rcp64(h), ih2 = ih * ih, iT = rcp64(T), iTh = iT * ih, iThh = iTh * ih;
    double HTT = 0.0;
    for (int k = lane; k <= L.N; k += NT) {
        // the stage block is accumulated in registers (read-modify-write through LDS serialised ~100 round trips)
        double Hpp[3][3], Huu[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, Cpu[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        double hpu = 0.0, hpT = 0.0, huT[2] = {0.0, 0.0};
        const double cs = S.ct[k], sn = S.st[k];
        const double* W = (k < L.N) ? in.Q : in.P;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) Hpp[a][b] = sf * 2.0 * W[3 * a + b];
            Hpp[a][a] += dw;
        }
        Hpp[0][0] += S.Einv[L.r_xb + 2 * k];
        Hpp[1][1] += S.Einv[L.r_xb + 2 * k + 1];
        if (k == L.N && L.variant == 4) { Hpp[0][0] += S.Einv[L.r_term]; Hpp[1][1] += S.Einv[L.r_term + 1]; Hpp[2][2] += S.Einv[L.r_term + 2]; }
        if (k == L.N && L.variant == 6) { Hpp[0][0] += S.Einv[L.r_tx]; Hpp[1][1] += S.Einv[L.r_tx + 1]; }
        double hth = 0.0;                                           // theta-theta Lagrangian curvature
        for (int i = 0; i < L.nO; ++i) {
            const int pr = k * L.nO + i;
            const double c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
            const double yd = S.y[L.r_dist + pr], Ei = S.Einv[L.r_dist + pr];
            const double gp[3] = {c0, c1, in.off * (-sn * c0 + cs * c1)};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) Hpp[a][b] += Ei * gp[a] * gp[b];
            hth += yd * in.off * (-cs * c0 - sn * c1);
            hth += S.nu[2 * pr] * (-cs * c0 - sn * c1) + S.nu[2 * pr + 1] * (sn * c0 - cs * c1);
        }
        double lu0 = 0.0, lu1 = 0.0;
        if (k < L.N) {
            const double* u = xv + L.iu(k);
            const double* yd = S.y + L.r_dyn + 3 * k;
            const double u0 = u[0], u1 = u[1], y0 = yd[0], y1 = yd[1], y2 = yd[2];
            hth += h * u0 * (y0 * cs + y1 * sn);
            hpu = h * (y0 * sn - y1 * cs);                            // theta - v
            const int ncost = (k + 1 < L.N ? 1 : 0) + (k >= 1 ? 1 : 0);  // acceleration cost pairs touching u_k
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    Huu[a][b] = sf * 2.0 * in.R1[2 * a + b] + sf * 2.0 * in.R2[2 * a + b] * ih2 * ncost;
                    if (k >= 1) Cpu[a][b] = -sf * 2.0 * in.R2[2 * a + b] * ih2;
                }
                Huu[a][a] += dw + S.Einv[L.r_ub + 2 * k + a];
            }
            lu0 = S.bx[L.iu(k)]; lu1 = S.bx[L.iu(k) + 1];
            // acceleration rows k (u_{k-1}, u_k) and k+1 (u_k, u_{k+1})
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double Ek = S.Einv[L.r_acc + 2 * k + c];
                Huu[c][c] += Ek * ih2;
                if (k >= 1) Cpu[c][c] -= Ek * ih2;
                if (k + 1 < L.N) Huu[c][c] += S.Einv[L.r_acc + 2 * (k + 1) + c] * ih2;
            }
            if (L.free_T) {
                // (theta,T), (u,T) and (T,T) entries
                hpT = -in.Ts * u0 * (-y0 * sn + y1 * cs);
                huT[0] = -in.Ts * (y0 * cs + y1 * sn);
                huT[1] = -in.Ts * y2;
                const double uc[2] = {u0, u1};
                double qa[2] = {0, 0}, qb[2] = {0, 0};
                if (k + 1 < L.N) { qa[0] = xv[L.iu(k + 1)] - u0; qa[1] = xv[L.iu(k + 1) + 1] - u1; }
                if (k >= 1) { qb[0] = u0 - xv[L.iu(k - 1)]; qb[1] = u1 - xv[L.iu(k - 1) + 1]; }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double q = (k == 0) ? in.u0[c] - uc[c] : -qb[c];     // u_{k-1} - u_k
                    const double ya = S.y[L.r_acc + 2 * k + c], Ek = S.Einv[L.r_acc + 2 * k + c];
                    huT[c] += ya * iTh + Ek * q * iThh;
                    HTT += ya * 2.0 * q * (iT * iTh) + Ek * q * q * (iTh * iTh);
                    if (k + 1 < L.N) {
                        const double qn = -qa[c];                              // u_k - u_{k+1}
                        const double yn = S.y[L.r_acc + 2 * (k + 1) + c], En = S.Einv[L.r_acc + 2 * (k + 1) + c];
                        huT[c] += -yn * iTh - En * qn * iThh;
                    }
                }
                // acceleration cost cross terms
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const double Ra = in.R2[2 * a] * qa[0] + in.R2[2 * a + 1] * qa[1];
                    const double Rb = in.R2[2 * a] * qb[0] + in.R2[2 * a + 1] * qb[1];
                    huT[a] += sf * 4.0 * ih2 * iT * (Ra - Rb);
                }
                if (k + 1 < L.N) {
                    const double qq = qa[0] * (in.R2[0] * qa[0] + in.R2[1] * qa[1]) + qa[1] * (in.R2[2] * qa[0] + in.R2[3] * qa[1]);
                    HTT += sf * 6.0 * qq * ih2 * (iT * iT);
                }
            }
        } else {
            Huu[0][0] = 1.0; Huu[1][1] = 1.0;             // no input at the last stage
        }
        Hpp[2][2] += hth;
        // write the 8x8 block over (dp(0:3), du_prev(3:5), dT(5), du(6:8)) and its gradient
        double* H = S.Lall + 36 * k;                    // packed lower triangle
        double* lv = S.lall + 8 * k;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b > a) continue;
                double v = 0.0;
                if (a < 3 && b < 3) v = Hpp[a][b];
                else if (a >= 6 && b >= 6) v = Huu[a - 6][b - 6];
                else if (a >= 3 && a < 5 && b >= 6) v = Cpu[a - 3][b - 6];
                else if (b >= 3 && b < 5 && a >= 6) v = Cpu[b - 3][a - 6];
                else if ((a == 2 && b == 6) || (a == 6 && b == 2)) v = hpu;
                else if ((a == 2 && b == 5) || (a == 5 && b == 2)) v = hpT;
                else if (a == 5 && b >= 6) v = huT[b - 6];
                else if (b == 5 && a >= 6) v = huT[a - 6];
                H[a * (a + 1) / 2 + b] = v;
            }
        lv[0] = S.bx[L.ip(k)]; lv[1] = S.bx[L.ip(k) + 1]; lv[2] = S.bx[L.ip(k) + 2];
        lv[3] = 0.0; lv[4] = 0.0; lv[5] = 0.0; lv[6] = lu0; lv[7] = lu1;
    }
    if (L.free_T) {
        HTT = red_sum(HTT);
        if (lane == 0) {
            const double w = (double)(L.N + 1);
            HTT += sf * 2.0 * w + dw * w + w * (S.Einv[L.r_T] + S.Einv[L.r_T + 1]);
            S.Lall[LS(5, 5)] += HTT;
            S.lall[5] = S.bx[L.iT()];
        }
    } else if (lane == 0) {
        S.Lall[LS(5, 5)] = 1.0;                        // dT pinned to zero in the fixed-time variants
    }
    SYNC();
}

// ---------------------------------------------------------------- level 1: local blocks, one lane per pair
// Writes Y[pr] = Kloc^-1 [G | rloc] (MW x 4) and Sloc[pr] = (G^T Y_G (3x3), G^T Y_r (3)). Returns 1 on a
// wrong-sign pivot.
__device__ __forceinline__ int local_blocks(const Lay& L, const Sh& S, const Inst& in, double dw, int lane) {
    int bad = 0;
#ifdef NO_LOCAL
    return 0;
#endif
    // TWO lanes per pair: both factor the block (registers), each solves two of the four right-hand sides
    // [G_x G_y | G_theta rloc] and produces the matching columns of Y and of the Schur complement G'Y.
    for (int w = lane; w < 2 * L.npair; w += NT) {
        const int pr = w >> 1;
        const bool hi = (w & 1) != 0;
        const int k = pr / L.nO, i = pr - k * L.nO;
        const int o0 = S.offm[i], m = S.offm[i + 1] - o0;
        const double cs = S.ct[k], sn = S.st[k];
        const double c0 = S.cc[2 * pr], c1 = S.cc[2 * pr + 1];
        const double* pk = S.x + L.ip(k);
        const double tx = pk[0] + cs * in.off, ty = pk[1] + sn * in.off;
        const double yn = S.y[L.r_norm + pr], En = S.Einv[L.r_norm + pr];
        const double yd = S.y[L.r_dist + pr], Ed = S.Einv[L.r_dist + pr];
        const double nu1 = S.nu[2 * pr], nu2 = S.nu[2 * pr + 1];
        double a0[OBCA_MAX_EDGES], a1[OBCA_MAX_EDGES], gn[OBCA_MAX_EDGES], gd[NW];
        double K[MW * (MW + 1) / 2], Yv[MW][2], G0[MW], G1[MW], G2[MW];
#define KP(a, b) K[((a) * ((a) + 1)) / 2 + (b)]
        const double dth = -sn * c0 + cs * c1;          // d(dist)/d(theta) / off  and  d(e1)/d(theta)
#pragma unroll
        for (int j = 0; j < OBCA_MAX_EDGES; ++j) {
            const bool on = j < m;
            a0[j] = on ? S.Aobs[((size_t)k * L.M + o0 + j) * 2] : 0.0;
            a1[j] = on ? S.Aobs[((size_t)k * L.M + o0 + j) * 2 + 1] : 0.0;
            const double bj = on ? S.bobs[(size_t)k * L.M + o0 + j] : 0.0;
            gn[j] = 2.0 * (a0[j] * c0 + a1[j] * c1);
            gd[j] = on ? (tx * a0[j] + ty * a1[j] - bj) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) gd[OBCA_MAX_EDGES + j] = -in.gego[j];
        // primal block (lower triangle)
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int b = 0; b < NW; ++b)
                if (b <= a) {
                    double v = Ed * gd[a] * gd[b];
                    if (a < OBCA_MAX_EDGES) v += En * gn[a] * gn[b] + yn * 2.0 * (a0[a] * a0[b] + a1[a] * a1[b]);
                    KP(a, b) = v;
                }
        double rl[MW];
#pragma unroll
        for (int j = 0; j < OBCA_MAX_EDGES; ++j) {
            const bool on = j < m;
            KP(j, j) += on ? (dw + S.Einv[L.r_lam + k * L.M + o0 + j]) : 1.0;   // padded slots: identity
            rl[j] = on ? -S.bx[L.il(k) + o0 + j] : 0.0;
            // coupling to the pose: columns (x, y, theta)
            G0[j] = Ed * gd[j] * c0 + yd * a0[j];
            G1[j] = Ed * gd[j] * c1 + yd * a1[j];
            G2[j] = Ed * gd[j] * in.off * dth + yd * in.off * (-sn * a0[j] + cs * a1[j]) +
                       nu1 * (-sn * a0[j] + cs * a1[j]) + nu2 * (-cs * a0[j] - sn * a1[j]);
            // rotation rows
            KP(NW, j) = cs * a0[j] + sn * a1[j];
            KP(NW + 1, j) = -sn * a0[j] + cs * a1[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = OBCA_MAX_EDGES + j;
            KP(a, a) += dw + S.Einv[L.r_mu + k * 4 * L.nO + 4 * i + j];
            rl[a] = -S.bx[L.imu(k) + 4 * i + j];
            G0[a] = Ed * gd[a] * c0;
            G1[a] = Ed * gd[a] * c1;
            G2[a] = Ed * gd[a] * in.off * dth;
            KP(NW, a) = (j == 0) ? 1.0 : (j == 2) ? -1.0 : 0.0;
            KP(NW + 1, a) = (j == 1) ? 1.0 : (j == 3) ? -1.0 : 0.0;
        }
        KP(NW, NW) = 0.0; KP(NW + 1, NW) = 0.0; KP(NW + 1, NW + 1) = 0.0;
        G0[NW] = 0.0; G1[NW] = 0.0; G2[NW] = dth; rl[NW] = -S.crot[2 * pr];
        G0[NW + 1] = 0.0; G1[NW + 1] = 0.0; G2[NW + 1] = -cs * c0 - sn * c1; rl[NW + 1] = -S.crot[2 * pr + 1];
#pragma unroll
        for (int a = 0; a < MW; ++a) { Yv[a][0] = hi ? G2[a] : G0[a]; Yv[a][1] = hi ? rl[a] : G1[a]; }
        // LDL^T without pivoting (quasi-definite when the primal block is positive definite), forward substitution
        // fused.  Rectangular constant-trip loops with predicates: after full unrolling every index is a literal,
        // so K and Yv live in registers (triangular bounds defeat the unroller and force them to scratch).
        // Inertia: the block has NW positive and 2 negative eigenvalues iff exactly 2 of its pivots are negative
        // (Sylvester) -- counted, not tested by position: an indefinite (lambda, mu) block that is positive definite on
        // the null space of the rotation rows gives one negative primal and one positive dual pivot and is fine.
        double dinv[MW];
#pragma unroll
        for (int j = 0; j < MW; ++j) {
            const double d = KP(j, j);
            dinv[j] = rcp64(d);
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
                    Yv[a][0] -= KP(a, j) * Yv[j][0];
                    Yv[a][1] -= KP(a, j) * Yv[j][1];
                }
            }
        }
        {
            int nneg = 0;                       // read off the reciprocals that the back substitution keeps anyway
#pragma unroll
            for (int j = 0; j < MW; ++j) {
                nneg += dinv[j] < 0.0 ? 1 : 0;
                if (!(fabs(dinv[j]) < INFINITY)) bad = 1;          // zero or NaN pivot
            }
            if (nneg != 2) bad = 1;
        }
#pragma unroll
        for (int jj = 0; jj < MW; ++jj) {
            const int j = MW - 1 - jj;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double v = Yv[j][c] * dinv[j];
#pragma unroll
                for (int a = 0; a < MW; ++a)
                    if (a > j) v -= KP(a, j) * Yv[a][c];
                Yv[j][c] = v;
            }
        }
#undef KP
        // own two columns of Y = Kloc^-1 [G | rloc] and of the Schur complement G'Y
        double* Yo = S.Y + (size_t)pr * (MW * 4) + (hi ? 2 : 0);
#pragma unroll
        for (int a = 0; a < MW; ++a) { Yo[4 * a] = Yv[a][0]; Yo[4 * a + 1] = Yv[a][1]; }
        double* So = S.Sloc + (size_t)pr * 12 + (hi ? 2 : 0);
        {
            double s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0, s20 = 0.0, s21 = 0.0;
#pragma unroll
            for (int e = 0; e < MW; ++e) {
                s00 += G0[e] * Yv[e][0]; s01 += G0[e] * Yv[e][1];
                s10 += G1[e] * Yv[e][0]; s11 += G1[e] * Yv[e][1];
                s20 += G2[e] * Yv[e][0]; s21 += G2[e] * Yv[e][1];
            }
            So[0] = s00; So[1] = s01; So[4] = s10; So[5] = s11; So[8] = s20; So[9] = s21;
        }
    }
    SYNC();
    // fold the Schur complements into the stage blocks
    for (int k = lane; k <= L.N; k += NT) {
        double* H = S.Lall + 36 * k;
        double* lv = S.lall + 8 * k;
        for (int i = 0; i < L.nO; ++i) {
            const double* So = S.Sloc + (size_t)(k * L.nO + i) * 12;
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b <= a; ++b) H[a * (a + 1) / 2 + b] -= 0.5 * (So[4 * a + b] + So[4 * b + a]);
                lv[a] += So[4 * a + 3];
            }
        }
    }
    SYNC();
    return red_or(bad);
}

// LU of I + Ppp E without pivoting (the pivots are those of I + E^1/2 Ppp E^1/2: all positive iff the elastic dynamics
// rows leave the value function convex), kept as factors: the sweep never needs the inverse itself, only products of it
// with a few vectors -- (I + Ppp E)^-1 r by substitution, and its transpose through E (I + Ppp E)^-1 E^-1.
struct Lu3 { double l10, l20, l21, u01, u02, u12, i0, i1, i2; };
// (explicit fma chains: the one- and four-wavefront instantiations must round alike, whatever the optimiser would contract)
__device__ __forceinline__ double dot3(double a0, double b0, double a1, double b1, double a2, double b2) {
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}
__device__ __forceinline__ int lu3_factor(const double* P, int ld, const double* E, Lu3& f) {
    const double a00 = fma(P[0], E[0], 1.0), a01 = P[1] * E[1], a02 = P[2] * E[2];
    const double a10 = P[ld] * E[0], a11 = fma(P[ld + 1], E[1], 1.0), a12 = P[ld + 2] * E[2];
    const double a20 = P[2 * ld] * E[0], a21 = P[2 * ld + 1] * E[1], a22 = fma(P[2 * ld + 2], E[2], 1.0);
    int bad = !(a00 > 0.0);
    f.i0 = rcp64(a00);
    f.l10 = a10 * f.i0; f.l20 = a20 * f.i0;
    const double b11 = fma(-f.l10, a01, a11), b12 = fma(-f.l10, a02, a12), b21 = fma(-f.l20, a01, a21), b22 = fma(-f.l20, a02, a22);
    bad |= !(b11 > 0.0);
    f.i1 = rcp64(b11);
    f.l21 = b21 * f.i1;
    const double c22 = fma(-f.l21, b12, b22);
    bad |= !(c22 > 0.0);
    f.u01 = a01; f.u02 = a02; f.u12 = b12;
    f.i2 = rcp64(c22);
    return bad;
}
__device__ __forceinline__ void lu3_solve(const Lu3& f, double r0, double r1, double r2, double& x0, double& x1, double& x2) {
    r1 = fma(-f.l10, r0, r1);
    r2 = fma(-f.l20, r0, r2);
    r2 = fma(-f.l21, r1, r2);
    x2 = r2 * f.i2;
    x1 = fma(-f.u12, x2, r1) * f.i1;
    x0 = fma(-f.u02, x2, fma(-f.u01, x1, r0)) * f.i0;
}

// entry e = 8 a + b of the stage's [F G] = d xi_{k+1} / d (xi_k, du_k), xi = (dp, du_prev, dT): rows 0-2 the pose
// update p + h (v cos, v sin, w) (column 5: its T-derivative), rows 3-4 pick du_k, row 5 carries dT.  Only ONE
// stage's 6 x 8 block is kept in LDS: stage k-1's is written while phase B of stage k runs.
__device__ __forceinline__ double fg_entry(const Lay& L, const Sh& S, const Inst& in, const double* xv, double h, int k, int e) {
    const int a = e >> 3, b = e & 7;
    const double cs = S.ct[k], sn = S.st[k];
    const double* u = S.uv + L.uvs * k + L.uvo;
    double v = 0.0;
    if (a < 3) {
        if (b < 3) v = (a == b) ? 1.0 : 0.0;
        if (b == 2) { if (a == 0) v = -h * u[0] * sn; else if (a == 1) v = h * u[0] * cs; }
        if (b == 5 && L.free_T) v = in.Ts * ((a == 0) ? u[0] * cs : (a == 1) ? u[0] * sn : u[1]);
        if (b == 6) v = (a == 0) ? h * cs : (a == 1) ? h * sn : 0.0;
        if (b == 7) v = (a == 2) ? h : 0.0;
    } else if (a < 5) {
        v = (b == 6 + (a - 3)) ? 1.0 : 0.0;
    } else {
        v = (b == 5) ? 1.0 : 0.0;
    }
    return v;
}

// ---------------------------------------------------------------- two-sided sweep (four-wavefront kernels)
// With four wavefronts per instance the serial sweep was the first wavefront's job alone (68 % of the time at N = 20).
// The stage chain is cut at stage m = N/2: the first wavefront runs the backward recursion from stage N down to m
// (cost-to-go P_m, q_m), the SECOND wavefront runs at the same time a forward recursion from stage 0 up to m: the
// cost-to-arrive  W_k(xi) = 1/2 xi' Pi_k xi + pi_k' xi  over xi = (dp_k, du_{k-1}, dT), stages 0..k-1 minimised out.  One
// forward stage eliminates v = (dp_k, du_{k-1}) (5 x 5 LDL^T; at stage 0 du_{-1} = 0 and only dp_0) from
//   W_k + stage block + 1/2 |dp_{k+1} - [F G]_p (xi_k, du_k) + ghat|^2_{E^-1}
// and lands on (dp_{k+1}, du_k, dT) = xi_{k+1}.  The halves meet in one 6 x 6 solve (Pi_m + P_m) xi_m = -(pi_m + q_m) and
// are recovered outwards at the same time.  Inertia test as before, by additivity: every eliminated block positive
// definite.  Blueprint and its check against the dense solve: oracle/kkt_structured.py (split > 0),
// tests/test_kkt_structured.py.  Each wavefront only reads what it wrote itself until the halves meet, so the stages
// are separated by wavefront-level fences (LDS operations of one wavefront complete in order), not workgroup barriers.
#if OBCA_NT >= 256
#ifndef OBCA_TWO_SIDED_DMAX
#define OBCA_TWO_SIDED_DMAX 1.0e6
#endif
#ifndef OBCA_SPLIT_NUM      /* the halves meet at stage m = N * NUM / 20: a forward stage (5 x 5 elimination) costs about 1.2-1.4 backward stages; 8, 9 and 10 were timed on C3: 387 / 374 / 380 ms per 8192 free-time solves */
#define OBCA_SPLIT_NUM 9
#endif
#define WSYNC() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// LDL^T without pivoting of a packed lower triangle (entry (i, j), i >= j, at i (i + 1) / 2 + j), in place: unit factor
// below the diagonal, reciprocal pivots in id.  Returns 1 on a non-positive pivot.
template <int n>
__device__ __forceinline__ int ldl_factor(double* s, double* id) {
    int bad = 0;
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const double d = s[j * (j + 1) / 2 + j];
        bad |= !(d > 0.0);
        id[j] = rcp64(d);
        double c[n];
#pragma unroll
        for (int i = j + 1; i < n; ++i) c[i] = s[i * (i + 1) / 2 + j];
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            const double l = c[i] * id[j];
            s[i * (i + 1) / 2 + j] = l;
#pragma unroll
            for (int t = j + 1; t <= i; ++t) s[i * (i + 1) / 2 + t] = fma(-l, c[t], s[i * (i + 1) / 2 + t]);
        }
    }
    return bad;
}
template <int n>
__device__ __forceinline__ void ldl_solve(const double* s, const double* id, double* y) {
#pragma unroll
    for (int j = 0; j < n; ++j)
#pragma unroll
        for (int i = j + 1; i < n; ++i) y[i] = fma(-s[i * (i + 1) / 2 + j], y[j], y[i]);
#pragma unroll
    for (int j = 0; j < n; ++j) y[j] *= id[j];
#pragma unroll
    for (int j = n - 1; j >= 0; --j)
#pragma unroll
        for (int i = j + 1; i < n; ++i) y[j] = fma(-s[i * (i + 1) / 2 + j], y[i], y[j]);
}
__device__ __forceinline__ double sym6(const double* P, int a, int b) { return a >= b ? P[6 * a + b] : P[6 * b + a]; }

// Forward half, run by ONE wavefront (lane = 0..63 within it).  Pi_{k+1}, pi_{k+1} go to slot k of Pk / qk (the backward
// half uses the slots m..N), the recovery map v = -(Z (xi_{k+1}; 1)) of stage k to Zk + 36 k (column c at 5 c).
// Lane (a, b) of the first 36 produces entry (a, b) of Pi_{k+1}, lanes 36..41 entry a of pi_{k+1}; every lane factors the
// stage's 5 x 5 block itself (redundant arithmetic is free here, a round trip through LDS is not).
__device__ __forceinline__ int riccati_forward_half(const Lay& L, const Sh& S, const Inst& in, int lane, int m) {
    const double* xv = S.x;
    const double T = L.free_T ? *S.Tv : 1.0;
    const double h = T * in.Ts;
    int bad = 0;
    if (lane < 42) {            // cost-to-arrive at stage 0: the elastic initial condition 1/2 |dp_0 + ghat|^2_{E^-1}
        double v = 0.0;
        if (lane < 36) { const int a = lane / 6, b = lane - 6 * a; if (a == b && a < 3) v = S.Es[L.r_init + a]; }
        else { const int a = lane - 36; if (a < 3) v = S.Es[L.r_init + a] * S.gs[L.r_init + a]; }
        S.Pi0[lane] = v;
    }
    WSYNC();
    const int a = lane < 36 ? lane / 6 : (lane < 42 ? lane - 36 : 0);       // row: index into w = (dp'(0:3), du(3:5), dT(5))
    const int b = lane < 36 ? lane - 6 * (lane / 6) : 6;                    // column of w, or 6: the gradient
    for (int k = 0; k < m; ++k) {
        const double* Pi = k == 0 ? S.Pi0 : S.Pk + 36 * (k - 1);
        const double* pi = k == 0 ? S.Pi0 + 36 : S.qk + 6 * (k - 1);
        const double* Lk = S.Lall + 36 * k;         // over z = (dp(0:3), du_prev(3:5), dT(5), du(6:8)), packed like s below
        const double* lk = S.lall + 8 * k;
        double D[3], gh[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { D[j] = S.Es[L.r_dyn + 3 * k + j]; gh[j] = S.gs[L.r_dyn + 3 * k + j]; }
        const double cs = S.ct[k], sn = S.st[k], u0 = S.uv[L.uvs * k + L.uvo], u1 = S.uv[L.uvs * k + L.uvo + 1];
        const double a02 = -h * u0 * sn, a12 = h * u0 * cs;         // d pose' / d theta: A = I + a02 e0 e2' + a12 e1 e2'
        double tc[3] = {0.0, 0.0, 0.0};
        if (L.free_T) { tc[0] = in.Ts * u0 * cs; tc[1] = in.Ts * u0 * sn; tc[2] = in.Ts * u1; }
        // ---- the eliminated block  S_vv = Pi_vv + L_vv + A' E^-1 A
        double s[15], id[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int l = 0; l <= j; ++l) s[j * (j + 1) / 2 + l] = Pi[6 * j + l] + Lk[j * (j + 1) / 2 + l];
        s[0] += D[0];
        s[2] += D[1];
        s[3] = fma(a02, D[0], s[3]);
        s[4] = fma(a12, D[1], s[4]);
        s[5] += fma(a02 * a02, D[0], fma(a12 * a12, D[1], D[2]));
        if (k == 0) {           // du_{-1} is data: rows 3, 4 leave the system
            s[6] = 0.0; s[7] = 0.0; s[8] = 0.0; s[9] = 1.0; s[10] = 0.0; s[11] = 0.0; s[12] = 0.0; s[13] = 0.0; s[14] = 1.0;
        }
        bad |= ldl_factor<5>(s, id);
        // ---- columns a and b of [S_vw c_v]: r = m + A' E^-1 f, f the column's coefficient in the dynamics residual
        // (dp'_c: -e_c;  du_c: B e_c;  dT: the T column;  gradient: -ghat), m its entries in Pi + stage block
        double fa[3], fb[3], ra[5], rb[5];
        auto column = [&](int c, double* f, double* r) {
            const int zi = c < 5 ? 3 + c : 5;           // (c = 3, 4 -> z index 6, 7;  c = 5 -> 5)
            f[0] = c < 3 ? (c == 0 ? -1.0 : 0.0) : c == 3 ? h * cs : c == 4 ? 0.0 : c == 5 ? tc[0] : -gh[0];
            f[1] = c < 3 ? (c == 1 ? -1.0 : 0.0) : c == 3 ? h * sn : c == 4 ? 0.0 : c == 5 ? tc[1] : -gh[1];
            f[2] = c < 3 ? (c == 2 ? -1.0 : 0.0) : c == 3 ? 0.0 : c == 4 ? h : c == 5 ? tc[2] : -gh[2];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                double v = 0.0;
                if (c >= 3) v = c == 6 ? lk[j] : Lk[zi * (zi + 1) / 2 + j];
                if (c >= 5) v += c == 6 ? pi[j] : Pi[30 + j];
                r[j] = v;
            }
            const double g0 = D[0] * f[0], g1 = D[1] * f[1], g2 = D[2] * f[2];
            r[0] += g0;
            r[1] += g1;
            r[2] += fma(a02, g0, fma(a12, g1, g2));
            if (k == 0) { r[3] = 0.0; r[4] = 0.0; }
        };
        // (column b first, through the solve; column a only afterwards, when the factor's registers are free again)
        column(b, fb, rb);
        ldl_solve<5>(s, id, rb);                    // rb: column b of Z = S_vv^-1 [S_vw c_v]
        column(a, fa, ra);
        // entry (a, b) of [S_ww c_w] - S_wv Z
        double out = fma(fa[0] * D[0], fb[0], fma(fa[1] * D[1], fb[1], fa[2] * D[2] * fb[2]));
        if (a >= 3 && b >= 3) {
            const int za = a < 5 ? 3 + a : 5, zb = b < 5 ? 3 + b : 5;
            if (b < 6) {
                out += Lk[LS(za, zb)];
                if (a == 5 && b == 5) out += Pi[35];
            } else {
                out += lk[za];
                if (a == 5) out += pi[5];
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) out = fma(-ra[j], rb[j], out);
        if (lane < 36) S.Pk[36 * k + lane] = out;
        else if (lane < 42) S.qk[6 * k + a] = out;
        if (a == 0 && lane < 42) {
#pragma unroll
            for (int j = 0; j < 5; ++j) S.Zk[36 * k + 5 * b + j] = rb[j];
        }
        WSYNC();
        if (__any(bad)) break;
    }
    return bad;
}
#endif

// ---------------------------------------------------------------- level 2: Riccati sweep + forward pass
// Two LDS round trips per stage: (A) every lane applies P~ to ONE column and produces one entry of the lower triangle
// of the symmetric 8x8 stage matrix Mall = Lall + [F G]' P~ [F G] (36 lanes) or of its gradient (8 lanes, same expression);
// (B) every lane inverts the 2x2 input block and produces one entry of P_k / q_k / K.  Returns 1 on a wrong-sign pivot; on success dx (poses, inputs, T) and the multiplier steps
// of the soft rows are written.
#if OBCA_NT >= 256
// Stage where the two halves of the sweep meet (0: one-sided sweep).  The forward half carries the elastic rows in
// information form (E^-1 enters its blocks), which loses digits to cancellation once E^-1 is huge (the last few
// iterations, when the elastic variables vanish: E ~ 1e-10 and below); the backward half's (I + P E)^-1 form does not.
// Measured on the blueprint (oracle/kkt_structured.py): step error <= 1e-11 relative while E >= 1e-6, up to 1e-8 below;
// in the kernel itself (-DOBCA_PROFILE -DOBCA_TWO_SIDED_CHECK: every two-sided solve repeated one-sided): see DESIGN.md.
// So the two-sided sweep serves the iterations with max E^-1 <= OBCA_TWO_SIDED_DMAX -- 96 % of the solves of C3's free-time
// half, 79 % of its gated half -- and the others run one-sided.  (Every wavefront evaluates the test itself: same data, same result, no barrier.)
__device__ __forceinline__ int two_sided_split(const Lay& L, const Sh& S, int lane) {
    int m = L.N >= 4 ? L.N * OBCA_SPLIT_NUM / 20 : 0;
    if (m > 0) {
        double dmax = 0.0;
        for (int r = L.r_init + (lane & 63); r < L.r_term; r += 64) dmax = fmax(dmax, S.Es[r]);
        if (wave_max(dmax) > OBCA_TWO_SIDED_DMAX) m = 0;
    }
    return m;
}
#endif
#ifdef OBCA_PROFILE
__device__ __forceinline__ int riccati(const Lay& L, const Sh& S, const Inst& in, int lane, long long* prof_t, bool allow_two = true) {
    long long rlast = wall_clock64();
#else
__device__ __forceinline__ int riccati(const Lay& L, const Sh& S, const Inst& in, int lane, bool allow_two = true) {
#endif
#ifdef NO_RICCATI
    return 0;
#endif
    const double* xv = S.x;
    const double T = L.free_T ? *S.Tv : 1.0;
    const double h = T * in.Ts;
    int bad = 0;
#if OBCA_NT >= 256
#define RSYNC() WSYNC()
    const int m = allow_two ? two_sided_split(L, S, lane) : 0;
    if (lane < 64) {
#else
#define RSYNC() SYNC()
    constexpr int m = 0;
#endif
    // [F G] of the first stage of the sweep, and the terminal value function
    for (int t = lane; t < 48; t += NT) S.FG[t] = fg_entry(L, S, in, xv, h, L.N - 1, t);
    {
        double* PN = S.Pk + 36 * L.N;
        double* qN = S.qk + 6 * L.N;
        if (lane < 36) {
            const int a = lane / 6, b = lane - 6 * a;
            PN[lane] = (a < 3 && b < 3) ? S.Lall[36 * L.N + LS(a, b)] : 0.0;
        } else if (lane < 42) {
            const int a = lane - 36;
            qN[a] = (a < 3) ? S.lall[8 * L.N + a] : 0.0;
        }
    }
    RSYNC();
    RPROF(12)
    for (int k = L.N - 1; k >= m; --k) {
        // ---- phase A ----------------------------------------------------------------------------------
        // Entry (a, b) of Mall = Lall + [F G]' P~ [F G] WITHOUT forming P~ = soft-min of (P, E): with M = (I + Ppp E)^-1,
        //   P~ f = ( M (Ppp f_p + Ppo f_o) ,  Poo f_o + Pop E M (E^-1 f_p - Ppo f_o) )
        // (second block through M' = E M E^-1), i.e. one 3x3 LU per lane and two substitutions instead of the explicit
        // inverse, the 3x6 product M [Ppp Ppo], the symmetrised 6x6 P~ and a 6x6 quadratic form -- same pivots, same inertia test.
        double E[3], Dv[3], gh[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { Dv[j] = S.Es[L.r_dyn + 3 * k + j]; E[j] = rcp64(Dv[j]); gh[j] = S.gs[L.r_dyn + 3 * k + j]; }
        if (NT == 64 || lane < 64) {        // (four wavefronts: the serial sweep is the first wavefront's job alone --
            // the others would only repeat it and compete for the LDS)
            const double* Pl = S.Pk + 36 * (k + 1);
            const double* ql = S.qk + 6 * (k + 1);
            // P is used in two halves (rows 0-2: LU and the two right-hand sides; rows 3-5: the second block of P~ f), loaded
            // one after the other -- all 36 entries at once were the register peak of the sweep
            double Pa[18], Pb[18];
#pragma unroll
            for (int i = 0; i < 18; ++i) Pa[i] = Pl[i];
            Lu3 lu;
            bad |= lu3_factor(Pa, 6, E, lu);
            // lanes 0..35: entry (a, b), a >= b, of the SYMMETRIC stage matrix (packed like the stage blocks: lane = LS(a, b));
            // lanes 36..43: entry a of its gradient, computed by the same expression: with f = (-ghat, 0) as the "column",
            //   lall + [F G]' (q~ - P~ (ghat, 0)) = lall + [F G]' (P~ f + q~),   q~ = (M q_p, q_o - Pop E M q_p),
            // i.e. q_p joins the right-hand sides of the two substitutions and q_o the second block -- no separate pass
            const bool isg = lane >= 36;
            const int a = isg ? (lane - 36) & 7 : (lane >= 28) ? 7 : (lane >= 21) ? 6 : (lane >= 15) ? 5 : (lane >= 10) ? 4 : (lane >= 6) ? 3 : (lane >= 3) ? 2 : (lane >= 1) ? 1 : 0;
            const int b = isg ? 0 : lane - a * (a + 1) / 2;
            const double gm = isg ? 1.0 : 0.0;
            double fb[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double fgc = S.FG[8 * c + b];
                fb[c] = isg ? (c < 3 ? -gh[c] : 0.0) : fgc;
            }
            double w3[3], t3[3], u3[3], yp[3], s2[3], yo[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                w3[i] = dot3(Pa[6 * i + 3], fb[3], Pa[6 * i + 4], fb[4], Pa[6 * i + 5], fb[5]);          // Ppo f_o
                const double qi = gm * ql[i];
                t3[i] = dot3(Pa[6 * i], fb[0], Pa[6 * i + 1], fb[1], Pa[6 * i + 2], fb[2]) + w3[i] + qi;
                u3[i] = fma(Dv[i], fb[i], -w3[i]) - qi;
            }
            lu3_solve(lu, t3[0], t3[1], t3[2], yp[0], yp[1], yp[2]);
            lu3_solve(lu, u3[0], u3[1], u3[2], s2[0], s2[1], s2[2]);
#pragma unroll
            for (int i = 0; i < 18; ++i) Pb[i] = Pl[18 + i];
            double fa[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) fa[c] = S.FG[8 * c + a];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                yo[i] = dot3(Pb[6 * i + 3], fb[3], Pb[6 * i + 4], fb[4], Pb[6 * i + 5], fb[5]) +
                        dot3(Pb[6 * i], E[0] * s2[0], Pb[6 * i + 1], E[1] * s2[1], Pb[6 * i + 2], E[2] * s2[2]) + gm * ql[3 + i];
            double v = isg ? S.lall[8 * k + a] : S.Lall[36 * k + lane];
#pragma unroll
            for (int c = 0; c < 3; ++c) v = fma(fa[3 + c], yo[c], fma(fa[c], yp[c], v));
            if (lane < 36) S.Mall[lane] = v;
            else if (lane < 44) S.mall[a] = v;
            // the factors, for the forward pass (nine numbers, like the inverse they replace)
            {
                const double fsel = lane == 0 ? lu.l10 : lane == 1 ? lu.l20 : lane == 2 ? lu.l21 : lane == 3 ? lu.u01 : lane == 4 ? lu.u02
                                  : lane == 5 ? lu.u12 : lane == 6 ? lu.i0 : lane == 7 ? lu.i1 : lu.i2;
                if (lane < 9) S.Mik[9 * k + lane] = fsel;
            }
        }
        RSYNC();
        RPROF(13)
        // ---- phase B ----------------------------------------------------------------------------------
        if (NT == 64 || lane < 64) {
        const double m00 = S.Mall[LS(6, 6)], m01 = S.Mall[LS(7, 6)], m11 = S.Mall[LS(7, 7)];
        const double d1 = m11 - m01 * m01 * rcp64(m00);
        if (!(m00 > 0.0) || !(d1 > 0.0)) bad = 1;
        const double idet = rcp64(m00 * d1);
        const double i00 = m11 * idet, i01 = -m01 * idet, i11 = m00 * idet;
        if (lane < 42) {
            const int a = (lane < 36) ? lane / 6 : lane - 36, b = (lane < 36) ? lane - 6 * (lane / 6) : 0;
            const double xa0 = S.Mall[LS(6, a)], xa1 = S.Mall[LS(7, a)];
            if (lane < 36) {            // P_k = Mxx - Mxu Muu^-1 Mxu'
                const double xb0 = S.Mall[LS(6, b)], xb1 = S.Mall[LS(7, b)];
                const double kb0 = -(i00 * xb0 + i01 * xb1), kb1 = -(i01 * xb0 + i11 * xb1);
                S.Pk[36 * k + lane] = S.Mall[LS(a, b)] + xa0 * kb0 + xa1 * kb1;
                if (a == 0) { S.Kk[12 * k + b] = kb0; S.Kk[12 * k + 6 + b] = kb1; }      // K = -Muu^-1 Mxu'
            } else {
                const double k0 = -(i00 * S.mall[6] + i01 * S.mall[7]), k1 = -(i01 * S.mall[6] + i11 * S.mall[7]);
                S.qk[6 * k + a] = S.mall[a] + xa0 * k0 + xa1 * k1;
                if (a == 0) { S.kapk[2 * k] = k0; S.kapk[2 * k + 1] = k1; }
            }
        }
        }
        if (k > m) {                        // phase A of this stage is over: its [F G] can make room for the next one
            // (with two wavefronts the second one writes it: it has no entry of P_k to compute)
            const int t = NT == 128 ? lane - 64 : lane;
            if (t >= 0 && t < 48) S.FG[t] = fg_entry(L, S, in, xv, h, k - 1, t);
        }
        RSYNC();
        RPROF(14)
#if OBCA_NT >= 256
        if (__any(bad)) break;
#else
        if (red_or(bad)) return 1;         // wrong-sign pivot: the attempt is over, no need to finish the sweep
#endif
    }
#if OBCA_NT >= 256
    } else if (lane < 128 && m > 0) {
        bad |= riccati_forward_half(L, S, in, lane - 64, m);
    }
    SYNC();                                 // the halves meet
    if (red_or(bad)) return 1;
#endif
    // stage 0: du_{-1} = 0, elastic initial condition, then the time scale
    double E0[3], D0[3], g0[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { D0[j] = S.Es[L.r_init + j]; E0[j] = rcp64(D0[j]); g0[j] = S.gs[L.r_init + j]; }
    Lu3 lu0;
    double s1[3] = {0.0, 0.0, 0.0}, X55 = 1.0, qt5 = 0.0;
    if (m == 0) {
    if (NT == 64 || lane < 64) {
        const double* P0 = S.Pk;
        const double* q0 = S.qk;
        bad |= lu3_factor(P0, 6, E0, lu0);
        // row / column T of P~: M Ppo e_T, and the (T, T) entry; q~_T
        double zq[3];
        lu3_solve(lu0, P0[5], P0[11], P0[17], s1[0], s1[1], s1[2]);
        lu3_solve(lu0, q0[0], q0[1], q0[2], zq[0], zq[1], zq[2]);
        X55 = P0[35] - dot3(P0[30], E0[0] * s1[0], P0[31], E0[1] * s1[1], P0[32], E0[2] * s1[2]);
        qt5 = q0[5] - dot3(P0[30], E0[0] * zq[0], P0[31], E0[1] * zq[1], P0[32], E0[2] * zq[2]);
        if (L.free_T && !(X55 > 0.0)) bad = 1;
    }
    bad = red_or(bad);
    RPROF(15)
    if (bad) return 1;
    }
    double dT = 0.0;
    double dp[3] = {0.0, 0.0, 0.0}, up[2] = {0.0, 0.0};
#if OBCA_NT >= 256
    if (m > 0) {
        // ---- the halves meet: (Pi_m + P_m) xi_m = -(pi_m + q_m), by the first two wavefronts (each for its own half)
        double xi[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (lane < 128) {
            const double* Pb = S.Pk + 36 * m;          // cost-to-go of stage m
            const double* Pf = S.Pk + 36 * (m - 1);    // cost-to-arrive at stage m
            double sm[21], idm[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#pragma unroll
                for (int l = 0; l <= j; ++l) sm[j * (j + 1) / 2 + l] = Pf[6 * j + l] + 0.5 * (Pb[6 * j + l] + Pb[6 * l + j]);
                xi[j] = -(S.qk[6 * (m - 1) + j] + S.qk[6 * m + j]);
            }
            bad |= ldl_factor<6>(sm, idm);
            ldl_solve<6>(sm, idm, xi);
        }
        bad = red_or(bad);
        RPROF(15)
        if (bad) return 1;
        dp[0] = xi[0]; dp[1] = xi[1]; dp[2] = xi[2]; up[0] = xi[3]; up[1] = xi[4]; dT = xi[5];
        if (lane >= 64 && lane < 128) {
            // ---- forward half recovered downwards by the second wavefront: (dp_k, du_{k-1}) = -Z_k (xi_{k+1}; 1); the
            // multiplier steps of the elastic dynamics rows are the gradient of the cost-to-arrive
            const bool w0 = lane == 64;
            if (w0) {
                S.dx[L.ip(m)] = xi[0]; S.dx[L.ip(m) + 1] = xi[1]; S.dx[L.ip(m) + 2] = xi[2];
                S.dx[L.iu(m - 1)] = xi[3]; S.dx[L.iu(m - 1) + 1] = xi[4];
                if (L.free_T) S.dx[L.iT()] = dT;
            }
            for (int k = m - 1; k >= 0; --k) {
                const double* Z = S.Zk + 36 * k;
                double v[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    double acc = Z[30 + j];
#pragma unroll
                    for (int c = 0; c < 6; ++c) acc = fma(Z[5 * c + j], xi[c], acc);
                    v[j] = -acc;
                }
                if (w0) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double acc = S.qk[6 * k + a];
#pragma unroll
                        for (int c = 0; c < 6; ++c) acc = fma(sym6(S.Pk + 36 * k, a, c), xi[c], acc);
                        S.dy[L.r_dyn + 3 * k + a] = acc;
                        S.dx[L.ip(k) + a] = v[a];
                    }
                    if (k > 0) { S.dx[L.iu(k - 1)] = v[3]; S.dx[L.iu(k - 1) + 1] = v[4]; }
                }
                xi[0] = v[0]; xi[1] = v[1]; xi[2] = v[2]; xi[3] = v[3]; xi[4] = v[4];
            }
            if (w0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) S.dy[L.r_init + a] = D0[a] * (xi[a] + g0[a]);
            }
        }
    }
#endif
    // ---- forward pass: every lane (of the first wavefront) carries the (tiny) state redundantly, lane 0 stores
    if (NT == 64 || lane < 64) {
    if (m == 0) {
    if (L.free_T) dT = -(qt5 - dot3(s1[0], g0[0], s1[1], g0[1], s1[2], g0[2])) / X55;
    {
        const double* P0 = S.Pk;
        const double* q0 = S.qk;
        double t[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a] = -g0[a] - E0[a] * (P0[6 * a + 5] * dT + q0[a]);
        // dp = M' t = E M (E^-1 t)
        double m0, m1, m2;
        lu3_solve(lu0, D0[0] * t[0], D0[1] * t[1], D0[2] * t[2], m0, m1, m2);
        dp[0] = E0[0] * m0; dp[1] = E0[1] * m1; dp[2] = E0[2] * m2;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (lane == 0) S.dy[L.r_init + a] = -(P0[6 * a] * dp[0] + P0[6 * a + 1] * dp[1] + P0[6 * a + 2] * dp[2] + P0[6 * a + 5] * dT + q0[a]);
    }
    if (lane == 0) {
        S.dx[0] = dp[0]; S.dx[1] = dp[1]; S.dx[2] = dp[2];
        if (L.free_T) S.dx[L.iT()] = dT;
    }
    }
    for (int k = m; k < L.N; ++k) {
        // all operands of the stage first (independent LDS reads, one wait), then the arithmetic
        double Kg[12], P1[18], Mi[9], q1[3], E[3], gh[3];
#pragma unroll
        for (int a = 0; a < 12; ++a) Kg[a] = S.Kk[12 * k + a];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int c = 0; c < 6; ++c) P1[6 * a + c] = S.Pk[36 * (k + 1) + 6 * a + c];
            q1[a] = S.qk[6 * (k + 1) + a];
            E[a] = S.Es[L.r_dyn + 3 * k + a];
            gh[a] = S.gs[L.r_dyn + 3 * k + a];
        }
#pragma unroll
        for (int a = 0; a < 9; ++a) Mi[a] = S.Mik[9 * k + a];
        const double kap0 = S.kapk[2 * k], kap1 = S.kapk[2 * k + 1];
        const double cs = S.ct[k], sn = S.st[k];
        const double uk0 = S.uv[L.uvs * k + L.uvo], uk1 = S.uv[L.uvs * k + L.uvo + 1];
        const double xi[6] = {dp[0], dp[1], dp[2], up[0], up[1], dT};
        double u[2] = {kap0, kap1};
#pragma unroll
        for (int a = 0; a < 6; ++a) { u[0] += Kg[a] * xi[a]; u[1] += Kg[6 + a] * xi[a]; }
        double ph[3];
        ph[0] = dp[0] - h * uk0 * sn * dp[2] + h * cs * u[0] - gh[0];
        ph[1] = dp[1] + h * uk0 * cs * dp[2] + h * sn * u[0] - gh[1];
        ph[2] = dp[2] + h * u[1] - gh[2];
        if (L.free_T) { ph[0] += in.Ts * uk0 * cs * dT; ph[1] += in.Ts * uk0 * sn * dT; ph[2] += in.Ts * uk1 * dT; }
        double t[3], dn[3], dyv[3];
        const double iE[3] = {rcp64(E[0]), rcp64(E[1]), rcp64(E[2])};
#pragma unroll
        for (int a = 0; a < 3; ++a)
            t[a] = ph[a] - (P1[6 * a + 3] * u[0] + P1[6 * a + 4] * u[1] + P1[6 * a + 5] * dT + q1[a]) * iE[a];
        {   // dn = M' t = E M (E^-1 t); here E[] holds the E^-1 of the formula (S.Einv of the dynamics rows)
            Lu3 lf;
            lf.l10 = Mi[0]; lf.l20 = Mi[1]; lf.l21 = Mi[2]; lf.u01 = Mi[3]; lf.u02 = Mi[4]; lf.u12 = Mi[5]; lf.i0 = Mi[6]; lf.i1 = Mi[7]; lf.i2 = Mi[8];
            double m0, m1, m2;
            lu3_solve(lf, E[0] * t[0], E[1] * t[1], E[2] * t[2], m0, m1, m2);
            dn[0] = m0 * iE[0]; dn[1] = m1 * iE[1]; dn[2] = m2 * iE[2];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            dyv[a] = -(P1[6 * a] * dn[0] + P1[6 * a + 1] * dn[1] + P1[6 * a + 2] * dn[2] + P1[6 * a + 3] * u[0] +
                       P1[6 * a + 4] * u[1] + P1[6 * a + 5] * dT + q1[a]);
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { S.dy[L.r_dyn + 3 * k + a] = dyv[a]; S.dx[L.ip(k + 1) + a] = dn[a]; }
            S.dx[L.iu(k)] = u[0];
            S.dx[L.iu(k) + 1] = u[1];
        }
        dp[0] = dn[0]; dp[1] = dn[1]; dp[2] = dn[2];
        up[0] = u[0]; up[1] = u[1];
    }
    }
    SYNC();
    RPROF(16)
    // local recovery: [dw; dnu] = Y_r - Y_G dp_k
    for (int pr = lane; pr < L.npair; pr += NT) {
        const int k = pr / L.nO, i = pr - k * L.nO;
        const int o0 = S.offm[i], m = S.offm[i + 1] - o0;
        const double* Yo = S.Y + (size_t)pr * (MW * 4);
        const double* d = S.dx + L.ip(k);
        for (int a = 0; a < MW; ++a) {
            const double v = Yo[4 * a + 3] - (Yo[4 * a] * d[0] + Yo[4 * a + 1] * d[1] + Yo[4 * a + 2] * d[2]);
            if (a < OBCA_MAX_EDGES) { if (a < m) S.dx[L.il(k) + o0 + a] = v; }
            else if (a < NW) S.dx[L.imu(k) + 4 * i + (a - OBCA_MAX_EDGES)] = v;
            else S.dnu[2 * pr + (a - NW)] = v;
        }
    }
    SYNC();
    RPROF(17)
    return 0;
}

// The window start of the ladder (oracle/ipm_dense.py:window_start): poses of the reference window (first pose x0), inputs
// by differences clipped to their box, free-time problem (iT >= 0): the time scale at which the window is driven at
// OBCA_WINDOW_SPEED_FRAC of the speed bound; lambda = mu = 0 (x is zero on entry).  Rare, one lane, out of line with scalar
// arguments only: inlined into the body it cost the four-wavefront kernels 400 B of scratch.
__device__ __noinline__ void window_start_point(double* x, const double* xref, const Inst* inp, int N, int NS, int iT) {
    const Inst& in = *inp;
    const int N1 = N + 1;
    double len = 0.0;
    for (int k = 0; k <= N; ++k) {
        for (int j = 0; j < 3; ++j) x[k * NS + j] = (k == 0) ? in.x0[j] : xref[j * N1 + k];
        if (k > 0) {
            const double ddx = x[k * NS] - x[(k - 1) * NS], ddy = x[k * NS + 1] - x[(k - 1) * NS + 1];
            len += sqrt(ddx * ddx + ddy * ddy);
        }
    }
    double h = in.Ts;
    if (iT >= 0) {
        const double T0 = fmin(fmax(1.0, len / (N * OBCA_WINDOW_SPEED_FRAC * in.uU[0] * in.Ts)), fmax(1.0, in.Tmax));
        x[iT] = T0;
        h *= T0;
    }
    for (int k = 0; k < N; ++k) {
        const double ddx = x[(k + 1) * NS] - x[k * NS], ddy = x[(k + 1) * NS + 1] - x[k * NS + 1];
        const double dth = x[(k + 1) * NS + 2] - x[k * NS + 2];
        x[k * NS + 3] = fmin(fmax(sqrt(ddx * ddx + ddy * ddy) / h, in.uL[0]), in.uU[0]);
        x[k * NS + 4] = fmin(fmax(dth / h, in.uL[1]), in.uU[1]);
    }
}

// The dodge starts of the ladder's last rung (oracle/ipm_dense.py:dodge_start; csrc/obca_device.h: OBCA_KIND_DODGE_*; fixed-time
// problems only): the window moved sideways by side * OBCA_DODGE_OFFSET (ramped in over OBCA_DODGE_RAMP stages), headings along the
// moved poses, inputs by differences, lambda / mu of every (stage, obstacle) pair on the half-space row with the largest gap.
// Rare, one lane, out of line (as window_start_point).  x is zero on entry.
__device__ __noinline__ void dodge_start_point(double* x, const double* xref, const double* Aobs, const double* bobs, const int* offm,
                                               const Inst* inp, int N, int NS, int M, int nO, double side) {
    const Inst& in = *inp;
    const int N1 = N + 1;
    for (int k = 0; k <= N; ++k) {
        double px = (k == 0) ? in.x0[0] : xref[0 * N1 + k], py = (k == 0) ? in.x0[1] : xref[1 * N1 + k];
        if (k > 0) {
            const int ka = k - 1, kb = k + 1 <= N ? k + 1 : N;
            const double ax = xref[0 * N1 + kb] - ((ka == 0) ? in.x0[0] : xref[0 * N1 + ka]);
            const double ay = xref[1 * N1 + kb] - ((ka == 0) ? in.x0[1] : xref[1 * N1 + ka]);
            const double len = sqrt(ax * ax + ay * ay);
            const double th = xref[2 * N1 + k];
            const double nx = len > 1e-9 ? -ay / len : -sin(th), ny = len > 1e-9 ? ax / len : cos(th);
            const double w = side * OBCA_DODGE_OFFSET * (k < OBCA_DODGE_RAMP ? (double)k / OBCA_DODGE_RAMP : 1.0);
            px += w * nx; py += w * ny;
        }
        x[k * NS] = px; x[k * NS + 1] = py;
    }
    x[2] = in.x0[2];
    for (int k = 1; k <= N; ++k) {
        const double prev = x[(k - 1) * NS + 2];
        double th = prev;
        if (k < N) {
            const double ddx = x[(k + 1) * NS] - x[k * NS], ddy = x[(k + 1) * NS + 1] - x[k * NS + 1];
            if (ddx * ddx + ddy * ddy > 1e-18) {
                double d = atan2(ddy, ddx) - prev;
                d -= 6.283185307179586 * floor(d / 6.283185307179586 + 0.5);
                th = prev + d;
            }
        }
        x[k * NS + 2] = th;
    }
    const double h = in.Ts;
    for (int k = 0; k < N; ++k) {
        const double ddx = x[(k + 1) * NS] - x[k * NS], ddy = x[(k + 1) * NS + 1] - x[k * NS + 1];
        const double dth = x[(k + 1) * NS + 2] - x[k * NS + 2];
        x[k * NS + 3] = fmin(fmax(sqrt(ddx * ddx + ddy * ddy) / h, in.uL[0]), in.uU[0]);
        x[k * NS + 4] = fmin(fmax(dth / h, in.uL[1]), in.uU[1]);
    }
    for (int k = 0; k <= N; ++k) {
        const double th = x[k * NS + 2], ct = cos(th), st = sin(th);
        const double tx = x[k * NS] + ct * in.off, ty = x[k * NS + 1] + st * in.off;
        const int il = k * NS + (k < N ? 5 : 3);
        for (int i = 0; i < nO; ++i) {
            const int o0 = offm[i], o1 = offm[i + 1];
            int jb = o0;
            double gb = -INFINITY, m0b = 0.0, m1b = 0.0, m2b = 0.0, m3b = 0.0, lb = 0.0;
            for (int j = o0; j < o1; ++j) {
                const double a0 = Aobs[(k * M + j) * 2], a1 = Aobs[(k * M + j) * 2 + 1];
                const double nrm = sqrt(a0 * a0 + a1 * a1);
                if (!(nrm > 0.0)) continue;
                const double v0 = a0 / nrm, v1 = a1 / nrm;
                const double r0 = ct * v0 + st * v1, r1 = -st * v0 + ct * v1;
                const double m0 = fmax(-r0, 0.0), m1 = fmax(-r1, 0.0), m2 = fmax(r0, 0.0), m3 = fmax(r1, 0.0);
                const double gap = -(in.gego[0] * m0 + in.gego[1] * m1 + in.gego[2] * m2 + in.gego[3] * m3) + (a0 * tx + a1 * ty - bobs[k * M + j]) / nrm;
                if (gap > gb) { gb = gap; jb = j; lb = 1.0 / nrm; m0b = m0; m1b = m1; m2b = m2; m3b = m3; }
            }
            for (int j = o0; j < o1; ++j) x[il + j] = (j == jb) ? lb : 0.0;
            x[il + M + 4 * i] = m0b; x[il + M + 4 * i + 1] = m1b; x[il + M + 4 * i + 2] = m2b; x[il + M + 4 * i + 3] = m3b;
        }
    }
}

}  // namespace

// ================================================================== the kernel
// The scalar part of a launch descriptor as wave-uniform values.  obca_solve_batch hands the descriptor over as a kernel
// argument (scalar registers from the start); the fused closed-loop kernel reads it from HBM with vector loads, and
// without the readfirstlane below every field -- 20 pointers, the shape -- would sit in vector registers for the whole
// solve (measured: 1.1 KB of scratch per lane).
template <bool FROM_MEMORY> __device__ __forceinline__ int uni(int v) { return FROM_MEMORY ? __builtin_amdgcn_readfirstlane(v) : v; }
template <bool FROM_MEMORY, class T> __device__ __forceinline__ T* uni(T* p) {
    if (!FROM_MEMORY) return p;
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
template <bool FROM_MEMORY> __device__ __forceinline__ double uni(double v) { return FROM_MEMORY ? lane_read(v, 0) : v; }
struct ObcaHead {
    int32_t B, N, nO, M, n_max, R_max, inst_off, two_sided, soc_lds;
    const int32_t* variant;
    const double *x0, *u0, *xref, *A, *b, *Ts, *term;
    double *xopt, *uopt, *ts_opt;
    int32_t *status, *iters;
    double *info, *prof, *warm_z;
    const int32_t* warm_use;
    double warm_mu;
    double *cert_z, *cert_y, *soc_ws, *gm_ws;
    long long gm_stride;
};

typedef const __attribute__((address_space(4))) ObcaLaunch ObcaLaunchConst;   // descriptor in HBM, read through the scalar cache

// The problem shape as the body sees it: ShapeAny -- read from the launch descriptor (any shape the kernel's limits allow);
// ShapeIs<N, nO, M> -- compile-time constants (csrc/obca_device.h: OBCA_SHAPES): same code, same arithmetic, same results,
// but every LDS offset, loop bound and index division folds.  The host only launches an instantiation for its own shape.
struct ShapeAny { static constexpr bool fixed = false; static constexpr int N = 0, nO = 0, M = 0, n_max = 0, R_max = 0, inst_off = 0, soc_lds = 0; };
// FT: the fixed-time variants only (the closed loop's groups with sensed boxes: their layouts have three rows less than the
// free-time layout the handle sizes for, csrc/obca_rollout.hip)
template <int N_, int NO_, int M_, bool FT = false>
struct ShapeIs {
    static constexpr bool fixed = true;
    static constexpr int N = N_, nO = NO_, M = M_;
    static constexpr int n_max = obca_shape_sizes(N_, NO_, M_).n_max, R_max = obca_shape_sizes(N_, NO_, M_).R_max - (FT ? 3 : 0),
                         inst_off = obca_shape_sizes(N_, NO_, M_).inst_off;
#if OBCA_NT == 64
    static constexpr int RPL = R_max <= 256 ? 4 : R_max <= 320 ? 5 : 6;       // rows per lane, as obca_solve_batch picks _r4 / _r5 / _r6
    static_assert(R_max <= 384, "shape beyond the one-wavefront kernels");
    static constexpr int soc_lds = obca_soc_lds_wave(N_, NO_, M_);             // scratch of the second-order correction: LDS offset, 0 = HBM
#else
    static constexpr int RPL = R_max <= 768 ? 3 : -5;                         // four wavefronts: as _mw_r3 / _mw_r5
    static_assert(R_max <= 1280, "shape beyond the four-wavefront LDS kernels");
    static constexpr int soc_lds = obca_soc_lds_mw(N_, NO_, M_);
#endif
};

template <int RPL, bool FROM_MEMORY = false, class DESC = const ObcaLaunch, class SHAPE = ShapeAny>
__device__ __forceinline__ void obca_ipm_body(DESC& Ain, const int inst, const bool first) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    ObcaHead A;
    {
        constexpr bool U = FROM_MEMORY;
        A.B = uni<U>(Ain.B); A.N = uni<U>(Ain.N); A.nO = uni<U>(Ain.nO); A.M = uni<U>(Ain.M); A.n_max = uni<U>(Ain.n_max);
        A.R_max = uni<U>(Ain.R_max); A.inst_off = uni<U>(Ain.inst_off); A.two_sided = uni<U>(Ain.two_sided); A.soc_lds = uni<U>(Ain.soc_lds);
        A.variant = uni<U>(Ain.variant); A.x0 = uni<U>(Ain.x0); A.u0 = uni<U>(Ain.u0); A.xref = uni<U>(Ain.xref); A.A = uni<U>(Ain.A);
        A.b = uni<U>(Ain.b); A.Ts = uni<U>(Ain.Ts); A.term = uni<U>(Ain.term); A.xopt = uni<U>(Ain.xopt); A.uopt = uni<U>(Ain.uopt);
        A.ts_opt = uni<U>(Ain.ts_opt); A.status = uni<U>(Ain.status); A.iters = uni<U>(Ain.iters); A.info = uni<U>(Ain.info);
        A.prof = uni<U>(Ain.prof); A.warm_z = uni<U>(Ain.warm_z); A.warm_use = uni<U>(Ain.warm_use); A.warm_mu = uni<U>(Ain.warm_mu);
        A.cert_z = uni<U>(Ain.cert_z); A.cert_y = uni<U>(Ain.cert_y); A.soc_ws = uni<U>(Ain.soc_ws);
        A.gm_ws = Ain.gm_ws; A.gm_stride = Ain.gm_stride;
        if constexpr (SHAPE::fixed) {
            A.N = SHAPE::N; A.nO = SHAPE::nO; A.M = SHAPE::M; A.n_max = SHAPE::n_max; A.R_max = SHAPE::R_max; A.inst_off = SHAPE::inst_off;
            A.soc_lds = SHAPE::soc_lds;
        }
    }
    if (inst >= A.B) return;
#ifdef OBCA_PROFILE
    const long long prof_body_t0 = wall_clock64();   // (prologue of a pass: descriptor, instance data, start point, scaling, row initialisation -> slot 10 of the one-wavefront kernels)
#endif
    if (A.variant[inst] == 0) {                      // masked out by the caller (device-side closed loop)
        if (lane == 0) { A.status[inst] = OBCA_STATUS_SKIPPED; A.iters[inst] = 0; }
        return;
    }
    {   // anything but obca_mpc4 / 6 / 8, or obca_mpc6 without its terminal set: a per-instance error, nothing is solved
        const int v = A.variant[inst];
        if ((v != 4 && v != 6 && v != 8) || (v == 6 && A.term == nullptr)) {
            if (lane == 0 && first) { A.status[inst] = OBCA_STATUS_BAD_VARIANT; A.iters[inst] = 0; }
            return;
        }
    }
    // The caller runs the body up to OBCA_MAX_PASSES times per instance: the start ladder (same rule in oracle/ipm_dense.py:solve
    // and csrc/obca_lpi_core.h:run_instance).  first: the first start of the order (include/obca_mpc.h: start_order; default
    // x0 -> reference window -> zeros).  Every further call looks at how the previous one ended:
    //   * feasible point (or nothing to solve): return;
    //   * obca_mpc4 converged with elastic variables left at the base penalty: the l1 penalty is exact only while rho exceeds
    //     the multipliers, so this is what "infeasible" looks like but also what a too small rho looks like (the open-loop
    //     problems of demo1 at N = 10 and of demo9 at N = 50) -- the SAME start again with rho x 100, then with rho x 1000;
    //   * anything else without a feasible point (infeasible stationary point of the penalty problem, line-search failure,
    //     iteration limit, filter full): the NEXT start of the order, if there is one, at the base penalty again (measured:
    //     keeping the raised penalty makes starts fail that succeed at the base one).
    // That is at most three starts x three penalties = OBCA_MAX_PASSES solves.  A genuinely infeasible problem stays
    // infeasible.  The state of the ladder lives in LDS (it must survive from call to call, and nothing of it may occupy a
    // register during the solve).
    // After the order's starts, fixed-time problems that still have no feasible point get the dodge rung (csrc/obca_device.h:
    // OBCA_KIND_DODGE_R / _L): two more passes, BOTH run, the feasible answer with the lower objective stays in the caller's buffers.
    // Between the two a feasible first answer is marked by the transient status OBCA_STATUS_DODGE_OK / _ACC (never seen by a
    // caller: the second pass always replaces it), which the pass loops treat as "go on".
    // WHICH pass's answer stays in the caller's buffers when no start ends at a feasible point: csrc/obca_device.h: OBCA_LADDER_REPLACES.
    // The status word in HBM is therefore that of the HELD answer; how the LAST pass ended -- what the ladder's next move depends on --
    // is a bit of ladder_state[0].
    __shared__ int ladder_state[4];      // [0] bits 0-3: escalation level of the current start (0: base penalty), bit 4: the last pass converged with elastic variables left, bit 5: an answer is held, bits 8-11: 1 + the start whose converged-infeasible answer is held (0: the held one did not converge); [1] iterations, [2] factorisations so far, [3] index of the current start (nstarts, nstarts + 1: the dodge passes)
    __shared__ double ladder_f;          // objective of a feasible first dodge pass
    const int order = OBCA_EFFECTIVE_ORDER(Ain.prm.opt.order, A.variant[inst], A.warm_z != nullptr && (A.warm_use == nullptr || A.warm_use[inst] != 0), Ain.prm.opt.nstarts == 1);
    int start_s = 0, escalated = 0, held_bits = 0;
    if (!first) {
        const int st0 = __builtin_amdgcn_readfirstlane(A.status[inst]);     // wave-uniform: the flags below stay scalar
        if (st0 == OBCA_STATUS_OK || st0 == OBCA_STATUS_ACCEPTABLE || st0 < OBCA_STATUS_NUMERIC) return;
        const int l0 = __builtin_amdgcn_readfirstlane(ladder_state[0]);
        escalated = l0 & 15; held_bits = l0 & 0xf20;
        start_s = __builtin_amdgcn_readfirstlane(ladder_state[3]);
        if (A.variant[inst] == 4 && (l0 & 16) && escalated < OBCA_N_ESCALATIONS) ++escalated;
        else { escalated = 0; if (++start_s >= Ain.prm.opt.nstarts + ((Ain.prm.opt.dodge && A.variant[inst] != 4) ? OBCA_DODGE_PASSES(A.variant[inst]) : 0)) return; }
    }
    const double rho_mult = escalated ? OBCA_RHO_ESCALATION(escalated) : 1.0;
    // (dodge passes nstarts .. nstarts + 3: right, left at OBCA_RESTART_MU, then -- obca_mpc8 only, OBCA_DODGE_PASSES -- right, left at OBCA_DODGE_LEVEL2_MU)
    const int kind = start_s < Ain.prm.opt.nstarts ? OBCA_START_KIND(order, start_s) : (((start_s - Ain.prm.opt.nstarts) & 1) == 0 ? OBCA_KIND_DODGE_R : OBCA_KIND_DODGE_L);
    const bool dodge2 = start_s >= Ain.prm.opt.nstarts + 2;
    const bool from_window = kind == OBCA_KIND_WINDOW;

    // ---- layout ------------------------------------------------------------------------------------
    Lay L;
    L.N = A.N; L.nO = A.nO; L.M = A.M;
    L.variant = A.variant[inst];
    L.free_T = (L.variant == 4) ? 1 : 0;
    L.NS = 5 + L.M + 4 * L.nO;
    L.n = (L.N + 1) * (3 + L.M + 4 * L.nO) + 2 * L.N + L.free_T;
    L.npair = (L.N + 1) * L.nO;
    L.r_init = 0; L.r_dyn = 3; L.r_term = 3 + 3 * L.N;
    L.r_xb = L.r_term + (L.variant == 4 ? 3 : 0);
    L.r_ub = L.r_xb + 2 * (L.N + 1);
    L.r_acc = L.r_ub + 2 * L.N;
    L.r_T = L.r_acc + 2 * L.N;
    L.r_tx = L.r_T + (L.free_T ? 2 : 0);
    L.r_norm = L.r_tx + (L.variant == 6 ? 2 : 0);
    L.r_dist = L.r_norm + L.npair;
    L.r_lam = L.r_dist + L.npair;
    L.r_mu = L.r_lam + (L.N + 1) * L.M;
    L.R = L.r_mu + (L.N + 1) * 4 * L.nO;

    // ---- carve (sizes are the handle-wide maxima computed on the host the same way).  GM (rows in memory, RPL = 0): only what
    // the stage-serial sweep touches -- O(N) doubles -- lives in LDS, everything that grows with the number of rows or
    // variables in this instance's slice of the HBM workspace (stays in L2: ~1 MB per instance at N = 74) -----------------
    constexpr bool GM = (RPL == 0);
    Sh S;
    Rows<RPL> W;
    {
        double* p = smem;
        double* q = nullptr;
        if constexpr (GM) q = A.gm_ws + (size_t)blockIdx.x * (size_t)A.gm_stride;
        auto take = [&](int cnt) { double* r = p; p += (cnt + 1) & ~1; return r; };
        auto takeG = [&](int cnt) { if constexpr (!GM) { double* r = p; p += (cnt + 1) & ~1; return r; } else { double* r = q; q += (cnt + 1) & ~1; return r; } };
        const int nmax = A.n_max, Rmax = A.R_max, np = L.npair, N1 = L.N + 1;
        S.x = takeG(nmax); S.dx = takeG(nmax > 120 ? nmax : 120); S.gf = take(5 * N1 + 1); S.bx = takeG(nmax);
        S.y = takeG(Rmax); S.Einv = takeG(Rmax); S.gh = takeG(Rmax); S.Lb = takeG(Rmax); S.Ub = takeG(Rmax); S.dy = take(3 * N1 + 3);
        S.ct = take(N1); S.st = take(N1); S.cc = takeG(2 * np); S.ctt = take(N1); S.stt = take(N1); S.cct = takeG(2 * np);
        S.nu = takeG(2 * np); S.dnu = takeG(2 * np); S.crot = takeG(2 * np);
        S.Aobs = takeG(N1 * L.M * 2); S.bobs = takeG(N1 * L.M); S.xref = take(3 * N1);
        S.Lall = take(36 * N1); S.lall = take(8 * N1);
        {   // the trial point xt and the row staging array tmp are only alive while Y (local solutions, from the local
            // blocks to the recovery of the step) is dead, and vice versa: they share its storage
            const int nx = (nmax + 1) & ~1, nr = (Rmax + 1) & ~1, ny = MW * 4 * np;
            S.Y = takeG(ny > nx + nr ? ny : nx + nr);
            S.xt = S.Y; S.tmp = S.Y + nx;
        }
        if constexpr (!GM) {
            S.Pk = take(36 * N1 > 12 * np ? 36 * N1 : 12 * np);
            S.Sloc = S.Pk;            // 12 doubles per pair, consumed before the Riccati sweep writes Pk
        } else {
            S.Pk = take(36 * N1);
            S.Sloc = takeG(12 * np);
        }
        S.qk = take(6 * N1); S.Kk = take(12 * N1); S.kapk = take(2 * N1); S.Mik = take(9 * (N1 + 1));
        // one stage's [F G], the 8 x 8 stage matrix and its gradient are only alive during the backward sweep, when the
        // step dx (written by the forward pass after it) is not: they share its storage (GM: dx is in HBM, they get their own LDS)
        if constexpr (!GM) { S.FG = S.dx; } else { S.FG = take(120); }
        S.Mall = S.FG + 48; S.mall = S.FG + 112;
        S.lsv = take(32);
        S.offm = reinterpret_cast<int*>(take(8));
        if constexpr (!GM) {
            S.Es = S.Einv; S.gs = S.gh; S.uv = S.x; S.Tv = S.x + L.iT();
            L.uvs = L.NS; L.uvo = 3;
        } else {
            double* es = take(3 * N1 + 3); double* gs = take(3 * N1 + 3); double* uv = take(2 * N1 + 2);
            S.Es = es; S.gs = gs; S.uv = uv; S.Tv = uv + 2 * N1;
            L.uvs = 2; L.uvo = 0;
            const int ns = (Rmax + NT - 1) / NT, rs = ns * NT;
            W.lane_ = lane; W.nslots_ = (L.R + NT - 1) / NT;
#define X(f) W.f##_ = takeG(rs);
            OBCA_ROW_FIELDS(X)
#undef X
        }
    }
    Inst& in = *reinterpret_cast<Inst*>(smem + A.inst_off);
    // two-sided sweep (four-wavefront kernels only: their launches ask for OBCA_ZK_DOUBLES(N) more LDS, BEHIND everything the
    // other kernels carve, so that one layout serves all): recovery maps of the forward half, cost-to-arrive at stage 0
    S.Zk = smem + A.inst_off + OBCA_INST_DOUBLES; S.Pi0 = S.Zk + 36 * ((L.N + 1) / 2);
    if constexpr (RPL < 0) {     // fifth row slot in LDS, behind the two-sided sweep's storage (host: OBCA_HYB_DOUBLES)
        const int nl = A.R_max > 4 * NT ? A.R_max - 4 * NT : 0;
        W.m_ = S.Zk + OBCA_ZK_DOUBLES(L.N) + 15 * (lane < nl ? lane : nl);
        W.gl_ = S.Zk + OBCA_ZK_DOUBLES(L.N) + 15 * (nl + 1) + 1 + lane;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i <= OBCA_MAX_OBST; ++i) S.offm[i] = Ain.offm[i];     // static indices: A stays in kernarg
    }

    // ---- instance data ----------------------------------------------------------------------------------
    if (lane == 0) {
        const bool fr = L.free_T != 0;
        for (int j = 0; j < 9; ++j) { in.Q[j] = fr ? Ain.prm.free_time.Q[j] : Ain.prm.fixed_time.Q[j]; in.P[j] = fr ? Ain.prm.free_time.P[j] : Ain.prm.fixed_time.P[j]; }
        for (int j = 0; j < 4; ++j) { in.R1[j] = fr ? Ain.prm.free_time.R1[j] : Ain.prm.fixed_time.R1[j]; in.R2[j] = fr ? Ain.prm.free_time.R2[j] : Ain.prm.fixed_time.R2[j]; in.gego[j] = Ain.prm.gego[j]; }
        for (int j = 0; j < 2; ++j) { in.xL[j] = Ain.prm.xL[j]; in.xU[j] = Ain.prm.xU[j]; in.uL[j] = Ain.prm.uL[j]; in.uU[j] = Ain.prm.uU[j]; }
        in.off = Ain.prm.off; in.dmin = Ain.prm.dmin;
        for (int j = 0; j < 3; ++j) in.x0[j] = A.x0[(size_t)inst * 3 + j];
        for (int j = 0; j < 2; ++j) in.u0[j] = A.u0[(size_t)inst * 2 + j];
        in.Ts = A.Ts[inst];
        for (int j = 0; j < 3; ++j) in.term[j] = (L.variant == 6) ? A.term[(size_t)inst * 3 + j] : 0.0;
    }
    SYNC();
    {
        const int N1 = L.N + 1;
        const double* xr = A.xref + (size_t)inst * 3 * N1;
        for (int t = lane; t < 3 * N1; t += NT) S.xref[t] = xr[t];
        const double* Ag = A.A + (size_t)inst * N1 * L.M * 2;
        const double* bg = A.b + (size_t)inst * N1 * L.M;
        for (int t = lane; t < N1 * L.M * 2; t += NT) {
            const int k = t / (2 * L.M), q = t - k * 2 * L.M;
            S.Aobs[t] = Ag[(size_t)(L.variant == 4 ? 0 : k) * L.M * 2 + q];     // q5: mpc4 reads step 0 only
        }
        for (int t = lane; t < N1 * L.M; t += NT) {
            const int k = t / L.M, q = t - k * L.M;
            S.bobs[t] = bg[(size_t)(L.variant == 4 ? 0 : k) * L.M + q];
        }
        SYNC();
        const double dis = (S.xref[0 * N1 + L.N] - in.x0[0]) + (S.xref[1 * N1 + L.N] - in.x0[1]);   // signed sum (q3)
        if (lane == 0) {
            in.Tmax = dis / (L.N * in.uU[0] * in.Ts) + 1.0;
            // (every thread read the ladder's state before the barriers above)
            ladder_state[0] = escalated | held_bits; ladder_state[3] = start_s;
            if (first) { ladder_state[1] = 0; ladder_state[2] = 0; }
        }
    }
    SYNC();

    ObcaOptsDev O;
    O.tol = Ain.prm.opt.tol; O.rho = Ain.prm.opt.rho * rho_mult; O.feas_tol = Ain.prm.opt.feas_tol; O.max_iter_free = Ain.prm.opt.max_iter_free; O.max_iter_fixed = Ain.prm.opt.max_iter_fixed; O.max_soc = Ain.prm.opt.max_soc;          // by value: A may live in HBM (fused closed-loop kernel)
    const int max_iter_v = L.free_T ? O.max_iter_free : O.max_iter_fixed;
    const int max_iter_w = start_s >= Ain.prm.opt.nstarts ? Ain.prm.opt.retry_iter : Ain.prm.opt.nstarts == 1 ? max_iter_v : (start_s == 0 ? Ain.prm.opt.patience : Ain.prm.opt.retry_iter);   // include/obca_mpc.h: patience
    const int max_iter = max_iter_v < max_iter_w ? max_iter_v : max_iter_w;
    const double acc_tol = L.free_T ? 1e-6 : 1e-8;                 // obca.py:1538
    const double acc_objchg = L.free_T ? 1e20 : 1e-6;

    // ---- start point: zeros, Topt = 1 (obca.py:856), then by kind: every pose at x0 / the reference window / nothing -- or,
    // when the caller asked for it (obca_set_warm_start; the reference never does) and this is the order's first cold
    // start, the previous solve's primal vector moved one stage forward (last stage repeated)
    const bool warm = kind == OBCA_WARM_KIND(order) && A.warm_z != nullptr && (A.warm_use == nullptr || A.warm_use[inst] != 0);
    if (warm) {
        const double* zp = A.warm_z + (size_t)inst * A.n_max;
        const int blk = L.NS - 2;                              // pose, lambda, mu of a stage (inputs handled apart)
        for (int t = lane; t < (L.N + 1) * blk; t += NT) {
            const int k = t / blk, q = t - k * blk;
            const int ks = k < L.N ? k + 1 : L.N;
            const int dst = (q < 3 ? L.ip(k) + q : L.il(k) + (q - 3));
            const int src = (q < 3 ? L.ip(ks) + q : L.il(ks) + (q - 3));
            S.x[dst] = zp[src];
        }
        for (int t = lane; t < 2 * L.N; t += NT) {
            const int k = t >> 1, j = t & 1;
            S.x[L.iu(k) + j] = zp[L.iu(k + 1 < L.N ? k + 1 : L.N - 1) + j];
        }
        if (L.free_T && lane == 0) S.x[L.iT()] = zp[L.iT()];
    } else {
        for (int t = lane; t < L.n; t += NT) S.x[t] = 0.0;
    }
    for (int t = lane; t < 2 * L.npair; t += NT) S.nu[t] = 0.0;
    SYNC();
    if (!warm && L.free_T && lane == 0) S.x[L.iT()] = 1.0;
    if (from_window && lane == 0) window_start_point(S.x, S.xref, &in, L.N, L.NS, L.free_T ? L.iT() : -1);
    if (kind >= OBCA_KIND_DODGE_R && lane == 0) dodge_start_point(S.x, S.xref, S.Aobs, S.bobs, S.offm, &in, L.N, L.NS, L.M, L.nO, kind == OBCA_KIND_DODGE_R ? -1.0 : 1.0);
    // x0 start: where IPOPT's first full Newton step lands from the all-zero start (the dynamics linearised at v = 0 read
    // x_{k+1} = x_k, the initial condition x_0 = x0)
    if (kind == OBCA_KIND_X0 && !warm) for (int t = lane; t < 3 * (L.N + 1); t += NT) { const int k = t / 3; S.x[L.ip(k) + (t - 3 * k)] = in.x0[t - 3 * k]; }
    SYNC();
    int status = OBCA_STATUS_MAXITER;
    int it = 0, nfact = 0;
    double sf = 1.0, rho = O.rho;
    // Iteration-level scalars that the factorisation never reads are kept in LDS (S.lsv), not in registers: written by
    // thread 0, read back where they are used (a handful of broadcast reads per iteration) -- about 25 registers less
    // across the factorisation, which is what lets the second-order correction fit without scratch.
    enum { IV_E0 = 11, IV_THMAX, IV_THMIN, IV_EMAX, IV_CNTNZ, IV_CNTROWS, IV_DWLAST, IV_FPREV, IV_F, IV_TAU, IV_PHIK, IV_N };
#define PUT(idx, v) do { if (lane == 0) S.lsv[idx] = (v); } while (0)
#define GET(idx) (S.lsv[idx])
    PUT(IV_E0, INFINITY);
    bool bad_bounds = false;

    // objective scaling: IPOPT's gradient rule applied to f + rho*sum(p+n)
    for (int t = lane; t <= L.igT(); t += NT) S.gf[t] = 0.0;  // (compact: the objective does not depend on lambda, mu)
    SYNC();
    eval_geom(L, S, S.x, S.ct, S.st, S.cc, lane);
    double f0 = eval_objective<true>(L, S, in, S.x, 1.0, lane);
    {
        double gm = 0.0;
        for (int t = lane; t <= L.igT(); t += NT) gm = dmaxabs(gm, S.gf[t]);
        gm = fmax(red_max(gm), O.rho);
        sf = (gm > OBCA_MAX_GRADIENT) ? OBCA_MAX_GRADIENT / gm : 1.0;
        rho = O.rho * sf;
    }
    SYNC();
    f0 = eval_objective<true>(L, S, in, S.x, sf, lane);
    PUT(IV_F, f0);
    double mu = (from_window || (kind >= OBCA_KIND_DODGE_R && !dodge2)) ? OBCA_RESTART_MU : (warm ? A.warm_mu : (dodge2 ? OBCA_DODGE_LEVEL2_MU : OBCA_MU_INIT));
    // rows: bounds, slacks with bound push, elastic variables on their 1-d central path
    {
        int bb = 0;
        // heavy per-row code (the row-type switch) runs in rolled loops that stage through LDS; the unrolled
        // register loops below only do arithmetic -- unrolling the switch RPL times exploded register pressure
        for (int r = lane; r < L.R; r += NT) {
            double lo, up;
            row_bounds(L, in, r, lo, up);
            S.Lb[r] = lo; S.Ub[r] = up;
            S.tmp[r] = row_value(L, S, in, S.x, S.ct, S.st, S.cc, r);
        }
#pragma unroll
        for (int j = 0; j < W.slots(); ++j) {
            const int r = lane + NT * j;
            W.s(j) = 0.0; W.p(j) = 1.0; W.n(j) = 1.0;
            W.zL(j) = 0.0; W.zU(j) = 0.0; W.zp(j) = 1.0; W.zn(j) = 1.0; W.g(j) = 0.0; W.dy(j) = 0.0;
            W.iDs(j) = 0.0; W.iDp(j) = 1.0; W.iDn(j) = 1.0; W.rs(j) = 0.0; W.rp(j) = 0.0; W.rn(j) = 0.0;
            if (r < L.R) {
                const double lo = S.Lb[r], up = S.Ub[r];
                const bool eq = row_iseq(L, r);
                const double g = S.tmp[r];
                double s = g;
                const bool hasL = lo > -INFINITY, hasU = up < INFINITY;
                if (eq) s = 0.0;
                else if (hasL && hasU) {
                    if (!(lo < up)) bb = 1;
                    const double pL = fmin(OBCA_BOUND_PUSH * fmax(1.0, fabs(lo)), OBCA_BOUND_FRAC * (up - lo));
                    const double pU = fmin(OBCA_BOUND_PUSH * fmax(1.0, fabs(up)), OBCA_BOUND_FRAC * (up - lo));
                    s = fmin(fmax(s, lo + pL), up - pU);
                } else if (hasL) s = fmax(s, lo + OBCA_BOUND_PUSH * fmax(1.0, fabs(lo)));
                else if (hasU) s = fmin(s, up - OBCA_BOUND_PUSH * fmax(1.0, fabs(up)));
                const double rr = g - s;
                const double a = (mu - rho * rr) / (2.0 * rho);
                const double en = a + sqrt(a * a + mu * rr / (2.0 * rho));
                const double ep = rr + en;
                W.g(j) = g; W.s(j) = s; W.p(j) = ep; W.n(j) = en;
                W.zp(j) = mu / ep; W.zn(j) = mu / en;
                W.zL(j) = (!eq && hasL) ? 1.0 : 0.0;
                W.zU(j) = (!eq && hasU) ? 1.0 : 0.0;
                S.y[r] = rho - mu / ep;
            }
        }
        bad_bounds = red_or(bb) != 0;
        for (int pr = lane; pr < L.npair; pr += NT) {
            double e1, e2;
            rot_value(L, S.x, S.ct, S.st, S.cc, pr, e1, e2);
            S.crot[2 * pr] = e1; S.crot[2 * pr + 1] = e2;
        }
        SYNC();
    }

    // constants of the error scaling: number of bound multipliers and of rows (Topt rows count N+1 times)
    double cnt_nz = 0.0, cnt_rows = 0.0;
#pragma unroll
    for (int j = 0; j < W.slots(); ++j) {
        const int r = lane + NT * j;
        if (r < L.R) {
            const double w = row_w(L, r);
            const bool eq = row_iseq(L, r);
            cnt_nz += w * (((!eq && S.Lb[r] > -INFINITY) ? 1.0 : 0.0) + ((!eq && S.Ub[r] < INFINITY) ? 1.0 : 0.0) + 2.0);
            cnt_rows += w;
        }
    }
    cnt_nz = red_sum(cnt_nz);
    cnt_rows = red_sum(cnt_rows) + 2.0 * L.npair;
    PUT(IV_CNTNZ, cnt_nz); PUT(IV_CNTROWS, cnt_rows);

    // filter: one entry per lane
    bool f_valid = false;
    double f_th = 0.0, f_phi = 0.0;
    PUT(IV_TAU, fmax(OBCA_TAU_MIN, 1.0 - mu));
    PUT(IV_DWLAST, 0.0); PUT(IV_EMAX, 0.0); PUT(IV_PHIK, NAN);
    int acc_count = 0;
    bool have_prev = false;

#ifdef OBCA_PROFILE
    long long prof_t[20];
    for (int i = 0; i < 20; ++i) prof_t[i] = 0;
#endif
    // IPOPT's second-order correction (max_soc = 4, kappa_soc = 0.99): when the FIRST trial step of a line search is rejected
    // without reducing the constraint violation, up to max_soc corrected steps are tried -- same matrix (same delta_w),
    // right-hand side from the accumulated residuals c_soc (rotation rows, kept in S.crot) and g_soc (rows, in the scratch
    // ws_gsoc below: LDS where it fits, else HBM) -- before the step length is halved.  A corrected solve is ANOTHER PASS of this iteration loop with
    // soc_pass set (the error evaluation and the barrier update are skipped, `it` does not advance), so that the one copy
    // of the solve pipeline serves both and no second loop is wrapped around the factorisation: such a loop kept ~130
    // more registers alive in the hot path (measured: 512 B of scratch per lane, 350 MB of HBM traffic per launch).  The
    // scalars of the interrupted line search wait in LDS (S.lsv), the original direction in the same scratch; ~1 line search in
    // 2000 on the C2 workload from the x0 start, one solve in six from the window start (round 5's default).
    enum { LS_TH = 0, LS_FOBJ, LS_DW, LS_AZ, LS_DPHI, LS_PHI, LS_AMIN, LS_PWTH, LS_PWDPHI, LS_ALPHA, LS_THOLD };
    bool soc_pass = false, use_soc = false, first_trial = true;
    int soc_it = 0;
    PROF_DECL
#ifdef OBCA_PROFILE
    if (NT == 64) prof_t[10] += prof_last - prof_body_t0;
#endif
    if (bad_bounds) status = OBCA_STATUS_BAD_BOUNDS;
    else
    for (it = 0; it <= max_iter; ++it) {
        PROF(11)
        double th = 0.0, fobj = 0.0, delta_w = 0.0;
        if (!soc_pass) {
        // ---- gradient of the Lagrangian and optimality error ------------------------------------------
        gather_grad<false>(L, S, in, S.bx, lane);                    // bx doubles as scratch for grad_x L here
        double rxmax = 0.0, crotmax = 0.0, nusum = 0.0, pnsum = 0.0;
        for (int t = lane; t < L.n; t += NT) rxmax = dmaxabs(rxmax, S.bx[t]);
        for (int t = lane; t < 2 * L.npair; t += NT) { crotmax = dmaxabs(crotmax, S.crot[t]); nusum += fabs(S.nu[t]); th += fabs(S.crot[t]); }
        rxmax = red_max(rxmax); crotmax = red_max(crotmax); nusum = red_sum(nusum);
        const ErrFirst ef = ipm_errors_first<RPL>(L, S, W, mu, rho, rxmax, crotmax, nusum, th, GET(IV_CNTNZ), GET(IV_CNTROWS), lane);
        th = ef.th; pnsum = ef.pnsum; PUT(IV_EMAX, ef.emax);
        const Err e0 = ef.e0;
        const double E0 = e0.E;
        PUT(IV_E0, E0);
        if (it == 0) {
            PUT(IV_THMAX, OBCA_THETA_MAX_FACT * fmax(1.0, th));
            PUT(IV_THMIN, OBCA_THETA_MIN_FACT * fmax(1.0, th));
        }
        if (E0 <= O.tol && e0.dual <= 1.0 && e0.prim <= 1e-4 && e0.comp <= 1e-4) { status = OBCA_STATUS_OK; break; }
        fobj = GET(IV_F) + rho * pnsum;
        const double objchg = have_prev ? fabs(fobj - GET(IV_FPREV)) / fmax(1.0, fabs(fobj)) : INFINITY;
        if (E0 <= acc_tol && e0.dual <= 1e10 && e0.prim <= 1e-2 && e0.comp <= 1e-2 && objchg <= acc_objchg) {
            if (++acc_count >= OBCA_ACCEPTABLE_ITER) { status = OBCA_STATUS_ACCEPTABLE; break; }
        } else acc_count = 0;
        if (it == max_iter) break;
        PROF(0)
        // ---- barrier parameter ---------------------------------------------------------------------------
        {
            const double mu_floor = O.tol / (OBCA_KAPPA_EPS + 1.0);
            bool first_mu = true;
            while (mu > mu_floor) {
                const Err em = first_mu ? ef.em : ipm_errors<RPL>(L, S, W, mu, rho, rxmax, crotmax, nusum, lane);
                first_mu = false;
                if (em.E > OBCA_KAPPA_EPS * mu) break;
                mu = fmax(mu_floor, fmin(OBCA_KAPPA_MU * mu, mu * sqrt(mu)));      // mu^theta_mu, theta_mu = 1.5
                PUT(IV_TAU, fmax(OBCA_TAU_MIN, 1.0 - mu));
                PUT(IV_PHIK, NAN);                 // the barrier function has changed
                f_valid = false;
            }
        }
        } else {
            th = S.lsv[LS_TH]; fobj = S.lsv[LS_FOBJ]; delta_w = S.lsv[LS_DW];
            (void)eval_objective<true>(L, S, in, S.x, sf, lane);   // the trial evaluation left ITS gradient in gf
        }
        PROF(1)
        // ---- Newton step with inertia correction -----------------------------------------------------------
        // (in LDS behind everything else where the host found room for it -- csrc/obca_device.h: obca_soc_lds_wave / _mw --, else this
        // instance's slice of the HBM scratch)
#define ws_dxo (A.soc_lds ? smem + A.soc_lds : A.soc_ws + (size_t)inst * (A.n_max + 2 * A.R_max + 2 * L.npair))
#define ws_dyo (ws_dxo + A.n_max)
#define ws_gsoc (ws_dyo + A.R_max)
#define ws_dnuo (ws_gsoc + A.R_max)
        const int max_soc = (A.soc_lds || A.soc_ws) ? O.max_soc : 0;
#pragma unroll
        for (int j = 0; j < W.slots(); ++j) {       // unconditional writes end the live ranges of the last step data,
            W.dy(j) = 0.0; W.iDs(j) = 0.0; W.iDp(j) = 0.0; W.iDn(j) = 0.0; W.rs(j) = 0.0; W.rp(j) = 0.0; W.rn(j) = 0.0;
        }                                      // so they do not occupy registers across the factorisation
        bool first_try = true;
        int fail = 0;
        const double dw_last = GET(IV_DWLAST);
        for (;;) {
#pragma unroll
            for (int j = 0; j < W.slots(); ++j) {
                const int r = lane + NT * j;
                if (r < L.R) {
                    const bool eq = row_iseq(L, r);
                    const double y = S.y[r];
                    const double lo_ = S.Lb[r], up_ = S.Ub[r];
                    const Lin q = row_lin(lo_, up_, eq, W.s(j), W.p(j), W.n(j), y, W.zL(j), W.zU(j), W.zp(j),
                                          W.zn(j), mu, rho, delta_w);
                    const double rg = soc_pass ? ws_gsoc[r] : W.g(j) - (eq ? 0.0 : W.s(j)) - W.p(j) + W.n(j);
                    const double gh = rg + q.rs * q.iDs + q.rp * q.iDp - q.rn * q.iDn;
                    const double Ei = rcp64(q.iDs + q.iDp + q.iDn);
                    S.Einv[r] = Ei;
                    S.gh[r] = gh;
                }
            }
            SYNC();
            if constexpr (GM) {     // LDS mirrors of what the stage-serial sweep reads: soft rows' E^-1 and ghat, inputs, time scale
                double* es = const_cast<double*>(S.Es); double* gs = const_cast<double*>(S.gs); double* uv = const_cast<double*>(S.uv);
                for (int t = lane; t < L.r_term; t += NT) { es[t] = S.Einv[t]; gs[t] = S.gh[t]; }
                for (int t = lane; t < 2 * L.N; t += NT) uv[t] = S.x[L.iu(t >> 1) + (t & 1)];
                if (lane == 0 && L.free_T) uv[2 * (L.N + 1)] = S.x[L.iT()];
                SYNC();
            }
            gather_grad<true>(L, S, in, S.bx, lane);
            PROF(2)
            assemble_stages(L, S, in, sf, delta_w, lane);
            PROF(3)
            int bad = local_blocks(L, S, in, delta_w, lane);
            PROF(4)
#ifdef OBCA_PROFILE
            if (!bad) bad = riccati(L, S, in, lane, prof_t, A.two_sided != 0);
#if !defined(OBCA_TWO_SIDED_CHECK) && OBCA_NT >= 256
            if (A.two_sided != 0) {         // slots 18 / 19: solves whose sweep ran two-sided / all solves, and the largest E^-1 seen
                double dmax = 0.0;
                for (int r = L.r_init + (lane & 63); r < L.r_term; r += 64) dmax = fmax(dmax, S.Einv[r]);
                dmax = wave_max(dmax);
                prof_t[18] += (dmax <= OBCA_TWO_SIDED_DMAX) ? 1 : 0;
                prof_t[19] += 1;
                if (dmax <= 4e6 && dmax > 1e6) prof_t[10] += 1;
            }
#endif
#if defined(OBCA_TWO_SIDED_CHECK) && OBCA_NT >= 256
            // dev check: every two-sided solve is repeated one-sided on the same data; slot 18 keeps the largest
            // difference of the steps (poses, inputs, T; relative to the step's largest entry, x 1e18), slot 19 that of
            // the multiplier steps of the elastic rows
            if (!bad && A.two_sided != 0 && two_sided_split(L, S, lane) > 0) {
                const int nsave = 5 * L.N + 3 + 1, nrow = L.r_term;
                for (int t = lane; t < nsave + nrow; t += NT) {
                    double v;
                    if (t < nsave) { const int k = t / 5, j = t - 5 * k; v = (t == nsave - 1) ? (L.free_T ? S.dx[L.iT()] : 0.0) : (j < 3 ? S.dx[L.ip(k) + j] : S.dx[L.iu(k) + j - 3]); }
                    else v = S.dy[t - nsave];
                    S.Zk[t] = v;
                }
                SYNC();
                (void)riccati(L, S, in, lane, prof_t, false);
                double ex = 0.0, sx = 0.0, ey = 0.0, sy = 0.0;
                for (int t = lane; t < nsave + nrow; t += NT) {
                    double v;
                    if (t < nsave) { const int k = t / 5, j = t - 5 * k; v = (t == nsave - 1) ? (L.free_T ? S.dx[L.iT()] : 0.0) : (j < 3 ? S.dx[L.ip(k) + j] : S.dx[L.iu(k) + j - 3]); }
                    else v = S.dy[t - nsave];
                    if (t < nsave) { ex = fmax(ex, fabs(v - S.Zk[t])); sx = fmax(sx, fabs(v)); }
                    else { ey = fmax(ey, fabs(v - S.Zk[t])); sy = fmax(sy, fabs(v)); }
                }
                ex = red_max(ex); sx = red_max(sx); ey = red_max(ey); sy = red_max(sy);
                const long long rx = (long long)(ex / fmax(sx, 1e-300) * 1e18), ry = (long long)(ey / fmax(sy, 1e-300) * 1e18);
                if (rx > prof_t[18]) prof_t[18] = rx;
                if (ry > prof_t[19]) prof_t[19] = ry;
            }
#endif
#else
            if (!bad) bad = riccati(L, S, in, lane, A.two_sided != 0);
#endif
            PROF(5)
            ++nfact;
            if (!bad || soc_pass) break;           // a corrected solve reuses the accepted delta_w: same matrix, same pivots
            if (first_try) {
                delta_w = (dw_last == 0.0) ? OBCA_DELTA_W_0 : fmax(OBCA_DELTA_W_MIN, OBCA_KAPPA_W_MINUS * dw_last);
                first_try = false;
            } else {
                delta_w *= (dw_last == 0.0) ? OBCA_KAPPA_W_PLUS_BAR : OBCA_KAPPA_W_PLUS;
            }
            if (delta_w > OBCA_DELTA_W_MAX) { fail = 1; break; }
        }
        if (fail) { status = OBCA_STATUS_NUMERIC; break; }
        if (!soc_pass && delta_w > 0.0) PUT(IV_DWLAST, delta_w);
        // ---- row steps, step lengths, directional derivative (a corrected solve only needs its own primal step length)
        double a_max = 1.0, a_z = 1.0, dphi = 0.0, phi = 0.0;
        // phi(x_k): the line search that accepted x_k evaluated it (IPOPT keeps that value too); it is only evaluated again --
        // one log per row -- when the barrier parameter has changed since, at the first iterate, or in a corrected solve
        const double phik = GET(IV_PHIK);
        const bool phi_known = !soc_pass && isfinite(phik);
        const double tau = GET(IV_TAU);
        for (int r = lane; r < L.R; r += NT)
            S.tmp[r] = row_soft(L, r) ? S.dy[r] : (row_jdx(L, S, in, r) + S.gh[r]) * S.Einv[r];
#pragma unroll
        for (int j = 0; j < W.slots(); ++j) {
            const int r = lane + NT * j;
            if (r < L.R) {
                const bool eq = row_iseq(L, r);
                const double lo_ = S.Lb[r], up_ = S.Ub[r];
            const bool hasL = !eq && lo_ > -INFINITY, hasU = !eq && up_ < INFINITY;
                const double dy = S.tmp[r];
                W.dy(j) = dy;
                {   // cached for the line search and the update (not kept live across the factorisation)
                    const Lin q = row_lin(lo_, up_, eq, W.s(j), W.p(j), W.n(j), S.y[r], W.zL(j), W.zU(j), W.zp(j),
                                          W.zn(j), mu, rho, delta_w);
                    W.iDs(j) = q.iDs; W.iDp(j) = q.iDp; W.iDn(j) = q.iDn; W.rs(j) = q.rs; W.rp(j) = q.rp; W.rn(j) = q.rn;
                }
                const double w = row_w(L, r);
                const double ds = (dy - W.rs(j)) * W.iDs(j);
                const double dp = (dy - W.rp(j)) * W.iDp(j);
                const double dn = (-dy - W.rn(j)) * W.iDn(j);
                const double s = W.s(j), p = W.p(j), n = W.n(j);
                double gs = eq ? 0.0 : W.rs(j) + S.y[r];
                const double ids = rcp64(ds), ip = rcp64(p), inn = rcp64(n);     // (ids only used under ds != 0)
                if (hasL) {
                    const double sl = s - lo_, zL = W.zL(j);
                    if (ds < 0.0) a_max = fmin(a_max, -tau * sl * ids);
                    const double dz = (mu - zL * ds) * rcp64(sl) - zL;
                    if (dz < 0.0) a_z = fmin(a_z, -tau * zL * rcp64(dz));
                }
                if (hasU) {
                    const double su = up_ - s, zU = W.zU(j);
                    if (ds > 0.0) a_max = fmin(a_max, tau * su * ids);
                    const double dz = (mu + zU * ds) * rcp64(su) - zU;
                    if (dz < 0.0) a_z = fmin(a_z, -tau * zU * rcp64(dz));
                }
                if (dp < 0.0) a_max = fmin(a_max, -tau * p * rcp64(dp));
                if (dn < 0.0) a_max = fmin(a_max, -tau * n * rcp64(dn));
                const double zp = W.zp(j), zn = W.zn(j);
                const double dzp = (mu - zp * dp) * ip - zp, dzn = (mu - zn * dn) * inn - zn;
                if (dzp < 0.0) a_z = fmin(a_z, -tau * zp * rcp64(dzp));
                if (dzn < 0.0) a_z = fmin(a_z, -tau * zn * rcp64(dzn));
                if (!phi_known) phi += w * row_barrier(lo_, up_, eq, s, p, n, mu, rho);
                dphi += w * (gs * ds + (rho - mu * ip) * dp + (rho - mu * inn) * dn);
            }
        }
        for (int t = lane; t < L.n; t += NT) {                       // same lane <-> entry assignment as a dense gradient
            const int k = t / L.NS, q = t - k * L.NS;
            if (L.free_T && t == L.iT()) dphi += S.gf[L.igT()] * S.dx[t];
            else if (q < (k < L.N ? 5 : 3)) dphi += S.gf[L.ig(k) + q] * S.dx[t];
        }
        a_max = red_min(a_max); a_z = red_min(a_z); dphi = red_sum(dphi); phi = phi_known ? phik : red_sum(phi) + GET(IV_F);
        double alpha_min;
        // the two powers of the switching condition do not depend on the step length: once per iteration, not per trial
        double pw_th = 0.0, pw_dphi = 1.0;
        double alpha = a_max, a_try = a_max, th_old = 0.0;
        if (!soc_pass) {
            if (dphi < 0.0) {
                pw_th = dpow(th, OBCA_S_THETA);
                pw_dphi = dpow(-dphi, OBCA_S_PHI);
                double c = fmin(OBCA_GAMMA_THETA, OBCA_GAMMA_PHI * th / (-dphi));
                if (th <= GET(IV_THMIN)) c = fmin(c, OBCA_DELTA * pw_th / pw_dphi);
                alpha_min = OBCA_GAMMA_ALPHA * c;
            } else alpha_min = OBCA_GAMMA_ALPHA * OBCA_GAMMA_THETA;
            first_trial = true; use_soc = false; soc_it = 0;
        } else {                                   // corrected direction: its own step length, everything else as parked
            a_try = a_max;
            a_z = S.lsv[LS_AZ]; dphi = S.lsv[LS_DPHI]; phi = S.lsv[LS_PHI]; alpha_min = S.lsv[LS_AMIN];
            pw_th = S.lsv[LS_PWTH]; pw_dphi = S.lsv[LS_PWDPHI]; alpha = S.lsv[LS_ALPHA]; th_old = S.lsv[LS_THOLD];
        }
        soc_pass = false;
        PROF(6)
        // ---- backtracking filter line search ---------------------------------------------------------------
        double f_t = 0.0, phi_acc = NAN;
        bool accepted = false, aug = false;
        for (;;) {
            for (int t = lane; t < L.n; t += NT) S.xt[t] = S.x[t] + a_try * S.dx[t];
            SYNC();
            eval_geom(L, S, S.xt, S.ctt, S.stt, S.cct, lane);
            f_t = eval_objective<true>(L, S, in, S.xt, sf, lane);     // gradient too: gf of the current point is spent
            double th_t = 0.0, phi_t = 0.0;
            for (int r = lane; r < L.R; r += NT) S.tmp[r] = row_value(L, S, in, S.xt, S.ctt, S.stt, S.cct, r);
#pragma unroll
            for (int j = 0; j < W.slots(); ++j) {
                const int r = lane + NT * j;
                if (r < L.R) {
                    const bool eq = row_iseq(L, r);
                    const double dy = W.dy(j), w = row_w(L, r);
                    const double lo_ = S.Lb[r], up_ = S.Ub[r];
                    const double st = eq ? 0.0 : W.s(j) + a_try * (dy - W.rs(j)) * W.iDs(j);
                    const double pt = W.p(j) + a_try * (dy - W.rp(j)) * W.iDp(j);
                    const double nt = W.n(j) + a_try * (-dy - W.rn(j)) * W.iDn(j);
                    const double gt = S.tmp[r];
                    th_t += w * fabs(gt - st - pt + nt);
                    phi_t += w * row_barrier(lo_, up_, eq, st, pt, nt, mu, rho);
                }
            }
            for (int pr = lane; pr < L.npair; pr += NT) {
                double e1, e2;
                rot_value(L, S.xt, S.ctt, S.stt, S.cct, pr, e1, e2);
                th_t += fabs(e1) + fabs(e2);
                S.bx[2 * pr] = e1; S.bx[2 * pr + 1] = e2;          // bx is free during the line search: the trial's
            }                                                      // rotation residuals (the accepted ones become crot)
            th_t = red_sum(th_t);
            phi_t = red_sum(phi_t) + f_t;
            bool ok = false;
            aug = false;
            const bool finite = isfinite(phi_t) && isfinite(th_t);
#if OBCA_NT == 64
            const bool blocked = (th_t >= GET(IV_THMAX)) || (__any(f_valid && th_t >= f_th && phi_t >= f_phi) != 0);
#else               // one filter entry per thread (NT entries)
            const bool blocked = (th_t >= GET(IV_THMAX)) || red_or(f_valid && th_t >= f_th && phi_t >= f_phi);
#endif
            if (finite && !blocked) {
                const bool switching = dphi < 0.0 && alpha * pw_dphi > OBCA_DELTA * pw_th;
                if (th <= GET(IV_THMIN) && switching) {
                    ok = phi_t <= phi + OBCA_ETA_PHI * alpha * dphi + 10.0 * 2.220446049250313e-16 * fabs(phi);
                } else {
                    ok = (th_t <= (1.0 - OBCA_GAMMA_THETA) * th) || (phi_t <= phi - OBCA_GAMMA_PHI * th);
                    aug = ok;
                }
            }
            if (ok) { accepted = true; phi_acc = phi_t; break; }
            // ---- second-order correction bookkeeping (rare)
            const bool fin_eval = isfinite(th_t) && isfinite(f_t);
            bool soc_start = false, soc_next = false;
            if (use_soc) {
                if (!fin_eval || th_t > OBCA_KAPPA_SOC * th_old || soc_it >= max_soc) {
                    for (int t = lane; t < L.n; t += NT) S.dx[t] = ws_dxo[t];           // back to the original direction
                    for (int t = lane; t < 2 * L.npair; t += NT) S.dnu[t] = ws_dnuo[t];
#pragma unroll
                    for (int j = 0; j < W.slots(); ++j) {
                        const int r = lane + NT * j;
                        if (r < L.R) W.dy(j) = ws_dyo[r];
                    }
                    use_soc = false;
                    SYNC();
                } else soc_next = true;
            } else if (first_trial && max_soc > 0 && fin_eval && th_t >= th) {
                soc_start = true;
            }
            first_trial = false;
            if (soc_start || soc_next) {
                const double cf = soc_start ? alpha : a_try;
#pragma unroll
                for (int j = 0; j < W.slots(); ++j) {
                    const int r = lane + NT * j;
                    if (r < L.R) {
                        const bool eq = row_iseq(L, r);
                        const double dy = W.dy(j);
                        const double st = eq ? 0.0 : W.s(j) + a_try * (dy - W.rs(j)) * W.iDs(j);
                        const double pt = W.p(j) + a_try * (dy - W.rp(j)) * W.iDp(j);
                        const double nt = W.n(j) + a_try * (-dy - W.rn(j)) * W.iDn(j);
                        const double res = S.tmp[r] - st - pt + nt;
                        const double old = soc_start ? W.g(j) - (eq ? 0.0 : W.s(j)) - W.p(j) + W.n(j) : ws_gsoc[r];
                        ws_gsoc[r] = cf * old + res;
                        if (soc_start) ws_dyo[r] = dy;
                    }
                }
                if (soc_start) {
                    for (int t = lane; t < L.n; t += NT) ws_dxo[t] = S.dx[t];
                    for (int t = lane; t < 2 * L.npair; t += NT) ws_dnuo[t] = S.dnu[t];
                }
                for (int t = lane; t < 2 * L.npair; t += NT) S.crot[t] = cf * S.crot[t] + S.bx[t];
                if (lane == 0) {
                    S.lsv[LS_TH] = th; S.lsv[LS_FOBJ] = fobj; S.lsv[LS_DW] = delta_w; S.lsv[LS_AZ] = a_z; S.lsv[LS_DPHI] = dphi;
                    S.lsv[LS_PHI] = phi; S.lsv[LS_AMIN] = alpha_min; S.lsv[LS_PWTH] = pw_th; S.lsv[LS_PWDPHI] = pw_dphi;
                    S.lsv[LS_ALPHA] = alpha; S.lsv[LS_THOLD] = th_t;
                }
                use_soc = true;
                soc_pass = true;
                ++soc_it;
                __threadfence_block();
                SYNC();
                break;                              // another pass of the iteration loop with the corrected right-hand side
            }
            alpha *= 0.5;
            a_try = alpha;
            if (alpha < alpha_min) break;
        }
        if (soc_pass) { --it; continue; }
        PROF(7)
        if (!accepted) { status = OBCA_STATUS_LINESEARCH; break; }
        if (aug) {
            const double tn = (1.0 - OBCA_GAMMA_THETA) * th, pn = phi - OBCA_GAMMA_PHI * th;
#if OBCA_NT == 64
            if (f_valid && f_th >= tn && f_phi >= pn) f_valid = false;          // dominated entries leave
            const unsigned long long freem = __ballot(!f_valid);
            if (freem == 0ull) { status = OBCA_STATUS_NUMERIC; break; }
            const int slot = __ffsll((long long)freem) - 1;
            if (lane == slot) { f_valid = true; f_th = tn; f_phi = pn; }
#else
            if (f_valid && f_th >= tn && f_phi >= pn) f_valid = false;          // dominated entries leave
            const unsigned long long freem = __ballot(!f_valid && lane < OBCA_FILTER_CAP(A.R_max));   // per wavefront
            const double cand = freem ? (double)((lane & ~63) + __ffsll((long long)freem) - 1) : 1e9;
            const double slot = red_min(cand);                                  // first free entry of the block
            const int full = slot > 1e8 ? 1 : 0;
            if (!full && lane == (int)slot) { f_valid = true; f_th = tn; f_phi = pn; }
            if (full) { status = OBCA_STATUS_NUMERIC; break; }
#endif
        }
        // ---- accept ------------------------------------------------------------------------------------------
        // bound multipliers follow the ORIGINAL direction (step a_z); primal variables, y and nu the accepted one
#pragma unroll
        for (int j = 0; j < W.slots(); ++j) {
            const int r = lane + NT * j;
            if (r < L.R) {
                const bool eq = row_iseq(L, r);
                const double lo_ = S.Lb[r], up_ = S.Ub[r];
            const bool hasL = !eq && lo_ > -INFINITY, hasU = !eq && up_ < INFINITY;
                const double dy = W.dy(j);
                const double dyz = use_soc ? ws_dyo[r] : dy;
                const double ds = (dy - W.rs(j)) * W.iDs(j);
                const double dp = (dy - W.rp(j)) * W.iDp(j);
                const double dn = (-dy - W.rn(j)) * W.iDn(j);
                const double dsz = (dyz - W.rs(j)) * W.iDs(j);
                const double dpz = (dyz - W.rp(j)) * W.iDp(j);
                const double dnz = (-dyz - W.rn(j)) * W.iDn(j);
                const double lo = lo_, up = up_;
                const double s_old = W.s(j), p_old = W.p(j), n_old = W.n(j);
                const double s = eq ? 0.0 : s_old + a_try * ds, p = p_old + a_try * dp, n = n_old + a_try * dn;
                const double ks = OBCA_KAPPA_SIGMA, iks = 1.0 / OBCA_KAPPA_SIGMA;
                if (hasL) {
                    const double zL = W.zL(j) + a_z * ((mu - W.zL(j) * dsz) * rcp64(s_old - lo) - W.zL(j));
                    const double msl = mu * rcp64(s - lo);
                    W.zL(j) = fmax(fmin(zL, ks * msl), msl * iks);
                }
                if (hasU) {
                    const double zU = W.zU(j) + a_z * ((mu + W.zU(j) * dsz) * rcp64(up - s_old) - W.zU(j));
                    const double msu = mu * rcp64(up - s);
                    W.zU(j) = fmax(fmin(zU, ks * msu), msu * iks);
                }
                const double zp = W.zp(j) + a_z * ((mu - W.zp(j) * dpz) * rcp64(p_old) - W.zp(j));
                const double zn = W.zn(j) + a_z * ((mu - W.zn(j) * dnz) * rcp64(n_old) - W.zn(j));
                const double mp = mu * rcp64(p), mn = mu * rcp64(n);
                W.zp(j) = fmax(fmin(zp, ks * mp), mp * iks);
                W.zn(j) = fmax(fmin(zn, ks * mn), mn * iks);
                W.s(j) = s; W.p(j) = p; W.n(j) = n;
                S.y[r] += a_try * dy;
            }
        }
        for (int t = lane; t < 2 * L.npair; t += NT) { S.nu[t] += a_try * S.dnu[t]; S.crot[t] = S.bx[t]; }
        for (int t = lane; t < L.n; t += NT) S.x[t] = S.xt[t];
        SYNC();
#undef ws_dxo
#undef ws_dyo
#undef ws_gsoc
#undef ws_dnuo
        PUT(IV_FPREV, fobj);
        have_prev = true;
        PROF(8)
        // ---- the new iterate IS the accepted trial point: its geometry (ctt/stt/cct), row values (tmp) and rotation
        // residuals (crot) were evaluated by the line search; only the objective gradient is new ------------------
        for (int k = lane; k <= L.N; k += NT) { S.ct[k] = S.ctt[k]; S.st[k] = S.stt[k]; }
        for (int t = lane; t < 2 * L.npair; t += NT) S.cc[t] = S.cct[t];
#pragma unroll
        for (int j = 0; j < W.slots(); ++j) {
            const int r = lane + NT * j;
            if (r < L.R) W.g(j) = S.tmp[r];
        }
        PUT(IV_F, f_t);           // objective and its gradient (gf) came with the accepted trial as well
        PUT(IV_PHIK, phi_acc);    // and so did the barrier function's value there (valid until the barrier parameter changes)
        SYNC();
        PROF(9)
    }
#ifdef OBCA_PROFILE
    if (A.prof && lane == 0) for (int i = 0; i < 20; ++i) A.prof[(size_t)inst * 20 + i] = (double)prof_t[i];
#endif

    const int l0 = __builtin_amdgcn_readfirstlane(ladder_state[0]);      // (read by every wavefront BEFORE the barrier, rewritten by thread 0 after it)
    SYNC();
    if ((status == OBCA_STATUS_OK || status == OBCA_STATUS_ACCEPTABLE) && GET(IV_EMAX) > O.feas_tol)
        status = OBCA_STATUS_INFEASIBLE;

    if (FROM_MEMORY) {
        // the output side of the descriptor is read again here instead of being carried through the solve in registers
        __asm__ volatile("" ::: "memory");
        constexpr bool U = FROM_MEMORY;
        if constexpr (!SHAPE::fixed) { A.n_max = uni<U>(Ain.n_max); A.R_max = uni<U>(Ain.R_max); }
        A.xopt = uni<U>(Ain.xopt); A.uopt = uni<U>(Ain.uopt); A.ts_opt = uni<U>(Ain.ts_opt); A.status = uni<U>(Ain.status);
        A.iters = uni<U>(Ain.iters); A.info = uni<U>(Ain.info); A.warm_z = uni<U>(Ain.warm_z);
        A.cert_z = uni<U>(Ain.cert_z); A.cert_y = uni<U>(Ain.cert_y);
    }
    // ---- dodge passes (ladder_state[3] >= nstarts): the answer in the caller's buffers is replaced only by a better one.
    // keep: this pass's answer goes out.  status_out: what the status word becomes (-100: left as it is).
    bool keep = true;
    int status_out = status;
    {
        const int ls = __builtin_amdgcn_readfirstlane(ladder_state[3]), nst = Ain.prm.opt.nstarts;
        if (ls < nst) {                // a start of the order: OBCA_LADDER_REPLACES
            const int held_code = (l0 >> 8) & 15;
            keep = OBCA_LADDER_REPLACES(status, ls + 1, l0 & 32, held_code ? OBCA_STATUS_INFEASIBLE : OBCA_STATUS_MAXITER, held_code);
            if (!keep) status_out = -100;
            if (lane == 0) ladder_state[0] = (l0 & 15) | (status == OBCA_STATUS_INFEASIBLE ? 16 : 0) | 32 | ((keep ? (status == OBCA_STATUS_INFEASIBLE ? ls + 1 : 0) : held_code) << 8);
        } else {
            const bool ok = status == OBCA_STATUS_OK || status == OBCA_STATUS_ACCEPTABLE;
            if (((ls - nst) & 1) == 0) {   // to the right (either level): goes out if feasible, marked "one more pass to come"
                keep = ok;
                status_out = ok ? (status == OBCA_STATUS_OK ? OBCA_STATUS_DODGE_OK : OBCA_STATUS_DODGE_ACC) : -100;
                if (ok && lane == 0) ladder_f = GET(IV_F) / sf;
            } else {                   // to the left: goes out if feasible and better than a feasible first one
                const int stp = __builtin_amdgcn_readfirstlane(A.status[inst]);
                const bool pend = stp == OBCA_STATUS_DODGE_OK || stp == OBCA_STATUS_DODGE_ACC;
                keep = ok && (!pend || GET(IV_F) / sf < ladder_f);
                status_out = keep ? status : (pend ? (stp == OBCA_STATUS_DODGE_OK ? OBCA_STATUS_OK : OBCA_STATUS_ACCEPTABLE) : -100);
            }
        }
    }
    if (keep && A.warm_z != nullptr && (status == OBCA_STATUS_OK || status == OBCA_STATUS_ACCEPTABLE)) {
        double* zp = A.warm_z + (size_t)inst * A.n_max;        // kept for the next solve of this instance
        for (int t = lane; t < L.n; t += NT) zp[t] = S.x[t];
    }
    // ---- certificate output (obca_set_certificate_buffers): primal vector and multipliers in objective units
    if (!keep) { A.cert_z = nullptr; A.cert_y = nullptr; }
    if (A.cert_z != nullptr) {
        double* zc = A.cert_z + (size_t)inst * A.n_max;
        for (int t = lane; t < L.n; t += NT) zc[t] = S.x[t];
    }
    if (A.cert_y != nullptr) {
        double* yc = A.cert_y + (size_t)inst * (A.R_max + 2 * L.npair);
        const double isf = 1.0 / sf;
        for (int r = lane; r < L.R; r += NT) yc[r] = S.y[r] * isf;
        for (int t = lane; t < 2 * L.npair; t += NT) yc[L.R + t] = S.nu[t] * isf;
    }
    // ---- outputs (last iterate on failure, like the reference's except-branch) --------------------------------
    {
        const int N1 = L.N + 1;
        double* xo = A.xopt + (size_t)inst * 3 * N1;
        double* uo = A.uopt + (size_t)inst * 2 * L.N;
        if (keep) {
            for (int t = lane; t < 3 * N1; t += NT) { const int j = t / N1, k = t - j * N1; xo[t] = S.x[L.ip(k) + j]; }
            for (int t = lane; t < 2 * L.N; t += NT) { const int j = t / L.N, k = t - j * L.N; uo[t] = S.x[L.iu(k) + j]; }
        }
        if (lane == 0) {
            if (keep) A.ts_opt[inst] = L.free_T ? S.x[L.iT()] * in.Ts : in.Ts;
            if (status_out != -100) A.status[inst] = status_out;
            // (counts of the whole sequence of passes; accumulated in LDS so that the pass number is dead after the prologue)
            const int itot = it + ladder_state[1], ftot = nfact + ladder_state[2];
            ladder_state[1] = itot; ladder_state[2] = ftot;
            A.iters[inst] = itot;
            if (A.info) {
                double* io = A.info + (size_t)inst * 4;
                if (keep) { io[0] = GET(IV_F) / sf; io[1] = GET(IV_EMAX); io[2] = GET(IV_E0); }
                io[3] = (double)ftot;
            }
        }
    }
}

// The closed-form terminal-set screen of obca_mpc6 (csrc/obca_device.h: obca_terminal_shortfall; rule: oracle/ipm_dense.py:
// terminal_set_shortfall), run by the pass drivers BEFORE the first pass of an instance: a call whose terminal set no trajectory can
// reach is answered here -- status 'infeasible', zero iterations, the x0 start as the iterate -- and the solver body is not
// entered.  Out of line and outside the body on purpose: inside the body's prologue the same forty lines cost the headline kernel
// 85 more scratch loads (the allocator's choices for the whole solve changed).  // Returns 1: screened out; 2: not screened, but with less than OBCA_DODGE_MIN_SPARE of reach beyond the terminal set -- the ladder's
// dodge rung is not tried (the drivers stop after the order's starts); 0: nothing to say.
__device__ __noinline__ int terminal_screen_dev(const ObcaLaunch* a, int inst, int lane, int nt) {
    if (inst >= a->B || a->term == nullptr || a->variant[inst] != 6) return 0;
    const double* x0s = a->x0 + (size_t)inst * 3;
    const double ts = a->Ts[inst];
    const double sh = obca_terminal_shortfall(a->N, ts, x0s[0], cos(x0s[2]), a->u0[(size_t)inst * 2], a->prm.uL[0], a->prm.uU[0], a->prm.xU[0],
                                              a->term[(size_t)inst * 3], a->prm.opt.feas_tol);
    if (!(sh > 0.0 && a->prm.opt.screen)) return (a->prm.opt.dodge && !(sh < -OBCA_DODGE_MIN_SPARE)) ? 2 : 0;
    const int N1 = a->N + 1;
    double* xo = a->xopt + (size_t)inst * 3 * N1;
    double* uo = a->uopt + (size_t)inst * 2 * a->N;
    for (int t = lane; t < 3 * N1; t += nt) xo[t] = x0s[t / N1];
    for (int t = lane; t < 2 * a->N; t += nt) uo[t] = 0.0;
    if (a->cert_z != nullptr) {
        double* zc = a->cert_z + (size_t)inst * a->n_max;
        const int NSs = 5 + a->M + 4 * a->nO, nn = N1 * (3 + a->M + 4 * a->nO) + 2 * a->N;
        for (int t = lane; t < nn; t += nt) { const int k = t / NSs, q = t - k * NSs; zc[t] = q < 3 ? x0s[q] : 0.0; }
    }
    if (a->cert_y != nullptr) {
        double* yc = a->cert_y + (size_t)inst * (a->R_max + 2 * N1 * a->nO);
        const int Rv = 3 + 3 * a->N + 2 * N1 + 4 * a->N + 2 + N1 * (2 * a->nO + a->M + 4 * a->nO);      // rows of obca_mpc6
        for (int t = lane; t < Rv + 2 * N1 * a->nO; t += nt) yc[t] = 0.0;
    }
    if (lane == 0) {
        a->ts_opt[inst] = ts; a->status[inst] = OBCA_STATUS_INFEASIBLE; a->iters[inst] = 0;
        if (a->info) { double* io = a->info + (size_t)inst * 4; io[0] = 0.0; io[1] = sh; io[2] = 0.0; io[3] = 0.0; }
    }
    return 1;
}

// rows per lane: 4 covers R <= 256 (N=5/6 with 3 obstacles), 6 covers R <= 384 (demo9, 4-5 obstacles)
// The solve and, for the instances that need them, the further passes of the start ladder (penalty escalation, next starts).
// Two forms, chosen per kernel by measurement (tools/build_variant.sh, tools/gpu_variant_bench.py, tools/kernel_resources.py):
//   LOOP   ONE copy of the body in a loop over the passes.  One-wavefront kernels: one double of the hot loop is spilled
//          (16 B of scratch) and the launch is still 3 % SHORTER than with the straight form (C2: 29.8 against 30.7 ms, M = 12:
//          33.3 against 34.3 ms) at a sixth of the code.
//   !LOOP  OBCA_MAX_PASSES inlined copies of the body -- the hot one and cold ones that return at once where nothing is left to
//          do.  Every copy reads its OWN descriptor (A2, A3 = further kernel arguments with the same content): were they all to
//          read A, the compiler would merge their identical prologue expressions and keep those values alive across the whole
//          first solve.  Four-wavefront kernels: 0 / 48 B of scratch instead of 96 / 208 B, C3 free-time 60.1 against 61.8 ms,
//          gated 300 against 321 ms per 2048 solves.
// (Measured and rejected: the hot copy straight + a loop over ONE cold copy, or the cold passes in an out-of-line function -- the
// hot copy then needs 48-240 B of scratch, or the callee 1.9 KB for its callee-saved registers.)
template <int RPL, bool LOOP, class DESC, class SHAPE = ShapeAny>
__device__ __forceinline__ void solve_passes(DESC& A, DESC& A2, DESC& A3) {
    const int inst = blockIdx.x;
    // (every kernel's first argument is the launch descriptor: its address is the kernarg segment's)
    __shared__ int drv_scr;           // what the screen said (kept in LDS: nothing of it may occupy a register during the solves)
    {
        const int scr = terminal_screen_dev((const ObcaLaunch*)__builtin_amdgcn_kernarg_segment_ptr(), inst, threadIdx.x, blockDim.x);
        if (scr == 1) return;
        if (threadIdx.x == 0) drv_scr = scr;
    }
    if constexpr (LOOP) {
        if (inst >= A.B) return;
#pragma clang loop unroll(disable)
        for (int pass = 0; pass < OBCA_MAX_PASSES; ++pass) {
            obca_ipm_body<RPL, false, DESC, SHAPE>(A, inst, pass == 0);
            __syncthreads();                                        // status written by thread 0 of this workgroup
            const int st = A.status[inst];
            if (st == OBCA_STATUS_OK || st == OBCA_STATUS_ACCEPTABLE || st < OBCA_STATUS_NUMERIC) return;   // nothing (more) to recover
            if (drv_scr == 2 && pass + 1 == A.prm.opt.nstarts) return;       // obca_mpc6 without room to dodge: the order's starts were all
        }
    } else {
        static_assert(OBCA_MAX_PASSES == 9, "one inlined copy of the body per pass");
        obca_ipm_body<RPL, false, DESC, SHAPE>(A, inst, true);
        if (inst >= A.B) return;
        __syncthreads();                                            // status written by thread 0 of this workgroup
        {
            const int st = A2.status[inst];
            if (st == OBCA_STATUS_OK || st == OBCA_STATUS_ACCEPTABLE || st < OBCA_STATUS_NUMERIC) return;   // nothing to recover
        }
        if (drv_scr == 2 && A2.prm.opt.nstarts == 1) return;        // obca_mpc6 without room to dodge: the order's starts were all
        obca_ipm_body<RPL, false, DESC, SHAPE>(A2, inst, false);
        __syncthreads();
        obca_ipm_body<RPL, false, DESC, SHAPE>(A3, inst, false);
        __syncthreads();
        if (drv_scr == 2 && A3.prm.opt.nstarts == 3) return;
        obca_ipm_body<RPL, false, DESC, SHAPE>(A2, inst, false);
        __syncthreads();
        obca_ipm_body<RPL, false, DESC, SHAPE>(A3, inst, false);
        __syncthreads();
        obca_ipm_body<RPL, false, DESC, SHAPE>(A2, inst, false);
        __syncthreads();
        obca_ipm_body<RPL, false, DESC, SHAPE>(A3, inst, false);
        __syncthreads();
        obca_ipm_body<RPL, false, DESC, SHAPE>(A2, inst, false);
        __syncthreads();
        obca_ipm_body<RPL, false, DESC, SHAPE>(A3, inst, false);
    }
}
// KARG: the descriptors are read through the kernarg segment pointer (constant address space) instead of the by-value
// parameters.  Which form leaves the allocator more room differs from kernel to kernel (tools/kernel_resources.py).
template <int RPL, bool LOOP, bool KARG = false, class SHAPE = ShapeAny>
__device__ __forceinline__ void solve_with_escalation(const ObcaLaunch& A, const ObcaLaunch& A2, const ObcaLaunch& A3) {
    if constexpr (KARG) {
        ObcaLaunchConst* Ap = (ObcaLaunchConst*)__builtin_amdgcn_kernarg_segment_ptr();
        solve_passes<RPL, LOOP, ObcaLaunchConst, SHAPE>(Ap[0], Ap[1], Ap[2]);
    } else {
        solve_passes<RPL, LOOP, const ObcaLaunch, SHAPE>(A, A2, A3);
    }
}

#ifndef OBCA_LOOP_R4          /* form of solve_passes per kernel (see above); overridable for A/B builds (tools/build_variant.sh) */
#define OBCA_LOOP_R4 true
#endif
#ifndef OBCA_LOOP_R56
#define OBCA_LOOP_R56 true
#endif
#ifndef OBCA_LOOP_MW
#define OBCA_LOOP_MW false
#endif
#ifndef OBCA_LOOP_GM
#define OBCA_LOOP_GM true
#endif
#if OBCA_NT == 64 && defined(OBCA_TU_SHAPE)
// A translation unit of compile-time-shape instantiations (csrc/obca_kernel_s*.hip define OBCA_TU_SHAPE(X) as X(N, nO, M) ...
// and include this file): only those kernels, none of the generic ones.  obca_ipm_kernel_s<N>_<nO>_<M>.
#ifndef OBCA_SHAPE_KERNEL_ATTR      /* dev knob (tools/build_variant.sh): e.g. __attribute__((amdgpu_waves_per_eu(2, 2))) for a 256-register build */
#define OBCA_SHAPE_KERNEL_ATTR
#endif
#define OBCA_DEFINE_SHAPE_KERNEL(N_, O_, M_)                                                                                   \
    extern "C" __global__ void __launch_bounds__(64) OBCA_SHAPE_KERNEL_ATTR obca_ipm_kernel_s##N_##_##O_##_##M_(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { \
        using SH = ShapeIs<N_, O_, M_>;                                                                                        \
        solve_with_escalation<SH::RPL, true, true, SH>(A, A2, A3);                                                            \
    }
OBCA_TU_SHAPE(OBCA_DEFINE_SHAPE_KERNEL)
#elif OBCA_NT == 64
extern "C" __global__ void __launch_bounds__(64) obca_ipm_kernel_r4(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { solve_with_escalation<4, OBCA_LOOP_R4>(A, A2, A3); }
extern "C" __global__ void __launch_bounds__(64) obca_ipm_kernel_r5(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { solve_with_escalation<5, OBCA_LOOP_R56>(A, A2, A3); }
extern "C" __global__ void __launch_bounds__(64) obca_ipm_kernel_r6(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { solve_with_escalation<6, OBCA_LOOP_R56>(A, A2, A3); }
// rows in an HBM workspace like obca_ipm_kernel_gm, but ONE wavefront per instance (mode 5): up to four instances per CU where the
// O(N) blocks of the sweep leave the LDS room, instead of one
extern "C" __global__ void __launch_bounds__(64) obca_ipm_kernel_gm1(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { solve_with_escalation<0, OBCA_LOOP_GM>(A, A2, A3); }

// ================================================================== fused closed loop
// One wavefront runs one step of one rollout at a time: lane 0 runs the harness of csrc/obca_rollout_core.h around
// the solves, the wave runs the solves (obca_mpc4, or obca_mpc6 and, where that fails, obca_mpc8).  Nothing is
// shared between rollouts, so there is no lock step: a rollout that meets an expensive solve (an infeasible
// obca_mpc6 runs to max_iter before the fallback) does not hold the others back.  `launches[g + a*MAX_GROUPS]`
// is the descriptor obca_solve_batch would use for problem shape g (= sensed moving obstacles), a = 1 the retry.
#include "obca_rollout_core.h"

// The harness runs once per step on lane 0; kept OUT of line so that nothing of it (the ~100 pointers of rollout::Dev,
// its vertex arrays) is hoisted out of the step loop and kept alive in registers across the solve.
__device__ __noinline__ void ro_prepare(const rollout::Dev* D, int b) { rollout::prepare(*D, b); }
__device__ __noinline__ void ro_finish(const rollout::Dev* D, int b) { rollout::finish(*D, b); }
__device__ __noinline__ int ro_screen(const ObcaLaunch* L, int b) { return terminal_screen_dev(L, b, 0, 1); }
__device__ __noinline__ int ro_retry(const rollout::Dev* D, int g, int b) { rollout::make_retry(*D, g, b); return D->var8[g][b]; }
// (flag in the low half, group in the high half: an out-parameter would be a stack slot, i.e. scratch)
__device__ __noinline__ long long ro_flag_sel(const rollout::Dev* D, int b) { return ((long long)D->sel[b] << 32) | (unsigned)D->flags[b]; }


// Scheduling.  A rollout used to be one workgroup's job for its whole life (grid = B): with 4096 rollouts of very different
// cost on 1024 SIMDs the launch ended 36 % after the ideal sum / slots (tools/gpu_tail.py).  Now the unit of work is ONE
// ROUND (OBCA_RO_BLOCK consecutive steps) of one rollout: persistent workgroups (one per SIMD) claim items from a counter, in
// order -- so every rollout has done round r before any starts round r + 1 -- and the item's workgroup first waits until the
// rollout's previous round is published (it was claimed a queue length earlier: practically always long over).
//   qmode 2 (default)  ONE QUEUE PER XCD: rollout b belongs to queue b % 8 and is only ever touched by workgroups that run on
//           XCD b % 8 (each workgroup asks the hardware where it runs: HW_REG_XCC_ID).  All CUs of an XCD share its L2, so the
//           rollout's state never has to leave it: the publishing wave drains its stores (vmcnt(0): they are in L2) and sets
//           the flag, the claiming workgroup invalidates its L1 (agent-scope acquire) -- NO L2 write-back.  With the global
//           queue below every hand-off wrote back its XCD's dirty lines, mostly other workgroups' live spill slots: 15.7 GB
//           of HBM writes per C5 launch for 0.25 GB of algorithmic traffic (profiles/r03_pmc_c5_WRITE_SIZE.csv).
//           Nothing here depends on HOW workgroups are placed: a queue whose XCD received no workgroup is simply left
//           over, and the host runs the global queue afterwards, which skips every item already done (obca_rollouts_run).
//   qmode 1 one global queue, items i = round * B + rollout; rollout state travels between XCDs, whose L2s are not coherent
//           with each other inside a kernel, through HBM: agent-scope release (L2 write-back) / acquire.
//   qmode 0 (sched == NULL) one workgroup per rollout.
// Same arithmetic on the same data in every mode: results identical word for word (tests/test_gpu_rollouts.py).
// sched[0]: next item of the global queue, sched[1]: abort flag (a wait that never ends must not hang the GPU),
// sched[2 + b]: rounds done by rollout b, sched[2 + B + 16 q]: next item of queue q.
#define OBCA_RO_SPIN_LIMIT (1 << 24)        /* x ~1 us of s_sleep: ~16 s */
#define OBCA_GETREG_XCC_ID (20 | (0 << 6) | ((4 - 1) << 11))     /* hwreg(HW_REG_XCC_ID, 0, 4) */

// The solves of one step of rollout b, group g (= sensed rectangles).  Attempts 0 .. P-1 (P = OBCA_MAX_PASSES): the passes of the
// solve (the start ladder -- the body returns at once where nothing is left to do); attempts P .. 2P-1: the same for obca_mpc8
// where obca_mpc6 failed (src/closed_loop.py:393-398).  One call site: the body is inlined once.
template <int RPL, class SHAPE>
__device__ __forceinline__ void ro_solve_step(const rollout::Dev& D, const ObcaLaunch* launches, const int g, const int b, int* ro_msg) {
    const int lane = threadIdx.x;
    bool nodge = false;
    for (int attempt = 0; attempt < 2 * OBCA_MAX_PASSES; ++attempt) {
        const ObcaLaunch* Lp = launches + g;
        if (attempt >= OBCA_MAX_PASSES) {
            if (g == 0) break;
            if (attempt == OBCA_MAX_PASSES) {
                if (lane == 0) ro_msg[0] = ro_retry(&D, g, b);
                __syncthreads();
                const int v8 = ro_msg[0];
                __syncthreads();
                if (v8 != 8) break;
            }
            Lp = launches + g + rollout::MAX_GROUPS;
        }
        if (attempt == 0 && g > 0) {      // obca_mpc6 that cannot reach its terminal set: answered by lane 0, on to obca_mpc8's turn
            if (lane == 0) ro_msg[0] = ro_screen(Lp, b);
            __syncthreads();
            const int scr = ro_msg[0];
            __syncthreads();
            if (scr == 1) { attempt = OBCA_MAX_PASSES - 1; continue; }
            nodge = scr == 2;
        }
        if (nodge && attempt == Lp->prm.opt.nstarts) { attempt = OBCA_MAX_PASSES - 1; continue; }     // ... without room to dodge: no rung
        obca_ipm_body<RPL, false, ObcaLaunchConst, SHAPE>(*(ObcaLaunchConst*)Lp, b, attempt == 0 || attempt == OBCA_MAX_PASSES);
        __syncthreads();
        {   // nothing (more) to recover: on to obca_mpc8's turn, or out
            const int st = __builtin_amdgcn_readfirstlane(((ObcaLaunchConst*)Lp)->status[b]);
            if (st == OBCA_STATUS_OK || st == OBCA_STATUS_ACCEPTABLE || st < OBCA_STATUS_NUMERIC)
                attempt = (attempt < OBCA_MAX_PASSES ? OBCA_MAX_PASSES : 2 * OBCA_MAX_PASSES) - 1;
        }
    }
}

template <int RPL>
__device__ __forceinline__ void rollout_fused_body(const rollout::Dev& D, const ObcaLaunch* launches, int n_steps, int* sched, int qmode) {
    const int lane = threadIdx.x;
    __shared__ int ro_msg[3];
    // (an item may cover OBCA_RO_BLOCK consecutive steps of its rollout: fewer hand-offs, coarser schedule)
    const int KB = sched ? OBCA_RO_BLOCK : 1;
    // this workgroup's queue: the rollouts q0, q0 + qs, q0 + 2 qs, ... (global queue: all of them)
    const bool local = sched && qmode == 2;
    const int q0 = local ? (__builtin_amdgcn_s_getreg(OBCA_GETREG_XCC_ID) & (OBCA_RO_XCDS - 1)) : 0;
    const int qs = local ? OBCA_RO_XCDS : 1;
    const int Bq = q0 < D.B ? (D.B - q0 + qs - 1) / qs : 0;
    int* const next = local ? sched + 2 + D.B + 16 * q0 : sched;
    const int total = sched ? ((n_steps + KB - 1) / KB) * Bq : n_steps;
#ifdef OBCA_RO_STATS
    long long st_wait = 0, st_work = 0, st_t = wall_clock64();
    int st_items = 0;
#endif
    for (int item = 0;; ++item) {
        int b = blockIdx.x, round = item;
        if (sched) {
            if (lane == 0) {
                int bb = -1, r = 0;
                const int i = __hip_atomic_fetch_add(next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i < total) {
                    r = i / Bq;
                    bb = q0 + qs * (i - r * Bq);
                    int spins = 0;
                    // (poll relaxed -- an acquire load in the loop would invalidate this CU's L1 on every turn -- and acquire ONCE below)
                    while (__hip_atomic_load(&sched[2 + bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) {
                        __builtin_amdgcn_s_sleep(32);
                        if (++spins > OBCA_RO_SPIN_LIMIT || __hip_atomic_load(&sched[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            __hip_atomic_store(&sched[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bb = -1;
                            break;
                        }
                    }
                }
                // the rollout's state was written by another workgroup, possibly on another XCD: agent-scope acquire on THIS CU
                // (drops its L1 lines) before anything of it is read -- also when no waiting was needed
                // (a round an earlier launch has already done -- the global queue run after the per-XCD queues -- is skipped)
                if (bb >= 0 && __hip_atomic_load(&sched[2 + bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > r) bb = -2;
                else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                ro_msg[0] = bb;
                ro_msg[2] = r;
            }
            __syncthreads();
            b = __builtin_amdgcn_readfirstlane(ro_msg[0]);
            round = __builtin_amdgcn_readfirstlane(ro_msg[2]);
            __syncthreads();
#ifdef OBCA_RO_STATS
            { const long long t = wall_clock64(); st_wait += t - st_t; st_t = t; }
#endif
            if (b == -2) continue;
            if (b < 0) break;
        } else if (item >= total || b >= D.B) {
            break;
        }
        bool running = true;
        for (int sub = 0; sub < KB && running; ++sub) {
        if (sched && round * KB + sub >= n_steps) break;
        if (lane == 0) {
            ro_prepare(&D, b);
            const long long fs = ro_flag_sel(&D, b);
            ro_msg[0] = (int)(unsigned)fs;
            ro_msg[1] = (int)(fs >> 32);
        }
        __syncthreads();
        running = ro_msg[0] == OBCA_RUN;
        const int g = __builtin_amdgcn_readfirstlane(ro_msg[1]);     // wave-uniform: the descriptor is read with scalar loads
        __syncthreads();
        if (running) {
            ro_solve_step<RPL, ShapeAny>(D, launches, g, b, ro_msg);
            if (lane == 0) ro_finish(&D, b);
            __syncthreads();
        }
        }
        if (!running && !sched) break;
        if (sched) {
            // publish: every store of this wave drained (they are in this XCD's L2 then), and -- global queue only: the next
            // workgroup may sit on another XCD -- ONE agent-scope release (writes back the XCD's L2), then the flag.
            // The explicit wait after the fence restates the one the compiler (ROCm 7.2) drops when it believes the wave has
            // nothing outstanding -- the flag must not overtake the write-back (MI355X guide, inter-workgroup visibility).
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (lane == 0) {
                if (!local) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&sched[2 + b], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifdef OBCA_RO_STATS
        { const long long t = wall_clock64(); st_work += t - st_t; st_t = t; ++st_items; }
#endif
    }
#ifdef OBCA_RO_STATS
    if (sched && lane == 0) {
        int* o = sched + 2 + D.B + 16 * OBCA_RO_XCDS + 4 * blockIdx.x;
        o[0] = (int)st_wait; o[1] = (int)st_work; o[2] = st_items; o[3] = (int)(wall_clock64() & 0x7fffffff);
    }
#endif
}

// _r4: every shape of the rollout has <= 256 rows (static obstacles only at N=5); _r6: <= 384 rows
extern "C" __global__ void __launch_bounds__(64)
obca_rollout_fused_kernel_r4(const rollout::Dev* Dp, const ObcaLaunch* launches, int n_steps, int* sched, int qmode) {
    rollout_fused_body<4>(*Dp, launches, n_steps, sched, qmode);
}
extern "C" __global__ void __launch_bounds__(64)
obca_rollout_fused_kernel_r5(const rollout::Dev* Dp, const ObcaLaunch* launches, int n_steps, int* sched, int qmode) {
    rollout_fused_body<5>(*Dp, launches, n_steps, sched, qmode);
}
extern "C" __global__ void __launch_bounds__(64)
obca_rollout_fused_kernel_r6(const rollout::Dev* Dp, const ObcaLaunch* launches, int n_steps, int* sched, int qmode) {
    rollout_fused_body<6>(*Dp, launches, n_steps, sched, qmode);
}

#elif defined(OBCA_TU_SHAPE)
// four wavefronts per instance, shape known at compile time: obca_ipm_kernel_mw_s<N>_<nO>_<M> (csrc/obca_kernel_mw_s*.hip).
// Straight-line passes and descriptors through the kernarg pointer, as measured for _mw_r3 / _mw_r5 (the other three combinations
// of the two need 128 / 304 / 320 B of scratch for (20, 5, 14) instead of 80).
#define OBCA_DEFINE_MW_SHAPE_KERNEL(N_, O_, M_)                                                                                \
    extern "C" __global__ void __launch_bounds__(OBCA_NT) obca_ipm_kernel_mw_s##N_##_##O_##_##M_(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { \
        using SH = ShapeIs<N_, O_, M_>;                                                                                        \
        solve_with_escalation<SH::RPL, OBCA_LOOP_MW, true, SH>(A, A2, A3);                                                    \
    }
OBCA_TU_SHAPE(OBCA_DEFINE_MW_SHAPE_KERNEL)
#else
#ifndef OBCA_KARG_R3
#define OBCA_KARG_R3 true
#endif
#ifndef OBCA_KARG_R5
#define OBCA_KARG_R5 true
#endif
// four wavefronts per instance: rows r = thread + 256 j, j < 3 (up to 768 rows) or j < 5 (up to 1280 rows)
extern "C" __global__ void __launch_bounds__(OBCA_NT) obca_ipm_kernel_mw_r3(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { solve_with_escalation<3, OBCA_LOOP_MW, OBCA_KARG_R3>(A, A2, A3); }
extern "C" __global__ void __launch_bounds__(OBCA_NT) obca_ipm_kernel_mw_r5(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { solve_with_escalation<-5, OBCA_LOOP_MW, OBCA_KARG_R5>(A, A2, A3); }
// shapes beyond the LDS / beyond 1280 rows (long horizons: N = 74 with five obstacles has 3976 rows): four wavefronts per
// instance, row state and every O(rows) array in the instance's slice of an HBM workspace (L2 resident), the O(N) blocks of
// the stage-serial Riccati sweep in LDS
extern "C" __global__ void __launch_bounds__(OBCA_NT) obca_ipm_kernel_gm(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3) { solve_with_escalation<0, OBCA_LOOP_GM>(A, A2, A3); }
#endif
