// obca_rollout_core.h -- the closed-loop harness of one rollout as serial code (one GPU lane per rollout).
//
// Restates the body of the reference's receding-horizon loop (src/closed_loop.py:345-432) for a batch laid out
// as structure-of-arrays in HBM: update_obstacle (:445-486), sensor (:591-629), update_reference_trajectory
// (:502-528), the fixed-time reference preparation (:360-374, update_path(allAviable=1) :570-587 with its
// resampling ratio int(N_fix / N_free)),
// rebuild_lObs (src/demo_setting.py:457-473) + obstacle_H_Represent (src/model_obstacle.py:37-102) for the
// moving rectangles, the dispatch mpc4 / mpc6 -> mpc8 (:380-398) and the state advance (:400-432).
// Compiles for the device (obca_rollout.hip) and for the host (tests/native/rollout_host.cpp); floating-point
// contraction is switched off so that the exact `==` edge tests of the reference see identical numbers.
#ifndef OBCA_ROLLOUT_CORE_H
#define OBCA_ROLLOUT_CORE_H

#include <math.h>
#include <stdint.h>
#include "../../include/obca_mpc.h"

#if defined(__HIPCC__)
#define RO_FN __host__ __device__ inline
#else
#define RO_FN inline
#endif

// exact arithmetic inside the harness (no fused multiply-add): first statement of every function below
#if defined(__clang__)
#define RO_EXACT _Pragma("clang fp contract(off)")
#else
#define RO_EXACT /* g++: built with -ffp-contract=off */
#endif

namespace rollout {

constexpr int DYN_W = 13;                      // reference 11-tuple + cos(theta), sin(theta)
constexpr int MAX_GROUPS = OBCA_MAX_DYN + 1;   // group g = number of sensed moving obstacles in the solve

struct Dev {
    int32_t B, N, n_static, Ms, n_dyn, P, S;
    int32_t Nf, Nm;                 // horizon of the fixed-time problem (N_fix) and max(N, Nf): stride of xprev / xol
    double sense_dis, ego_l, ego_w;
    // per-rollout constants
    const double *goal, *path, *As, *bs;
    const int32_t* path_len;
    // per-rollout state (H0: src/closed_loop.py:18-111)
    double *x0, *u0, *Ts, *Ts_opt, *xprev, *dyn;
    int32_t *k, *flags, *sel;
    // solver inputs: reference window of the free-time problem [B,3,N+1] / of the fixed-time problems [B,3,Nf+1]
    double *xref, *xref_fix, *term;
    // per group
    int32_t *var[MAX_GROUPS], *var8[MAX_GROUPS];
    double *A[MAX_GROUPS], *b[MAX_GROUPS];
    double *xopt[MAX_GROUPS], *uopt[MAX_GROUPS], *ts[MAX_GROUPS];
    int32_t *status[MAX_GROUPS], *iters[MAX_GROUPS], *status8[MAX_GROUPS], *iters8[MAX_GROUPS];
    // history (what the reference hands to its plot routine, :435-441)
    double *xc, *uc, *Tc, *xol, *dh;
    double* vtx;                    // [B, OBCA_MAX_DYN, 4, 2] vertices of the present rectangles of the current step (scratch of prepare())
    int32_t *vh, *ih, *sh;
    // optional warm start (obca_rollouts_set_warm_start): per group the primal vectors kept by the solver and the flags
    // telling it which rollouts may start from them
    int32_t warm;
    double* wz[MAX_GROUPS];
    int32_t* wuse[MAX_GROUPS];
};

// one polygon edge -> one row [a0 a1 | b]; branch order and exact comparisons of src/model_obstacle.py:63-89
RO_FN void edge_row(double x1, double y1, double x2, double y2, double* a, double* bb) {
    RO_EXACT
    if (x1 == x2) {
        if (y2 < y1) { a[0] = 1.0; a[1] = 0.0; *bb = x1; }
        else { a[0] = -1.0; a[1] = 0.0; *bb = -x1; }
        return;
    }
    if (y1 == y2) {
        if (x1 < x2) { a[0] = 0.0; a[1] = 1.0; *bb = y1; }
        else { a[0] = 0.0; a[1] = -1.0; *bb = -y1; }
        return;
    }
    const double s = (y2 - y1) / (x2 - x1);
    const double c = y1 - s * x1;
    if (x1 < x2) { a[0] = -s; a[1] = 1.0; *bb = c; }
    else { a[0] = s; a[1] = -1.0; *bb = -c; }
}

// clockwise rectangle (src/demo_setting.py:405-429)
RO_FN void rect_vertices(double cx, double cy, double c, double s, double length, double width, double V[4][2]) {
    RO_EXACT
    const double l = length / 2, w = width / 2;
    V[0][0] = cx - l * c - w * s; V[0][1] = cy - l * s + w * c;
    V[1][0] = cx + l * c - w * s; V[1][1] = cy + l * s + w * c;
    V[2][0] = cx + l * c + w * s; V[2][1] = cy + l * s - w * c;
    V[3][0] = cx - l * c + w * s; V[3][1] = cy - l * s - w * c;
}

RO_FN bool at_goal(const Dev& D, int b) {
    RO_EXACT
    const double dx = D.x0[3 * b] - D.goal[2 * b], dy = D.x0[3 * b + 1] - D.goal[2 * b + 1];
    return !(dx * dx + dy * dy >= 0.1);                                  // src/closed_loop.py:345
}

RO_FN void reset(const Dev& D, int b, const double* start, const double* dyn0, double Ts0) {
    RO_EXACT
    for (int j = 0; j < 3; ++j) D.x0[3 * b + j] = start[3 * b + j];
    D.u0[2 * b] = 0.0; D.u0[2 * b + 1] = 0.0;
    D.Ts[b] = Ts0; D.Ts_opt[b] = Ts0;
    D.k[b] = 0; D.sel[b] = 0;
    for (int t = 0; t < D.n_dyn * DYN_W; ++t) D.dyn[(size_t)b * D.n_dyn * DYN_W + t] = dyn0[(size_t)b * D.n_dyn * DYN_W + t];
    for (int t = 0; t < 3 * (D.Nm + 1); ++t) D.xprev[(size_t)b * 3 * (D.Nm + 1) + t] = 0.0;
    for (int j = 0; j < 3; ++j) D.xc[((size_t)b * (D.S + 1)) * 3 + j] = start[3 * b + j];
    D.flags[b] = at_goal(D, b) ? OBCA_DONE_GOAL : OBCA_RUN;
}

// everything before the solve; writes the solver inputs of the group the rollout falls into
RO_FN void prepare(const Dev& D, int b) {
    RO_EXACT
    const int N = D.N, N1 = N + 1, nd = D.n_dyn;
    for (int g = 0; g <= nd; ++g) { D.var[g][b] = 0; if (D.warm) D.wuse[g][b] = 0; }
    if (D.flags[b] != OBCA_RUN) return;
    const int k = D.k[b];
    const int prev_group = D.sel[b];                 // problem shape of the previous step (which succeeded if k > 0)
    double Ts_opt = D.Ts_opt[b];
    const double* x0 = D.x0 + 3 * b;

    // H2 update_obstacle: appear at k == t_start, afterwards advance by Ts_opt * v along the heading
    // (index lists packed four bits per entry and the vertex lists in a per-rollout HBM row: as local arrays they were 304 B
    // of stack frame -- scratch -- in the fused closed-loop kernel)
    static_assert(OBCA_MAX_DYN <= 8, "index lists are packed four bits per entry into one int");
    unsigned present = 0, sensed = 0;
    int np = 0;
    double (*V)[4][2] = reinterpret_cast<double (*)[4][2]>(D.vtx + (size_t)b * OBCA_MAX_DYN * 8);
    for (int i = 0; i < nd; ++i) {
        double* info = D.dyn + ((size_t)b * nd + i) * DYN_W;
        double* rec = D.dh + (((size_t)b * D.S + k) * nd + i) * 4;
        rec[2] = 0.0; rec[3] = 0.0;
        if ((double)k < info[9]) { rec[0] = info[0]; rec[1] = info[1]; continue; }
        if ((double)k > info[9]) {
            info[0] = info[0] + Ts_opt * info[5] * info[11];
            info[1] = info[1] + Ts_opt * info[5] * info[12];
        }
        rec[0] = info[0]; rec[1] = info[1]; rec[2] = 1.0;
        rect_vertices(info[0], info[1], info[11], info[12], info[3], info[4], V[np]);
        present |= (unsigned)i << (4 * np);
        ++np;
    }

    // H3 sensor: any vertex of a present obstacle within sense_dis of the car-front point
    int ns = 0;
    {
        const double c = cos(x0[2]), s = sin(x0[2]), l = D.ego_l, w = D.ego_w;
        const double v2x = x0[0] + l * c - w * s, v2y = x0[1] + l * s + w * c;
        const double v3x = x0[0] + l * c + w * s, v3y = x0[1] + l * s - w * c;
        const double fx = (v2x + v3x) / 2, fy = (v2y + v3y) / 2;
        for (int j = 0; j < np; ++j)
            for (int q = 0; q < 4; ++q) {
                const double dx = fx - V[j][q][0], dy = fy - V[j][q][1];
                if (sqrt(dx * dx + dy * dy) <= D.sense_dis) {
                    const int pj = (int)((present >> (4 * j)) & 15u);
                    sensed |= (unsigned)pj << (4 * ns);
                    ++ns;
                    D.dh[(((size_t)b * D.S + k) * nd + pj) * 4 + 3] = 1.0;
                    break;
                }
            }
    }
    const bool fixtime = ns > 0;

    // H4 update_reference_trajectory: window from the first strict minimum of the squared distance; N+1 points for the
    // free-time problem, N_fix+1 for the fixed-time one (:354, :361)
    const bool free_step = (k == 0 || !fixtime);
    const int Nw = free_step ? N : D.Nf, Nw1 = Nw + 1;
    double* xr = free_step ? D.xref + (size_t)b * 3 * N1 : D.xref_fix + (size_t)b * 3 * Nw1;
    {
        const double* p = D.path + (size_t)b * 3 * D.P;
        const int len = D.path_len[b];
        double best = 100000.0;
        int i0 = 0;
        for (int i = 0; i < len; ++i) {
            const double dx = x0[0] - p[i], dy = x0[1] - p[D.P + i];
            const double d = dx * dx + dy * dy;
            if (d < best) { best = d; i0 = i; }
        }
        for (int t = 0; t < Nw1; ++t) {
            const int i = (i0 + t < len - 1) ? i0 + t : len - 1;
            for (int j = 0; j < 3; ++j) xr[j * Nw1 + t] = p[j * D.P + i];
        }
    }

    const double* As = D.As + (size_t)b * D.Ms * 2;
    const double* bs = D.bs + (size_t)b * D.Ms;
    if (free_step) {                                                     // H6: obca_mpc4 on the static obstacles
        double* Ag = D.A[0] + (size_t)b * N1 * D.Ms * 2;
        double* bg = D.b[0] + (size_t)b * N1 * D.Ms;
        for (int kk = 0; kk < N1; ++kk) {
            for (int q = 0; q < 2 * D.Ms; ++q) Ag[kk * 2 * D.Ms + q] = As[q];
            for (int q = 0; q < D.Ms; ++q) bg[kk * D.Ms + q] = bs[q];
        }
        D.sel[b] = 0;
        D.var[0][b] = 4;
        if (D.warm && k > 0 && prev_group == 0) D.wuse[0][b] = 1;
        return;
    }

    // H5 fixed-time reference: shifted previous plan in front (:363-364), then update_path(allAviable=1) (:570-587): the first
    // N_free segments of the window are resampled into int(N_fix/N_free) points each (numpy.linspace without the end
    // point: start + j * ((stop - start) / ratio)), the window's LAST point closes the list, yaw recomputed, step
    // rescaled, Ts overwritten (q7).  N_fix is a multiple of N_free here, so the list has N_fix + 1 points again.
    const int Nf = D.Nf, Nf1 = Nf + 1, ratio = Nf / N;
    const double* xp = D.xprev + (size_t)b * 3 * (D.Nm + 1);
    for (int i = 0; i < Nf - 5; ++i)
        for (int j = 0; j < 3; ++j) xr[j * Nf1 + i] = xp[j * (D.Nm + 1) + i + 1];
    if (ratio > 1) {
        const double lastx = xr[Nf], lasty = xr[Nf1 + Nf];
        // in place, last segment first: segment i reads columns i, i+1 and writes columns i*ratio .. (i+1)*ratio - 1, all
        // beyond every column a LATER (smaller) segment still reads
        for (int i = N - 1; i >= 0; --i) {
            const double ax = xr[i], bx_ = xr[i + 1], ay = xr[Nf1 + i], by = xr[Nf1 + i + 1];
            const double stx = (bx_ - ax) / (double)ratio, sty = (by - ay) / (double)ratio;
            for (int j = 0; j < ratio; ++j) {
                xr[i * ratio + j] = (double)j * stx + ax;
                xr[Nf1 + i * ratio + j] = (double)j * sty + ay;
            }
        }
        xr[Nf] = lastx; xr[Nf1 + Nf] = lasty;
    }
    for (int i = 0; i < Nf; ++i) xr[2 * Nf1 + i] = atan2(xr[Nf1 + i + 1] - xr[Nf1 + i], xr[i + 1] - xr[i]);
    xr[2 * Nf1 + Nf] = xr[2 * Nf1 + Nf - 1];
    Ts_opt = ((double)N * Ts_opt) / (double)Nf;
    D.Ts_opt[b] = Ts_opt;
    D.Ts[b] = Ts_opt;
    D.term[3 * b] = x0[0] + 5; D.term[3 * b + 1] = 1.0; D.term[3 * b + 2] = 9.0;      // :371

    // S5/S4: static rows, then the first ns PRESENT rectangles (q8) moved with the SENSED obstacles' velocities
    const int g = ns, Mg = D.Ms + 4 * ns;
    double* Ag = D.A[g] + (size_t)b * Nf1 * Mg * 2;
    double* bg = D.b[g] + (size_t)b * Nf1 * Mg;
    for (int kk = 0; kk < Nf1; ++kk) {
        double* Ak = Ag + (size_t)kk * Mg * 2;
        double* bk = bg + (size_t)kk * Mg;
        for (int q = 0; q < 2 * D.Ms; ++q) Ak[q] = As[q];
        for (int q = 0; q < D.Ms; ++q) bk[q] = bs[q];
        for (int j = 0; j < ns; ++j) {
            const double* info = D.dyn + ((size_t)b * nd + (int)((sensed >> (4 * j)) & 15u)) * DYN_W;
            const double sx = Ts_opt * info[5] * info[11] * (double)kk;
            const double sy = Ts_opt * info[5] * info[12] * (double)kk;
            // (the four moved vertices as scalars: a local array indexed in a loop is a stack frame)
            const double x0_ = V[j][0][0] + sx, y0_ = V[j][0][1] + sy, x1_ = V[j][1][0] + sx, y1_ = V[j][1][1] + sy;
            const double x2_ = V[j][2][0] + sx, y2_ = V[j][2][1] + sy, x3_ = V[j][3][0] + sx, y3_ = V[j][3][1] + sy;
            double* Ar = Ak + 2 * (D.Ms + 4 * j);
            double* br = bk + D.Ms + 4 * j;
            edge_row(x0_, y0_, x1_, y1_, Ar, br);
            edge_row(x1_, y1_, x2_, y2_, Ar + 2, br + 1);
            edge_row(x2_, y2_, x3_, y3_, Ar + 4, br + 2);
            edge_row(x3_, y3_, x0_, y0_, Ar + 6, br + 3);
        }
    }
    D.sel[b] = g;
    D.var[g][b] = 6;
    if (D.warm && k > 0 && prev_group == g) D.wuse[g][b] = 1;
}

RO_FN bool status_feasible(int st) { return st == OBCA_STATUS_OK || st == OBCA_STATUS_ACCEPTABLE; }

// between the obca_mpc6 launch and the obca_mpc8 launch of group g (:393-398)
RO_FN void make_retry(const Dev& D, int g, int b) {
    RO_EXACT
    D.var8[g][b] = (D.var[g][b] == 6 && !status_feasible(D.status[g][b])) ? 8 : 0;
}

// state advance (:400-432)
RO_FN void finish(const Dev& D, int b) {
    RO_EXACT
    if (D.flags[b] != OBCA_RUN) return;
    const int g = D.sel[b], k = D.k[b];
    const int N = (g == 0) ? D.N : D.Nf, N1 = N + 1, Nm1 = D.Nm + 1;         // horizon of the problem this step solved
    int st = D.status[g][b], it = D.iters[g][b], variant = (g == 0) ? 4 : 6;
    if (g > 0 && D.var8[g][b] == 8) { st = D.status8[g][b]; it += D.iters8[g][b]; variant = 8; }
    D.vh[(size_t)b * D.S + k] = variant;
    D.ih[(size_t)b * D.S + k] = it;
    D.sh[(size_t)b * D.S + k] = st;
    if (!status_feasible(st)) { D.flags[b] = OBCA_DONE_FAILED; return; }
    const double* xo = D.xopt[g] + (size_t)b * 3 * N1;
    const double* uo = D.uopt[g] + (size_t)b * 2 * N;
    for (int j = 0; j < 3; ++j)
        for (int t = 0; t < Nm1; ++t) {
            const double v = t < N1 ? xo[j * N1 + t] : 0.0;
            D.xprev[(size_t)b * 3 * Nm1 + j * Nm1 + t] = v;
            D.xol[((size_t)b * D.S + k) * 3 * Nm1 + j * Nm1 + t] = v;
        }
    for (int j = 0; j < 2; ++j) {
        D.u0[2 * b + j] = uo[j * N];
        D.uc[((size_t)b * D.S + k) * 2 + j] = uo[j * N];
    }
    for (int j = 0; j < 3; ++j) {
        D.x0[3 * b + j] = xo[j * N1 + 1];
        D.xc[((size_t)b * (D.S + 1) + k + 1) * 3 + j] = xo[j * N1 + 1];
    }
    D.Ts_opt[b] = D.ts[g][b];
    D.Tc[(size_t)b * D.S + k] = D.ts[g][b];
    D.k[b] = k + 1;
    if (k + 1 == D.S) D.flags[b] = OBCA_DONE_CAP;
    else if (at_goal(D, b)) D.flags[b] = OBCA_DONE_GOAL;
}

}  // namespace rollout
#endif
