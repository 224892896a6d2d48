// CPU exercise of the closed-loop harness core (csrc/obca_rollout_core.h) -- tests only.  Same source as the
// device kernels of csrc/obca_rollout.hip; the solves go through the CPU build of the lane-per-instance core
// (lpi_host.cpp).  Lets the harness be compared with the Python `closedLoop` mirror without a GPU.
#include "lpi_host.cpp"
#include "../../vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd/csrc/obca_rollout_core.h"
#include <vector>

extern "C" int rollout_host_run(const obca_rollout_dims* d, const double* start, const double* goal, const double* path,
                                const int* path_len, const double* As, const double* bs, const double* dyn, double Ts0,
                                double sense_dis, const HostParams* hp, int n_steps,
                                double* x_closed, double* u_closed, double* T_closed, double* x_openloop,
                                int* variant_hist, int* iters_hist, int* status_hist, double* dyn_hist, int* steps, int* flags,
                                double* xref_hist /* [B,S,3,max(N,N_fix)+1] solver reference of every step, may be NULL */,
                                double warm_mu /* > 0: warm start as obca_rollouts_set_warm_start does */) {
    using namespace rollout;
    Dev D;
    memset(&D, 0, sizeof(D));
    D.B = d->batch; D.N = d->N; D.n_static = d->n_static; D.n_dyn = d->n_dyn; D.P = d->path_max; D.S = d->max_steps;
    D.Nf = d->N_fix > 0 ? d->N_fix : d->N;
    D.Nm = D.Nf > D.N ? D.Nf : D.N;
    D.Ms = 0;
    for (int i = 0; i < d->n_static; ++i) D.Ms += d->m_static[i];
    D.sense_dis = sense_dis; D.ego_l = hp->ego[0]; D.ego_w = hp->ego[1];
    const size_t B = D.B, N1 = D.N + 1, S = D.S, nd = D.n_dyn, Nm1 = D.Nm + 1, Nf1 = D.Nf + 1;
    std::vector<std::vector<double>> dv;
    std::vector<std::vector<int>> iv;
    auto da = [&](size_t n) { dv.emplace_back(n ? n : 1, 0.0); return dv.back().data(); };
    auto ia = [&](size_t n) { iv.emplace_back(n ? n : 1, 0); return iv.back().data(); };
    D.goal = goal; D.path = path; D.path_len = path_len; D.As = As; D.bs = bs;
    D.x0 = da(B * 3); D.u0 = da(B * 2); D.Ts = da(B); D.Ts_opt = da(B); D.xprev = da(B * 3 * Nm1); D.dyn = da(B * nd * DYN_W);
    D.k = ia(B); D.flags = ia(B); D.sel = ia(B); D.xref = da(B * 3 * N1); D.xref_fix = da(B * 3 * Nf1); D.term = da(B * 3);
    D.xc = x_closed; D.uc = u_closed; D.Tc = T_closed; D.xol = x_openloop; D.dh = dyn_hist; D.vh = variant_hist; D.ih = iters_hist; D.sh = status_hist;
    D.vtx = da(B * OBCA_MAX_DYN * 8);
    for (int g = 0; g <= D.n_dyn; ++g) {
        const size_t Mg = D.Ms + 4 * g, N = g == 0 ? D.N : D.Nf, N1 = N + 1;
        D.var[g] = ia(B); D.var8[g] = ia(B); D.A[g] = da(B * N1 * Mg * 2); D.b[g] = da(B * N1 * Mg);
        D.xopt[g] = da(B * 3 * N1); D.uopt[g] = da(B * 2 * N); D.ts[g] = da(B);
        D.status[g] = ia(B); D.iters[g] = ia(B); D.status8[g] = ia(B); D.iters8[g] = ia(B);
        if (warm_mu > 0.0) { D.wz[g] = da(B * (N1 * (3 + Mg + 4 * (d->n_static + g)) + 2 * N + 1)); D.wuse[g] = ia(B); }
    }
    D.warm = warm_mu > 0.0 ? 1 : 0;
    for (int b = 0; b < D.B; ++b) reset(D, b, start, dyn, Ts0);
    for (int step = 0; step < n_steps; ++step) {
        for (int b = 0; b < D.B; ++b) {
            const int k = D.k[b];
            prepare(D, b);
            if (xref_hist && D.flags[b] == OBCA_RUN) {       // [B,S,3,Nm+1]: the reference window the solver of this step is given
                const bool fr = D.sel[b] == 0;
                const size_t n1 = fr ? N1 : Nf1;
                const double* src = fr ? D.xref + (size_t)b * 3 * N1 : D.xref_fix + (size_t)b * 3 * Nf1;
                for (size_t j = 0; j < 3; ++j)
                    for (size_t t = 0; t < n1; ++t) xref_hist[(((size_t)b * S + k) * 3 + j) * Nm1 + t] = src[j * n1 + t];
            }
        }
        for (int g = 0; g <= D.n_dyn; ++g) {
            int m[OBCA_MAX_OBST];
            for (int i = 0; i < d->n_static; ++i) m[i] = d->m_static[i];
            for (int i = 0; i < g; ++i) m[d->n_static + i] = 4;
            HostParams h6 = *hp;
            if (g > 0) h6.single_start = 1;   // obca_mpc6: obca_mpc8 follows (csrc/obca_rollout.hip)
            int rc = lpi_host_solve_batch_warm(g == 0 ? D.N : D.Nf, d->n_static + g, m, D.var[g], D.B, D.x0, D.u0, g == 0 ? D.xref : D.xref_fix, D.A[g], D.b[g], D.Ts, D.term,
                                               &h6, D.xopt[g], D.uopt[g], D.ts[g], D.status[g], D.iters[g], nullptr,
                                               D.warm ? D.wz[g] : nullptr, D.warm ? D.wuse[g] : nullptr, warm_mu);
            if (rc) return rc;
            if (g == 0) continue;
            for (int b = 0; b < D.B; ++b) make_retry(D, g, b);
            rc = lpi_host_solve_batch_warm(D.Nf, d->n_static + g, m, D.var8[g], D.B, D.x0, D.u0, D.xref_fix, D.A[g], D.b[g], D.Ts, D.term,
                                           hp, D.xopt[g], D.uopt[g], D.ts[g], D.status8[g], D.iters8[g], nullptr,
                                           D.warm ? D.wz[g] : nullptr, D.warm ? D.wuse[g] : nullptr, warm_mu);
            if (rc) return rc;
        }
        for (int b = 0; b < D.B; ++b) finish(D, b);
    }
    for (int b = 0; b < D.B; ++b) { steps[b] = D.k[b]; flags[b] = D.flags[b]; }
    return 0;
}

// Test hook, CPU counterpart of obca_rollouts_debug_harness (csrc/obca_rollout.hip): for every step k of `ks` the harness
// part of a step alone -- step counter k, inherited step length Ts_opt, pose x0 (NULL: the start pose) -- on ONE rollout;
// what it hands the solver of group g after the LAST of them goes to variant / A / b.
extern "C" int rollout_host_debug_harness(const obca_rollout_dims* d, const double* start, const double* goal, const double* path,
                                          const int* path_len, const double* As, const double* bs, const double* dyn, double sense_dis,
                                          const double* ego, const int* ks, int n_ks, double Ts_opt, const double* x0, int g,
                                          int* variant, double* A, double* b) {
    using namespace rollout;
    Dev D;
    memset(&D, 0, sizeof(D));
    D.B = 1; D.N = d->N; D.n_static = d->n_static; D.n_dyn = d->n_dyn; D.P = d->path_max; D.S = d->max_steps;
    D.Nf = d->N_fix > 0 ? d->N_fix : d->N;
    D.Nm = D.Nf > D.N ? D.Nf : D.N;
    for (int i = 0; i < d->n_static; ++i) D.Ms += d->m_static[i];
    D.sense_dis = sense_dis; D.ego_l = ego[0]; D.ego_w = ego[1];
    const size_t N1 = D.N + 1, S = D.S, nd = D.n_dyn, Nm1 = D.Nm + 1, Nf1 = D.Nf + 1;
    std::vector<std::vector<double>> dv;
    std::vector<std::vector<int>> iv;
    auto da = [&](size_t n) { dv.emplace_back(n ? n : 1, 0.0); return dv.back().data(); };
    auto ia = [&](size_t n) { iv.emplace_back(n ? n : 1, 0); return iv.back().data(); };
    D.goal = goal; D.path = path; D.path_len = path_len; D.As = As; D.bs = bs;
    D.x0 = da(3); D.u0 = da(2); D.Ts = da(1); D.Ts_opt = da(1); D.xprev = da(3 * Nm1); D.dyn = da(nd * DYN_W);
    D.k = ia(1); D.flags = ia(1); D.sel = ia(1); D.xref = da(3 * N1); D.xref_fix = da(3 * Nf1); D.term = da(3);
    D.xc = da((S + 1) * 3); D.uc = da(S * 2); D.Tc = da(S); D.xol = da(S * 3 * Nm1); D.dh = da(S * (nd ? nd : 1) * 4);
    D.vh = ia(S); D.ih = ia(S); D.sh = ia(S);
    D.vtx = da(OBCA_MAX_DYN * 8);
    for (int q = 0; q <= D.n_dyn; ++q) {
        const size_t Mq = D.Ms + 4 * q, Nq1 = (q == 0 ? D.N : D.Nf) + 1;
        D.var[q] = ia(1); D.var8[q] = ia(1); D.A[q] = da(Nq1 * Mq * 2); D.b[q] = da(Nq1 * Mq);
    }
    reset(D, 0, start, dyn, Ts_opt);
    for (int i = 0; i < n_ks; ++i) {
        D.k[0] = ks[i]; D.Ts_opt[0] = Ts_opt; D.flags[0] = OBCA_RUN;
        if (x0) for (int j = 0; j < 3; ++j) D.x0[j] = x0[j];
        prepare(D, 0);
    }
    if (g < 0 || g > D.n_dyn) return -22;
    const size_t Mg = D.Ms + 4 * g, Ng1 = (g == 0 ? D.N : D.Nf) + 1;
    *variant = D.var[g][0];
    for (size_t t = 0; t < Ng1 * Mg * 2; ++t) A[t] = D.A[g][t];
    for (size_t t = 0; t < Ng1 * Mg; ++t) b[t] = D.b[g][t];
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// CPU exercise of the global-planner core (csrc/obca_astar_core.h), same source as the device kernel.
#include "../../vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd/csrc/obca_astar_core.h"

extern "C" int astar_host_batch(const unsigned char* grid, int B, int rows, int cols, const int* start, const int* goal,
                                const double* yaw9, int path_max, double* path, int* path_len) {
    std::vector<unsigned char> work(astar::work_bytes(rows * cols));
    for (int b = 0; b < B; ++b)
        path_len[b] = astar::plan(grid + (size_t)b * rows * cols, rows, cols, start[2 * b], start[2 * b + 1], goal[2 * b],
                                  goal[2 * b + 1], work.data(), yaw9, path + (size_t)b * 3 * path_max, path_max);
    return 0;
}
