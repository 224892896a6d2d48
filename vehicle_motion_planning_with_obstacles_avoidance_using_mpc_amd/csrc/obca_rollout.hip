// obca_rollout.hip -- device-resident closed loop (obca_rollouts_* of include/obca_mpc.h): the harness of
// csrc/obca_rollout_core.h as one-lane-per-rollout kernels around obca_solve_batch, everything enqueued on one
// HIP stream, no host synchronisation inside a step.
#include <hip/hip_runtime.h>
#include <new>
#include <string.h>
#include <vector>
#include "obca_device.h"
#include "obca_rollout_core.h"

extern "C" __global__ void obca_rollout_fused_kernel_r4(const rollout::Dev* Dp, const ObcaLaunch* launches, int n_steps, int* sched, int qmode);
extern "C" __global__ void obca_rollout_fused_kernel_r5(const rollout::Dev* Dp, const ObcaLaunch* launches, int n_steps, int* sched, int qmode);
extern "C" __global__ void obca_rollout_fused_kernel_r6(const rollout::Dev* Dp, const ObcaLaunch* launches, int n_steps, int* sched, int qmode);

namespace {

__global__ void rollout_reset_kernel(rollout::Dev D, const double* start, const double* dyn0, double Ts0) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < D.B) rollout::reset(D, b, start, dyn0, Ts0);
}
__global__ void rollout_prepare_kernel(rollout::Dev D) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < D.B) rollout::prepare(D, b);
}
__global__ void rollout_retry_kernel(rollout::Dev D, int g) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < D.B) rollout::make_retry(D, g, b);
}
__global__ void rollout_finish_kernel(rollout::Dev D) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < D.B) rollout::finish(D, b);
}

}  // namespace

struct obca_rollouts {
    obca_rollout_dims dims;
    rollout::Dev D;
    obca_handle* solver[rollout::MAX_GROUPS];
    obca_params params;
    bool ready;
    std::vector<void*> allocs;
    // the problem shapes of one step are independent: each gets its own stream, forked from / joined to the caller's
    hipStream_t gstream[rollout::MAX_GROUPS];
    hipEvent_t fork, join[rollout::MAX_GROUPS];
    // fused path: descriptors in HBM for the persistent one-wave-per-rollout kernel (obca_kernel.hip)
    rollout::Dev* dD;
    ObcaLaunch* dL;
    int32_t* sched;           /* [2 + B + 16 x 8] work queues of the fused kernel: next item (global queue), abort flag, rounds done per rollout, next item of each XCD's queue */
    int n_slots;              /* workgroups the device holds at once (one per SIMD) */
    int sched_mode;           /* OBCA_ROLLOUT_QUEUE: 2 step-granular work queue per XCD (default), 1 one global queue, 0 one workgroup per rollout */
    ObcaLaunch hL[2 * rollout::MAX_GROUPS];            // [g]: obca_mpc4 (g = 0) / obca_mpc6, [g + MAX_GROUPS]: obca_mpc8 where obca_mpc6 failed
    bool fused_ok;
    bool queue_ran;           /* the last obca_rollouts_run used the device-side work queue (its abort flag is then checked by read) */
    int32_t rows_max;
    int64_t lds_max;
    int mode;                 /* 0 auto (fused when every shape fits the wave kernel), 1 lock-step launches */
    double warm_mu;           /* > 0: warm start enabled */
    // constants owned by the handle (copied at reset so the caller's buffers may go away)
    double *goal, *path, *As, *bs;
    int32_t* path_len;
};

namespace {

template <class T>
bool dev_alloc(obca_rollouts* r, T*& p, size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, sizeof(T) * (count ? count : 1)) != hipSuccess) return false;
    if (hipMemset(q, 0, sizeof(T) * (count ? count : 1)) != hipSuccess) { (void)hipFree(q); return false; }
    r->allocs.push_back(q);
    p = static_cast<T*>(q);
    return true;
}

}  // namespace

extern "C" void obca_rollouts_destroy(obca_rollouts* r) {
    if (!r) return;
    ObcaDeviceGuard guard(r->dims.device);
    for (void* p : r->allocs) (void)hipFree(p);
    for (int g = 0; g < rollout::MAX_GROUPS; ++g) {
        if (r->solver[g]) obca_destroy(r->solver[g]);
        if (r->gstream[g]) (void)hipStreamDestroy(r->gstream[g]);
        if (r->join[g]) (void)hipEventDestroy(r->join[g]);
    }
    if (r->fork) (void)hipEventDestroy(r->fork);
    delete r;
}

extern "C" int obca_rollouts_create(const obca_rollout_dims* d, obca_rollouts** out) {
    if (!out) return OBCA_E_INVAL;
    *out = nullptr;
    if (!d || d->N < 1 || d->N > 127 || d->n_static < 1 || d->n_dyn < 0 || d->n_dyn > OBCA_MAX_DYN ||
        d->n_static + d->n_dyn > OBCA_MAX_OBST || d->path_max < 2 || d->batch < 1 || d->max_steps < 1)
        return OBCA_E_INVAL;
    {   // fixed-time horizon: a multiple of N whose shift-in (N_fix - 5 columns of the previous plan) a free-time plan covers
        const int nf = d->N_fix > 0 ? d->N_fix : d->N;
        if (nf < d->N || nf % d->N != 0 || nf > 127 || (nf > d->N && nf - 5 > d->N)) return OBCA_E_INVAL;
    }
    for (int i = 0; i < d->n_static; ++i)
        if (d->m_static[i] < 1 || d->m_static[i] > OBCA_MAX_EDGES) return OBCA_E_INVAL;
    obca_rollouts* r = new (std::nothrow) obca_rollouts();
    if (!r) return OBCA_E_NOMEM;
    r->dims = *d;
    r->ready = false;
    for (int g = 0; g < rollout::MAX_GROUPS; ++g) { r->solver[g] = nullptr; r->gstream[g] = nullptr; r->join[g] = nullptr; }
    r->fork = nullptr;
    ObcaDeviceGuard guard(d->device);
    if (!guard.ok) { delete r; return OBCA_E_HIP; }
    rollout::Dev& D = r->D;
    memset(&D, 0, sizeof(D));
    D.B = d->batch; D.N = d->N; D.n_static = d->n_static; D.n_dyn = d->n_dyn; D.P = d->path_max; D.S = d->max_steps;
    D.Nf = d->N_fix > 0 ? d->N_fix : d->N;
    D.Nm = D.Nf > D.N ? D.Nf : D.N;
    D.Ms = 0;
    for (int i = 0; i < d->n_static; ++i) D.Ms += d->m_static[i];
    const size_t B = d->batch, N1 = d->N + 1, S = d->max_steps, nd = d->n_dyn, Nm1 = D.Nm + 1, Nf1 = D.Nf + 1;
    bool ok = true;
    ok = ok && dev_alloc(r, r->goal, B * 2) && dev_alloc(r, r->path, B * 3 * D.P) && dev_alloc(r, r->path_len, B) &&
         dev_alloc(r, r->As, B * D.Ms * 2) && dev_alloc(r, r->bs, B * D.Ms);
    D.goal = r->goal; D.path = r->path; D.path_len = r->path_len; D.As = r->As; D.bs = r->bs;
    ok = ok && dev_alloc(r, D.x0, B * 3) && dev_alloc(r, D.u0, B * 2) && dev_alloc(r, D.Ts, B) && dev_alloc(r, D.Ts_opt, B) &&
         dev_alloc(r, D.xprev, B * 3 * Nm1) && dev_alloc(r, D.dyn, B * nd * rollout::DYN_W) && dev_alloc(r, D.k, B) &&
         dev_alloc(r, D.flags, B) && dev_alloc(r, D.sel, B) && dev_alloc(r, D.xref, B * 3 * N1) && dev_alloc(r, D.xref_fix, B * 3 * Nf1) && dev_alloc(r, D.term, B * 3);
    ok = ok && dev_alloc(r, D.xc, B * (S + 1) * 3) && dev_alloc(r, D.uc, B * S * 2) && dev_alloc(r, D.Tc, B * S) &&
         dev_alloc(r, D.xol, B * S * 3 * Nm1) && dev_alloc(r, D.dh, B * S * nd * 4) && dev_alloc(r, D.vh, B * S) &&
         dev_alloc(r, D.ih, B * S) && dev_alloc(r, D.sh, B * S) && dev_alloc(r, D.vtx, B * OBCA_MAX_DYN * 8);
    int rc = ok ? OBCA_OK : OBCA_E_NOMEM;
    for (int g = 0; g <= d->n_dyn && rc == OBCA_OK; ++g) {
        const size_t Mg = D.Ms + 4 * g, Ng = g == 0 ? D.N : D.Nf, Ng1 = Ng + 1;     // group 0: free-time horizon; others: N_fix
        ok = dev_alloc(r, D.var[g], B) && dev_alloc(r, D.var8[g], B) && dev_alloc(r, D.A[g], B * Ng1 * Mg * 2) &&
             dev_alloc(r, D.b[g], B * Ng1 * Mg) && dev_alloc(r, D.xopt[g], B * 3 * Ng1) && dev_alloc(r, D.uopt[g], B * 2 * Ng) &&
             dev_alloc(r, D.ts[g], B) && dev_alloc(r, D.status[g], B) && dev_alloc(r, D.iters[g], B) &&
             dev_alloc(r, D.status8[g], B) && dev_alloc(r, D.iters8[g], B);
        if (!ok) { rc = OBCA_E_NOMEM; break; }
        obca_dims sd;
        memset(&sd, 0, sizeof(sd));
        sd.N = (int32_t)Ng; sd.n_obs = d->n_static + g; sd.max_batch = d->batch; sd.device = d->device;
        for (int i = 0; i < d->n_static; ++i) sd.m[i] = d->m_static[i];
        for (int i = 0; i < g; ++i) sd.m[d->n_static + i] = 4;
        rc = obca_create(&sd, &r->solver[g]);
        if (rc == OBCA_OK && g > 0 &&
            (hipStreamCreateWithFlags(&r->gstream[g], hipStreamNonBlocking) != hipSuccess ||
             hipEventCreateWithFlags(&r->join[g], hipEventDisableTiming) != hipSuccess))
            rc = OBCA_E_HIP;
    }
    if (rc == OBCA_OK && hipEventCreateWithFlags(&r->fork, hipEventDisableTiming) != hipSuccess) rc = OBCA_E_HIP;
    r->dD = nullptr; r->dL = nullptr; r->fused_ok = false; r->lds_max = 0; r->mode = 0; r->warm_mu = 0.0;
    if (rc == OBCA_OK && !(dev_alloc(r, r->dD, 1) && dev_alloc(r, r->dL, 2 * rollout::MAX_GROUPS))) rc = OBCA_E_NOMEM;
    r->sched = nullptr; r->n_slots = 1024; r->sched_mode = 2; r->queue_ran = false;
    if (rc == OBCA_OK && !dev_alloc(r, r->sched, (size_t)d->batch + 2 + 16 * 8 + 4 * 4096)) rc = OBCA_E_NOMEM;   // (+ per-workgroup statistics of -DOBCA_RO_STATS builds)
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d->device) == hipSuccess && cus > 0) r->n_slots = 4 * cus;
        else (void)hipGetLastError();
        // the per-XCD queues assume that HW_REG_XCC_ID & 7 names the L2 a workgroup runs under: true where the device reports
        // eight XCCs (MI355X); on anything else -- or when the runtime cannot say -- hand-offs go through HBM (global queue)
        int xccs = 0;
        if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, d->device) != hipSuccess) { (void)hipGetLastError(); xccs = 0; }
        if (xccs != OBCA_RO_XCDS) r->sched_mode = 1;
        if (const char* e = getenv("OBCA_ROLLOUT_QUEUE")) { const int v = atoi(e); if (v >= 0 && v <= 2 && !(v == 2 && xccs != OBCA_RO_XCDS)) r->sched_mode = v; }
    }
    if (rc != OBCA_OK) { obca_rollouts_destroy(r); return rc; }
    *out = r;
    return OBCA_OK;
}

extern "C" int obca_rollouts_reset(obca_rollouts* r, const double* start, const double* goal, const double* path,
                                   const int32_t* path_len, const double* static_A, const double* static_b,
                                   const double* dyn, double Ts0, double sense_dis, const obca_params* params,
                                   void* hip_stream) {
    if (!r || !start || !goal || !path || !path_len || !static_A || !static_b || !params || (r->dims.n_dyn > 0 && !dyn))
        return OBCA_E_INVAL;
    // the parameters are checked BEFORE anything is touched: a refused call leaves the handle as it was (state, descriptors, ready)
    {
        ObcaOptsDev probe;
        if (params->struct_size != (uint32_t)sizeof(obca_params) ||
            !obca_resolve_starts(&probe, params->start_order, params->single_start, params->patience, params->retry_iter, r->dims.N, params->dodge, params->terminal_screen))
            return OBCA_E_INVAL;
    }
    ObcaDeviceGuard guard(r->dims.device);
    if (!guard.ok) return OBCA_E_HIP;
    hipStream_t s = (hipStream_t)hip_stream;
    r->ready = false;                      // from here on the device state is being rewritten: not usable again until this call has succeeded
    r->queue_ran = false;                  // an aborted work queue of an EARLIER run says nothing about the rollouts started here
    rollout::Dev& D = r->D;
    const size_t B = D.B, N1 = D.Nm + 1, S = D.S, nd = D.n_dyn;
    bool ok = hipMemcpyAsync(r->goal, goal, sizeof(double) * B * 2, hipMemcpyDeviceToDevice, s) == hipSuccess &&
              hipMemcpyAsync(r->path, path, sizeof(double) * B * 3 * D.P, hipMemcpyDeviceToDevice, s) == hipSuccess &&
              hipMemcpyAsync(r->path_len, path_len, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, s) == hipSuccess &&
              hipMemcpyAsync(r->As, static_A, sizeof(double) * B * D.Ms * 2, hipMemcpyDeviceToDevice, s) == hipSuccess &&
              hipMemcpyAsync(r->bs, static_b, sizeof(double) * B * D.Ms, hipMemcpyDeviceToDevice, s) == hipSuccess;
    ok = ok && hipMemsetAsync(D.xc, 0, sizeof(double) * B * (S + 1) * 3, s) == hipSuccess &&
         hipMemsetAsync(D.uc, 0, sizeof(double) * B * S * 2, s) == hipSuccess &&
         hipMemsetAsync(D.Tc, 0, sizeof(double) * B * S, s) == hipSuccess &&
         hipMemsetAsync(D.xol, 0, sizeof(double) * B * S * 3 * N1, s) == hipSuccess &&
         hipMemsetAsync(D.dh, 0, sizeof(double) * (B * S * nd * 4 ? B * S * nd * 4 : 1), s) == hipSuccess &&
         hipMemsetAsync(D.vh, 0, sizeof(int32_t) * B * S, s) == hipSuccess &&
         hipMemsetAsync(D.ih, 0, sizeof(int32_t) * B * S, s) == hipSuccess &&
         hipMemsetAsync(D.sh, 0, sizeof(int32_t) * B * S, s) == hipSuccess;
    if (!ok) return OBCA_E_HIP;
    r->params = *params;
    D.sense_dis = sense_dis;
    D.ego_l = params->ego[0];                                  // the gate uses ego[0], ego[1] (src/closed_loop.py:594)
    D.ego_w = params->ego[1];
    hipLaunchKernelGGL(rollout_reset_kernel, dim3((D.B + 63) / 64), dim3(64), 0, s, D, start, dyn, Ts0);
    if (hipGetLastError() != hipSuccess) return OBCA_E_HIP;
    for (int g = 0; g <= D.n_dyn; ++g) {
        const int rc = obca_set_warm_start(r->solver[g], D.warm ? D.wz[g] : nullptr, D.warm ? D.wuse[g] : nullptr, r->warm_mu);
        if (rc != OBCA_OK) return rc;
    }
    // descriptors of the fused kernel: the launch obca_solve_batch would make per shape, first attempt and retry
    r->fused_ok = true;
    r->lds_max = 0;
    r->rows_max = 0;
    memset(r->hL, 0, sizeof(r->hL));
    for (int g = 0; g <= D.n_dyn; ++g)
        for (int a = 0; a < 2; ++a) {
            int64_t lds = 0;
            int wave_ok = 0;
            // obca_mpc6 (a = 0 of the groups with sensed boxes) runs the first start of the order only: its failure is answered
            // by obca_mpc8 on the same inputs (src/closed_loop.py:393-398); obca_mpc4 and obca_mpc8, after which the reference
            // stops the rollout (:401-413), run the whole ladder
            obca_params pg = r->params;
            if (g > 0 && a == 0) pg.single_start = 1;
            const int rc = obca_internal_fill_launch(r->solver[g], a ? D.var8[g] : D.var[g], D.B, D.x0, D.u0, g == 0 ? D.xref : D.xref_fix, D.A[g], D.b[g],
                                                     D.Ts, D.term, &pg, D.xopt[g], D.uopt[g], D.ts[g],
                                                     a ? D.status8[g] : D.status[g], a ? D.iters8[g] : D.iters[g], nullptr,
                                                     &r->hL[g + a * rollout::MAX_GROUPS], &lds, &wave_ok);
            if (rc != OBCA_OK) return rc;
            if (!wave_ok) r->fused_ok = false;
            if (lds > r->lds_max) r->lds_max = lds;
            // groups g >= 1 only ever solve the fixed-time variants, whose layouts have 3 (obca_mpc6) / 5 (obca_mpc8) rows
            // less than the free-time layout the handle sizes for: with one sensed box at N=5 that is 256 instead of 259
            // rows, which fits the 4-rows-per-lane kernel
            if (g > 0) r->hL[g + a * rollout::MAX_GROUPS].R_max -= 3;
            if (r->hL[g + a * rollout::MAX_GROUPS].R_max > r->rows_max) r->rows_max = r->hL[g + a * rollout::MAX_GROUPS].R_max;
        }
    if (hipMemcpyAsync(r->dD, &r->D, sizeof(rollout::Dev), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(r->dL, r->hL, sizeof(r->hL), hipMemcpyHostToDevice, s) != hipSuccess)
        return OBCA_E_HIP;
    if (r->fused_ok && r->lds_max > 64 * 1024 &&
        hipFuncSetAttribute(r->rows_max <= 256   ? reinterpret_cast<const void*>(obca_rollout_fused_kernel_r4)
                            : r->rows_max <= 320 ? reinterpret_cast<const void*>(obca_rollout_fused_kernel_r5)
                                                 : reinterpret_cast<const void*>(obca_rollout_fused_kernel_r6),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_max) != hipSuccess)
        return OBCA_E_HIP;
    r->ready = true;
    return OBCA_OK;
}

extern "C" int obca_rollouts_step(obca_rollouts* r, void* hip_stream) {
    if (!r || !r->ready) return OBCA_E_INVAL;
    ObcaDeviceGuard guard(r->dims.device);
    if (!guard.ok) return OBCA_E_HIP;
    hipStream_t s = (hipStream_t)hip_stream;
    const rollout::Dev& D = r->D;
    const dim3 grid((D.B + 63) / 64), block(64);
    hipLaunchKernelGGL(rollout_prepare_kernel, grid, block, 0, s, D);
    if (D.n_dyn > 0 && hipEventRecord(r->fork, s) != hipSuccess) return OBCA_E_HIP;
    for (int g = 0; g <= D.n_dyn; ++g) {
        // group 0 (obca_mpc4, static obstacles) stays on the caller's stream; the fixed-time groups run beside it
        hipStream_t gs = g == 0 ? s : r->gstream[g];
        if (g > 0 && hipStreamWaitEvent(gs, r->fork, 0) != hipSuccess) return OBCA_E_HIP;
        obca_params pg = r->params;
        if (g > 0) pg.single_start = 1;   // obca_mpc6: obca_mpc8 follows (see obca_rollouts_reset)
        int rc = obca_solve_batch(r->solver[g], D.var[g], D.B, D.x0, D.u0, g == 0 ? D.xref : D.xref_fix, D.A[g], D.b[g], D.Ts, D.term, &pg,
                                  D.xopt[g], D.uopt[g], D.ts[g], D.status[g], D.iters[g], nullptr, (void*)gs);
        if (rc != OBCA_OK) return rc;
        if (g == 0) continue;
        hipLaunchKernelGGL(rollout_retry_kernel, grid, block, 0, gs, D, g);
        rc = obca_solve_batch(r->solver[g], D.var8[g], D.B, D.x0, D.u0, D.xref_fix, D.A[g], D.b[g], D.Ts, D.term, &r->params,
                              D.xopt[g], D.uopt[g], D.ts[g], D.status8[g], D.iters8[g], nullptr, (void*)gs);
        if (rc != OBCA_OK) return rc;
        if (hipEventRecord(r->join[g], gs) != hipSuccess || hipStreamWaitEvent(s, r->join[g], 0) != hipSuccess) return OBCA_E_HIP;
    }
    hipLaunchKernelGGL(rollout_finish_kernel, grid, block, 0, s, D);
    if (hipGetLastError() != hipSuccess) return OBCA_E_HIP;
    return OBCA_OK;
}

extern "C" int obca_rollouts_set_warm_start(obca_rollouts* r, int enable, double mu_init) {
    if (!r || (enable && !(mu_init > 0.0))) return OBCA_E_INVAL;
    ObcaDeviceGuard guard(r->dims.device);
    if (!guard.ok) return OBCA_E_HIP;
    rollout::Dev& D = r->D;
    if (enable && !D.wz[0]) {
        for (int g = 0; g <= D.n_dyn; ++g) {
            obca_dims sd;
            memset(&sd, 0, sizeof(sd));
            sd.N = g == 0 ? D.N : D.Nf; sd.n_obs = D.n_static + g; sd.max_batch = D.B; sd.device = r->dims.device;
            for (int i = 0; i < D.n_static; ++i) sd.m[i] = r->dims.m_static[i];
            for (int i = 0; i < g; ++i) sd.m[D.n_static + i] = 4;
            const int64_t n = obca_primal_size(&sd);
            if (n < 0) return OBCA_E_INVAL;
            if (!dev_alloc(r, D.wz[g], (size_t)D.B * (size_t)n) || !dev_alloc(r, D.wuse[g], (size_t)D.B)) return OBCA_E_NOMEM;
        }
    }
    D.warm = enable ? 1 : 0;
    r->warm_mu = enable ? mu_init : 0.0;
    r->ready = false;                      // takes effect with the next obca_rollouts_reset
    return OBCA_OK;
}

extern "C" int obca_rollouts_debug_stats(obca_rollouts* r, int32_t* out, int n) {      // -DOBCA_RO_STATS builds: [n_slots][4] ints to host
    if (!r || !out || n < 0 || n > 4 * 4096) return OBCA_E_INVAL;
    ObcaDeviceGuard guard(r->dims.device);
    if (!guard.ok) return OBCA_E_HIP;
    return hipMemcpy(out, r->sched + 2 + r->D.B + 16 * 8, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? OBCA_OK : OBCA_E_HIP;
}

__global__ void rollout_debug_state_kernel(rollout::Dev D, int k, double Ts_opt, double x, double y, double th, int set_pose) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= D.B) return;
    D.k[b] = k; D.Ts_opt[b] = Ts_opt; D.flags[b] = OBCA_RUN;
    if (set_pose) { D.x0[3 * b] = x; D.x0[3 * b + 1] = y; D.x0[3 * b + 2] = th; }
}

// Test hook: the harness part of a step alone (rows S4, S5, H2-H5 of SURVEY 8a), with step counter, inherited step length
// and pose given by the caller, and what it hands the solver of group g copied to HOST buffers.
extern "C" int obca_rollouts_debug_harness(obca_rollouts* r, int32_t k, double Ts_opt, const double* x0_host, int32_t g,
                                           int32_t* variant, double* A, double* b, void* hip_stream) {
    if (!r || !r->ready || k < 0 || k >= r->D.S || g < 0 || g > r->D.n_dyn) return OBCA_E_INVAL;
    ObcaDeviceGuard guard(r->dims.device);
    if (!guard.ok) return OBCA_E_HIP;
    hipStream_t s = (hipStream_t)hip_stream;
    const rollout::Dev& D = r->D;
    const dim3 grid((D.B + 63) / 64), block(64);
    hipLaunchKernelGGL(rollout_debug_state_kernel, grid, block, 0, s, D, (int)k, Ts_opt, x0_host ? x0_host[0] : 0.0,
                       x0_host ? x0_host[1] : 0.0, x0_host ? x0_host[2] : 0.0, x0_host ? 1 : 0);
    hipLaunchKernelGGL(rollout_prepare_kernel, grid, block, 0, s, D);
    if (hipGetLastError() != hipSuccess) return OBCA_E_HIP;
    const size_t B = D.B, N1 = (g == 0 ? D.N : D.Nf) + 1, Mg = D.Ms + 4 * g;
    auto cp = [&](void* dst, const void* src, size_t bytes) {
        return !dst || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s) == hipSuccess;
    };
    if (!cp(variant, D.var[g], sizeof(int32_t) * B) || !cp(A, D.A[g], sizeof(double) * B * N1 * Mg * 2) ||
        !cp(b, D.b[g], sizeof(double) * B * N1 * Mg)) return OBCA_E_HIP;
    return hipStreamSynchronize(s) == hipSuccess ? OBCA_OK : OBCA_E_HIP;
}

extern "C" int obca_rollouts_queue_mode(const obca_rollouts* r) { return r ? r->sched_mode : OBCA_E_INVAL; }

extern "C" int obca_rollouts_set_mode(obca_rollouts* r, int mode) {
    if (!r || mode < 0 || mode > 1) return OBCA_E_INVAL;
    r->mode = mode;
    return OBCA_OK;
}

extern "C" int obca_rollouts_run(obca_rollouts* r, int32_t n_steps, void* hip_stream) {
    if (!r || !r->ready || n_steps < 0) return OBCA_E_INVAL;
    ObcaDeviceGuard guard(r->dims.device);
    if (!guard.ok) return OBCA_E_HIP;
    if (n_steps == 0) return OBCA_OK;
    if (r->fused_ok && r->mode == 0) {
        // persistent workgroups taking (round, rollout) items from a queue (sched_mode 2: one queue per XCD, 1: one global
        // queue); one workgroup per rollout when the queue is off (0).  After the per-XCD queues the global queue runs once
        // more: it skips every item already done -- all of them, unless an XCD received no workgroup (nothing in HIP
        // promises a placement) -- so the result never depends on where the hardware put the workgroups.
        const bool queue = r->sched_mode != 0 && (long long)n_steps * r->D.B < (1ll << 30);
        int* sched = queue ? r->sched : nullptr;
        r->queue_ran = queue;
        if (queue && hipMemsetAsync(r->sched, 0, sizeof(int32_t) * ((size_t)r->D.B + 2 + 16 * 8), (hipStream_t)hip_stream) != hipSuccess) return OBCA_E_HIP;
        const int grid = queue ? (r->D.B < r->n_slots ? r->D.B : r->n_slots) : r->D.B;
        // test hook (OBCA_ROLLOUT_LOCAL_STEPS = k): the per-XCD queues cover only the first k steps (rounded down to whole
        // rounds), so that the clean-up pass has real work -- the rounds an XCD without workgroups would leave over
        int local_steps = (int)n_steps;
        if (const char* e = getenv("OBCA_ROLLOUT_LOCAL_STEPS")) { const int v = atoi(e) / OBCA_RO_BLOCK * OBCA_RO_BLOCK; if (v > 0 && v < local_steps) local_steps = v; }
        for (int qmode = queue ? r->sched_mode : 0; ; qmode = 1) {
            const int steps_now = qmode == 2 ? local_steps : (int)n_steps;
            if (r->rows_max <= 256)
                hipLaunchKernelGGL(obca_rollout_fused_kernel_r4, dim3(grid), dim3(64), (size_t)r->lds_max, (hipStream_t)hip_stream,
                                   (const rollout::Dev*)r->dD, (const ObcaLaunch*)r->dL, steps_now, sched, qmode);
            else if (r->rows_max <= 320)
                hipLaunchKernelGGL(obca_rollout_fused_kernel_r5, dim3(grid), dim3(64), (size_t)r->lds_max, (hipStream_t)hip_stream,
                                   (const rollout::Dev*)r->dD, (const ObcaLaunch*)r->dL, steps_now, sched, qmode);
            else
                hipLaunchKernelGGL(obca_rollout_fused_kernel_r6, dim3(grid), dim3(64), (size_t)r->lds_max, (hipStream_t)hip_stream,
                                   (const rollout::Dev*)r->dD, (const ObcaLaunch*)r->dL, steps_now, sched, qmode);
            if (qmode != 2) break;
        }
        return hipGetLastError() == hipSuccess ? OBCA_OK : OBCA_E_HIP;
    }
    for (int i = 0; i < n_steps; ++i) {
        const int rc = obca_rollouts_step(r, hip_stream);
        if (rc != OBCA_OK) return rc;
    }
    return OBCA_OK;
}

extern "C" int obca_rollouts_read(obca_rollouts* r, double* x_closed, double* u_closed, double* T_closed,
                                  double* x_openloop, int32_t* variant_hist, int32_t* iters_hist, int32_t* status_hist,
                                  double* dyn_hist, int32_t* steps, int32_t* flags, void* hip_stream) {
    if (!r || !r->ready) return OBCA_E_INVAL;
    ObcaDeviceGuard guard(r->dims.device);
    if (!guard.ok) return OBCA_E_HIP;
    hipStream_t s = (hipStream_t)hip_stream;
    const rollout::Dev& D = r->D;
    const size_t B = D.B, N1 = D.Nm + 1, S = D.S, nd = D.n_dyn;
    auto cp = [&](void* dst, const void* src, size_t bytes) {
        return !dst || bytes == 0 || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess;
    };
    const bool ok = cp(x_closed, D.xc, sizeof(double) * B * (S + 1) * 3) && cp(u_closed, D.uc, sizeof(double) * B * S * 2) &&
                    cp(T_closed, D.Tc, sizeof(double) * B * S) && cp(x_openloop, D.xol, sizeof(double) * B * S * 3 * N1) &&
                    cp(variant_hist, D.vh, sizeof(int32_t) * B * S) && cp(iters_hist, D.ih, sizeof(int32_t) * B * S) && cp(status_hist, D.sh, sizeof(int32_t) * B * S) &&
                    cp(dyn_hist, D.dh, sizeof(double) * B * S * nd * 4) && cp(steps, D.k, sizeof(int32_t) * B) &&
                    cp(flags, D.flags, sizeof(int32_t) * B);
    if (!ok) return OBCA_E_HIP;
    // the fused kernel's work queue gives up (instead of hanging the GPU) if a rollout's previous round is never published:
    // that must not pass for a result
    // (only after a run that used the queue: this is the one place where read synchronises with the stream)
    int32_t aborted = 0;
    if (r->queue_ran && (hipMemcpyAsync(&aborted, r->sched + 1, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
                         hipStreamSynchronize(s) != hipSuccess)) return OBCA_E_HIP;
    return aborted ? OBCA_E_HIP : OBCA_OK;
}
