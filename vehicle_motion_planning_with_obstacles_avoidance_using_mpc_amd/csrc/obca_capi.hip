// obca_capi.hip -- host side of libobca_mpc.so: the C ABI declared in include/obca_mpc.h.
// Plain pointers and sizes only; every failure is a return code (no exception leaves this file).

#include <hip/hip_runtime.h>
#include <new>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "obca_device.h"
#include "obca_lpi_core.h"

extern "C" __global__ void obca_ipm_kernel_r4(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);
extern "C" __global__ void obca_ipm_kernel_r5(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);
extern "C" __global__ void obca_ipm_kernel_r6(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);
extern "C" __global__ void obca_ipm_kernel_mw_r3(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);          // four wavefronts per instance (obca_kernel_mw.hip)
extern "C" __global__ void obca_ipm_kernel_mw_r5(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);
extern "C" __global__ void obca_ipm_kernel_gm(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);             // four wavefronts, rows in an HBM workspace
extern "C" __global__ void obca_ipm_kernel_gm1(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);            // one wavefront, rows in an HBM workspace
// compile-time-shape instantiations of the one-wavefront kernel (csrc/obca_kernel_s*.hip; list: csrc/obca_device.h OBCA_SHAPES)
#define OBCA_DECLARE_SHAPE_KERNEL(N_, O_, M_) extern "C" __global__ void obca_ipm_kernel_s##N_##_##O_##_##M_(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);
OBCA_SHAPES(OBCA_DECLARE_SHAPE_KERNEL)
#define OBCA_DECLARE_MW_SHAPE_KERNEL(N_, O_, M_) extern "C" __global__ void obca_ipm_kernel_mw_s##N_##_##O_##_##M_(ObcaLaunch A, ObcaLaunch A2, ObcaLaunch A3);
OBCA_MW_SHAPES(OBCA_DECLARE_MW_SHAPE_KERNEL)
typedef void (*obca_wave_kernel_t)(ObcaLaunch, ObcaLaunch, ObcaLaunch);
extern "C" __global__ void obca_lpi_kernel(ObcaLaunch A, double* ws, unsigned long long stride, const int* offm, int ipw);

struct obca_handle {
    obca_dims dims;
    int32_t M, n_max, R_max, inst_off;
    int32_t offm[OBCA_MAX_OBST + 1];
    int two_sided;                     // -1: where only the four-wavefront kernels fit (default), 0: never, 1: always
    int64_t lds_pad;          // dev knob OBCA_LDS_PAD: bytes added to the one-wavefront kernels' LDS request (occupancy experiments)
    int64_t lds_bytes, lds_bytes_mw;   // one-wavefront kernels; four-wavefront kernels (+ the two-sided sweep's storage) -- both incl. the SOC scratch where it lives in LDS
    int32_t soc_lds, soc_lds_mw;       // its offset there (doubles), 0 = in HBM (csrc/obca_device.h: obca_soc_lds_wave / _mw)
    int64_t lds_bytes_gm, gm_doubles;  // obca_ipm_kernel_gm: its LDS (O(N) blocks only) and its HBM workspace per workgroup
    int32_t inst_off_gm;
    bool gm_ok;
    double* gm_ws;                     // allocated on first use: max_batch slices
    bool gm_ws_failed;                 // ... and that allocation failed once: auto mode stops choosing the workspace kernels for this handle
    double* prof;
    int mode;                 /* 0 auto, 1 wave-per-instance (LDS), 2 lane-per-instance (HBM workspace), 3 four waves per instance */
    obca_wave_kernel_t shape_kernel;   /* instantiation of the one-wavefront kernel for exactly this shape, or nullptr */
    obca_wave_kernel_t shape_kernel_mw; /* likewise of the four-wavefront LDS kernel */
    bool specialise;          /* use it (default; OBCA_SPECIALISE=0 / obca_set_shape_specialisation(h, 0): the generic kernels) */
    bool wave_ok;             /* the one-wavefront LDS kernel can hold this shape */
    bool mw_ok;               /* the four-wavefront LDS kernel can hold this shape */
    double* warm_z;           /* obca_set_warm_start */
    const int32_t* warm_use;
    double warm_mu;
    double *cert_z, *cert_y;  /* obca_set_certificate_buffers */
    double* soc_ws;           /* scratch of the second-order correction (wave kernels), max_batch x soc_stride doubles */
    double* ws;               /* lane kernel workspace, allocated on first use */
    size_t ws_stride;
    int* d_offm;
    int64_t ws_doubles;
};

namespace {

constexpr int MW = OBCA_MAX_EDGES + 6;

bool dims_ok(const obca_dims* d) {
    if (!d) return false;
    if (d->N < 1 || d->N > 127) return false;     // the wave kernel only ever sees N < 64 (rows / LDS), the lane kernel any
    if (d->n_obs < 1 || d->n_obs > OBCA_MAX_OBST) return false;
    for (int i = 0; i < d->n_obs; ++i)
        if (d->m[i] < 1 || d->m[i] > OBCA_MAX_EDGES) return false;
    return true;
}

// Auto mode, shapes beyond the one-wavefront LDS kernel (> 384 rows): ONE wavefront per instance with the row state in the HBM
// workspace (obca_ipm_kernel_gm1) where an instance has at most three obstacles, the four-wavefront LDS kernel otherwise.  Measured
// (round 5, tools/gpu_gm1_shapes.py, 8192 instances of the C3 generator): three obstacles / 6 rows per stage N = 12 / 16 / 20 / 26:
// 114.6 / 151.6 / 178.7 / 255.5 ms against 152.9 / 182.8 / 194.0 / 270.7 ms on four wavefronts (with few rows per stage the stage-serial
// sweep dominates, and four times as many instances in flight hide its latency); five obstacles / 14 rows per stage N = 8 ... 14:
// 187 ... 276 ms against 131 ... 214 ms (the local blocks of five obstacles keep four wavefronts busy); four obstacles / 10 rows per stage,
// obca_mpc6 / 8, N = 10 ... 20: 140 ... 302 ms against 107 ... 179 ms; three obstacles with the fixed-time variants gain like the free-time
// one (tools/gpu_gm1_four.py).  Same words as the
// four-wavefront kernels with the one-sided sweep.  A function of the SHAPE only -- and only of the MEASURED region (round 6, advisor):
// horizons up to N = 26 that the four-wavefront LDS kernel could run as well.  Longer horizons and shapes beyond the LDS keep the
// kernels they had before gm1 existed (four wavefronts: LDS resident where it fits, HBM workspace otherwise) until someone measures
// them; and a handle whose workspace cannot be allocated falls back to the LDS-resident kernel, which needs none.
#define OBCA_GM1_MAX_N 26
bool auto_gm1(const obca_handle* h) { return h->mode == 0 && !h->wave_ok && h->mw_ok && h->gm_ok && h->dims.n_obs <= 3 && h->dims.N <= OBCA_GM1_MAX_N && !h->gm_ws_failed; }

// (the carve-up itself: csrc/obca_device.h: obca_shape_sizes, shared with the kernels)
int64_t lds_doubles(int N, int nO, int M, int& n_max, int& R_max, int& inst_off) {
    const ObcaShapeSizes z = obca_shape_sizes(N, nO, M);
    n_max = z.n_max; R_max = z.R_max; inst_off = z.inst_off;
    return z.lds_doubles;
}

// the same for obca_ipm_kernel_gm (GM branch of the carve-up): LDS doubles, offset of the instance block, workspace doubles
int64_t gm_doubles(int N, int nO, int M, int n_max, int R_max, int& inst_off, int64_t& ws) {
    const int N1 = N + 1, np = N1 * nO;
    int64_t t = 0, g = 0;
    auto take = [&](int64_t c) { t += (c + 1) & ~int64_t(1); };
    auto takeG = [&](int64_t c) { g += (c + 1) & ~int64_t(1); };
    takeG(n_max); takeG(n_max > 120 ? n_max : 120); take(5 * N1 + 1); takeG(n_max);
    for (int i = 0; i < 5; ++i) takeG(R_max);
    take(3 * N1 + 3);
    take(N1); take(N1); takeG(2 * np); take(N1); take(N1); takeG(2 * np);
    takeG(2 * np); takeG(2 * np); takeG(2 * np);
    takeG(N1 * M * 2); takeG(N1 * M); take(3 * N1);
    take(36 * N1); take(8 * N1);
    {
        const int64_t nx = (n_max + 1) & ~1, nr = (R_max + 1) & ~1, ny = (int64_t)MW * 4 * np;
        takeG(ny > nx + nr ? ny : nx + nr);
    }
    take(36 * N1); takeG(12 * np);
    take(6 * N1); take(12 * N1); take(2 * N1); take(9 * (N1 + 1));
    take(120);               // FG / Mall / mall
    take(32); take(8);
    take(3 * N1 + 3); take(3 * N1 + 3); take(2 * N1 + 2);      // mirrors of the soft rows' E^-1, ghat; inputs + time scale
    const int64_t rs = (int64_t)((R_max + 255) / 256) * 256;
    for (int i = 0; i < 15; ++i) takeG(rs);                    // row state (OBCA_ROW_FIELDS)
    inst_off = (int)t;
    take(OBCA_INST_DOUBLES);
    ws = g;
    return t;
}

}  // namespace

extern "C" int64_t obca_lds_bytes(const obca_dims* d) {
    if (!dims_ok(d)) return -1;
    int M = 0;
    for (int i = 0; i < d->n_obs; ++i) M += d->m[i];
    int n_max, R_max, io;
    return 8 * lds_doubles(d->N, d->n_obs, M, n_max, R_max, io);
}

extern "C" int obca_create(const obca_dims* d, obca_handle** out) {
    if (!out) return OBCA_E_INVAL;
    *out = nullptr;
    if (!dims_ok(d) || d->max_batch < 1) return OBCA_E_INVAL;
    obca_handle* h = new (std::nothrow) obca_handle;
    if (!h) return OBCA_E_NOMEM;
    h->dims = *d;
    h->M = 0;
    h->offm[0] = 0;
    for (int i = 0; i < OBCA_MAX_OBST; ++i) {
        if (i < d->n_obs) h->M += d->m[i];
        h->offm[i + 1] = h->M;
    }
    h->lds_bytes = 8 * lds_doubles(d->N, d->n_obs, h->M, h->n_max, h->R_max, h->inst_off);
    h->lds_bytes_mw = h->lds_bytes + 8 * (OBCA_ZK_DOUBLES(d->N) + OBCA_HYB_DOUBLES(h->R_max));
    h->wave_ok = !(h->lds_bytes + OBCA_LDS_STATIC_BYTES > OBCA_LDS_CU_BYTES || h->R_max > 384);     // rows live in registers: <= 6 per lane
    h->mw_ok = !(h->lds_bytes_mw + OBCA_LDS_STATIC_BYTES > OBCA_LDS_CU_BYTES || h->R_max > 1280);   // 256 threads x 3 or 5 rows; up to 64 B of static LDS
    // the second-order correction's scratch joins the LDS request where that costs no occupancy (csrc/obca_device.h)
    h->soc_lds = h->wave_ok ? obca_soc_lds_wave(d->N, d->n_obs, h->M) : 0;
    h->soc_lds_mw = h->mw_ok ? obca_soc_lds_mw(d->N, d->n_obs, h->M) : 0;
    if (h->soc_lds) h->lds_bytes += 8 * obca_soc_doubles(d->N, d->n_obs, h->M);
    if (h->soc_lds_mw) h->lds_bytes_mw += 8 * obca_soc_doubles(d->N, d->n_obs, h->M);
    h->lds_bytes_gm = 8 * (gm_doubles(d->N, d->n_obs, h->M, h->n_max, h->R_max, h->inst_off_gm, h->gm_doubles) + OBCA_ZK_DOUBLES(d->N));
    h->gm_ok = h->lds_bytes_gm + OBCA_LDS_STATIC_BYTES <= OBCA_LDS_CU_BYTES;
    h->gm_ws = nullptr; h->gm_ws_failed = false;
    ObcaDeviceGuard guard(d->device);
    if (!guard.ok) { delete h; return OBCA_E_HIP; }
    // a kernel whose LDS request the runtime refuses is simply not offered (the lane kernel serves every shape)
    if (h->gm_ok && h->lds_bytes_gm > 64 * 1024 &&
        (hipFuncSetAttribute(reinterpret_cast<const void*>(obca_ipm_kernel_gm), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)h->lds_bytes_gm) != hipSuccess ||
         hipFuncSetAttribute(reinterpret_cast<const void*>(obca_ipm_kernel_gm1), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)h->lds_bytes_gm) != hipSuccess)) {
        (void)hipGetLastError();
        h->gm_ok = false;
    }
    if (h->mw_ok && h->lds_bytes_mw > 64 * 1024 &&
        hipFuncSetAttribute(h->R_max <= 768 ? reinterpret_cast<const void*>(obca_ipm_kernel_mw_r3)
                                            : reinterpret_cast<const void*>(obca_ipm_kernel_mw_r5),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_mw) != hipSuccess) {
        (void)hipGetLastError();
        h->mw_ok = false;
    }
    if (h->wave_ok && h->lds_bytes > 64 * 1024) {
        const void* fn = h->R_max <= 256   ? reinterpret_cast<const void*>(obca_ipm_kernel_r4)
                         : h->R_max <= 320 ? reinterpret_cast<const void*>(obca_ipm_kernel_r5)
                                           : reinterpret_cast<const void*>(obca_ipm_kernel_r6);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) {
            (void)hipGetLastError();
            h->wave_ok = false;
        }
    }
    h->mode = 0;
    h->shape_kernel = nullptr;
    h->shape_kernel_mw = nullptr;
    h->specialise = true;
#define OBCA_MATCH_MW_SHAPE_KERNEL(N_, O_, M_) if (d->N == N_ && d->n_obs == O_ && h->M == M_) h->shape_kernel_mw = obca_ipm_kernel_mw_s##N_##_##O_##_##M_;
    OBCA_MW_SHAPES(OBCA_MATCH_MW_SHAPE_KERNEL)
    if (h->shape_kernel_mw && h->mw_ok && h->lds_bytes_mw > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(h->shape_kernel_mw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes_mw) != hipSuccess) {
        (void)hipGetLastError();
        h->shape_kernel_mw = nullptr;
    }
#define OBCA_MATCH_SHAPE_KERNEL(N_, O_, M_) if (d->N == N_ && d->n_obs == O_ && h->M == M_) h->shape_kernel = obca_ipm_kernel_s##N_##_##O_##_##M_;
    OBCA_SHAPES(OBCA_MATCH_SHAPE_KERNEL)
    if (const char* e = getenv("OBCA_SPECIALISE")) h->specialise = atoi(e) != 0;
    h->lds_pad = 0;
    if (const char* e = getenv("OBCA_LDS_PAD")) { const long v = atol(e); if (v > 0 && h->lds_bytes + v <= 64 * 1024) h->lds_pad = v; }
    h->two_sided = -1;
    if (const char* e = getenv("OBCA_TWO_SIDED")) { const int v = atoi(e); if (v >= -1 && v <= 1) h->two_sided = v; }
    if (const char* e = getenv("OBCA_MODE")) {
        const int m = atoi(e);                                     // out of range or not available for this shape: auto
        if (m >= 0 && m <= 5 && !(m == 1 && !h->wave_ok) && !(m == 3 && !h->mw_ok) && !((m == 4 || m == 5) && !h->gm_ok)) h->mode = m;
    }
    h->ws = nullptr; h->d_offm = nullptr;
    h->ws_stride = ((size_t)d->max_batch + 63) / 64 * 64;
    h->ws_doubles = lpi::carve(d->N, d->n_obs, h->M, h->n_max, h->R_max).total;
    h->prof = nullptr;
    h->warm_z = nullptr; h->warm_use = nullptr; h->warm_mu = 0.0;
    h->cert_z = nullptr; h->cert_y = nullptr;
    h->soc_ws = nullptr;
    if (h->wave_ok || h->mw_ok || h->gm_ok) {
        const size_t stride = (size_t)h->n_max + 2 * (size_t)h->R_max + 2 * (size_t)(d->N + 1) * d->n_obs;
        if (hipMalloc(&h->soc_ws, sizeof(double) * stride * (size_t)d->max_batch) != hipSuccess) { h->soc_ws = nullptr; delete h; return OBCA_E_NOMEM; }
    }
    *out = h;
    return OBCA_OK;
}

extern "C" void obca_destroy(obca_handle* h) {
    if (!h) return;
    ObcaDeviceGuard guard(h->dims.device);
    if (h->ws) (void)hipFree(h->ws);
    if (h->gm_ws) (void)hipFree(h->gm_ws);
    if (h->soc_ws) (void)hipFree(h->soc_ws);
    if (h->d_offm) (void)hipFree(h->d_offm);
    delete h;
}

extern "C" int obca_set_mode(obca_handle* h, int mode) {
    if (!h || mode < 0 || mode > 5) return OBCA_E_INVAL;
    if ((mode == 4 || mode == 5) && !h->gm_ok) return OBCA_E_LDS;
    if (mode == 1 && !h->wave_ok) return OBCA_E_LDS;
    if (mode == 3 && !h->mw_ok) return OBCA_E_LDS;
    h->mode = mode;
    return OBCA_OK;
}

extern "C" int64_t obca_primal_size(const obca_dims* d) {
    if (!dims_ok(d)) return -1;
    int M = 0;
    for (int i = 0; i < d->n_obs; ++i) M += d->m[i];
    return (int64_t)(d->N + 1) * (3 + M + 4 * d->n_obs) + 2 * d->N + 1;
}

extern "C" int obca_set_warm_start(obca_handle* h, double* z, const int32_t* use, double mu_init) {
    if (!h || (z && !(mu_init > 0.0))) return OBCA_E_INVAL;
    h->warm_z = z; h->warm_use = z ? use : nullptr; h->warm_mu = mu_init;
    return OBCA_OK;
}

extern "C" int64_t obca_dual_size(const obca_dims* d) {
    if (!dims_ok(d)) return -1;
    int M = 0;
    for (int i = 0; i < d->n_obs; ++i) M += d->m[i];
    int n_max, R_max, io;
    (void)lds_doubles(d->N, d->n_obs, M, n_max, R_max, io);
    return (int64_t)R_max + 2 * (int64_t)(d->N + 1) * d->n_obs;
}

extern "C" int obca_set_certificate_buffers(obca_handle* h, double* z, double* y) {
    if (!h) return OBCA_E_INVAL;
    h->cert_z = z; h->cert_y = y;
    return OBCA_OK;
}

extern "C" int obca_set_shape_specialisation(obca_handle* h, int on) {
    if (!h || on < 0 || on > 1) return OBCA_E_INVAL;
    h->specialise = on != 0;
    return OBCA_OK;
}

extern "C" int obca_shape_is_specialised(const obca_handle* h) {
    if (!h) return OBCA_E_INVAL;
    if (!h->specialise || h->mode == 2 || h->mode == 4 || h->mode == 5 || auto_gm1(h)) return 0;
    const bool mw = h->mode == 3 || (h->mode == 0 && !h->wave_ok && h->mw_ok);
    return mw ? (h->shape_kernel_mw ? 1 : 0) : (h->wave_ok && h->shape_kernel ? 1 : 0);
}

extern "C" int obca_set_two_sided_sweep(obca_handle* h, int on) {
    if (!h || on < -1 || on > 1) return OBCA_E_INVAL;
    h->two_sided = on;
    return OBCA_OK;
}

extern "C" void obca_set_profile_buffer(obca_handle* h, double* prof) { if (h) h->prof = prof; }

int obca_internal_fill_launch(obca_handle* h, const int32_t* variant, int32_t B,
                              const double* x0, const double* u0, const double* xref,
                              const double* A, const double* b, const double* Ts, const double* term,
                              const obca_params* p,
                              double* xopt, double* uopt, double* ts_opt, int32_t* status, int32_t* iters,
                              double* info, ObcaLaunch* out, int64_t* lds_bytes, int* wave_ok) {
    if (!h || !variant || !x0 || !u0 || !xref || !A || !b || !Ts || !p || !xopt || !uopt || !ts_opt || !status ||
        !iters || !out)
        return OBCA_E_INVAL;
    if (B < 0 || B > h->dims.max_batch) return OBCA_E_INVAL;
    if (p->struct_size != (uint32_t)sizeof(obca_params)) return OBCA_E_INVAL;      // another layout, or never initialised (obca_params_init)
    ObcaLaunch& L = *out;
    memset(&L, 0, sizeof(L));
    L.B = B; L.N = h->dims.N; L.nO = h->dims.n_obs; L.M = h->M; L.n_max = h->n_max; L.R_max = h->R_max; L.inst_off = h->inst_off;
    L.two_sided = h->two_sided < 0 ? (h->wave_ok ? 0 : 1) : h->two_sided;
    for (int i = 0; i <= OBCA_MAX_OBST; ++i) L.offm[i] = h->offm[i];
    L.variant = variant; L.x0 = x0; L.u0 = u0; L.xref = xref; L.A = A; L.b = b; L.Ts = Ts; L.term = term;
    L.xopt = xopt; L.uopt = uopt; L.ts_opt = ts_opt; L.status = status; L.iters = iters; L.info = info; L.prof = h->prof;
    L.warm_z = h->warm_z; L.warm_use = h->warm_use; L.warm_mu = h->warm_mu;
    L.cert_z = h->cert_z; L.cert_y = h->cert_y;
    L.soc_ws = h->soc_ws; L.soc_lds = h->soc_lds;
    auto cpw = [](ObcaWeightsDev& d, const obca_weights& s) {
        // the reference's double loops use Q[i,j] for every (i,j): only the symmetric part matters
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {
                d.Q[3 * a + c] = 0.5 * (s.Q[3 * a + c] + s.Q[3 * c + a]);
                d.P[3 * a + c] = 0.5 * (s.P[3 * a + c] + s.P[3 * c + a]);
            }
        for (int a = 0; a < 2; ++a)
            for (int c = 0; c < 2; ++c) {
                d.R1[2 * a + c] = 0.5 * (s.R1[2 * a + c] + s.R1[2 * c + a]);
                d.R2[2 * a + c] = 0.5 * (s.R2[2 * a + c] + s.R2[2 * c + a]);
            }
    };
    cpw(L.prm.free_time, p->free_time);
    cpw(L.prm.fixed_time, p->fixed_time);
    for (int j = 0; j < 2; ++j) { L.prm.xL[j] = p->xL[j]; L.prm.xU[j] = p->xU[j]; L.prm.uL[j] = p->uL[j]; L.prm.uU[j] = p->uU[j]; }
    const double Lc = p->ego[0] + p->ego[2], Wc = p->ego[1] + p->ego[3];       // src/obca.py:1019-1026
    L.prm.gego[0] = Lc / 2; L.prm.gego[1] = Wc / 2; L.prm.gego[2] = Lc / 2; L.prm.gego[3] = Wc / 2;
    L.prm.off = Lc / 2 - p->ego[2];
    L.prm.dmin = p->dmin;
    L.prm.opt.tol = p->tol > 0 ? p->tol : 1e-8;
    L.prm.opt.rho = p->rho > 0 ? p->rho : 1e4;
    L.prm.opt.feas_tol = p->feas_tol > 0 ? p->feas_tol : 1e-6;
    L.prm.opt.max_iter_free = p->max_iter_free > 0 ? p->max_iter_free : 3000;
    L.prm.opt.max_iter_fixed = p->max_iter_fixed > 0 ? p->max_iter_fixed : 1000;
    L.prm.opt.max_soc = p->max_soc == 0 ? OBCA_MAX_SOC : (p->max_soc < 0 ? 0 : p->max_soc);
    if (!obca_resolve_starts(&L.prm.opt, p->start_order, p->single_start, p->patience, p->retry_iter, h->dims.N, p->dodge, p->terminal_screen)) return OBCA_E_INVAL;
    if (lds_bytes) *lds_bytes = h->lds_bytes;
    if (wave_ok) *wave_ok = h->wave_ok ? 1 : 0;
    return OBCA_OK;
}

extern "C" int obca_solve_batch(obca_handle* h, const int32_t* variant, int32_t B,
                                const double* x0, const double* u0, const double* xref,
                                const double* A, const double* b, const double* Ts, const double* term,
                                const obca_params* p,
                                double* xopt, double* uopt, double* ts_opt, int32_t* status, int32_t* iters,
                                double* info, void* hip_stream) {
    ObcaLaunch L;
    const int rc = obca_internal_fill_launch(h, variant, B, x0, u0, xref, A, b, Ts, term, p, xopt, uopt, ts_opt, status,
                                             iters, info, &L, nullptr, nullptr);
    if (rc != OBCA_OK) return rc;
    if (B == 0) return OBCA_OK;
    ObcaDeviceGuard guard(h->dims.device);
    if (!guard.ok) return OBCA_E_HIP;
    // wave kernel (working set in LDS) whenever the shape fits one CU; the lane kernel (working set in an HBM
    // workspace, one instance per lane) takes the shapes beyond the LDS -- measured on MI355X it is latency bound
    // (every access is an L2/HBM round trip at one wave per SIMD) and 4-10x slower where both run
    if (h->mode == 1 && !h->wave_ok) return OBCA_E_LDS;
    if (h->mode == 3 && !h->mw_ok) return OBCA_E_LDS;
    if (h->mode == 4 || h->mode == 5 || auto_gm1(h) || (h->mode == 0 && !h->wave_ok && !h->mw_ok && h->gm_ok)) {
        // shapes beyond the LDS: four wavefronts per instance, rows and O(rows) arrays in the handle's HBM workspace
        if (!h->gm_ok) return OBCA_E_LDS;
        // (OBCA_FAIL_WORKSPACE_ALLOC in the environment: the allocation is treated as failed -- the only way to exercise the branch below
        // on a 288 GB device; tests/test_gpu_edge_cases.py)
        if (!h->gm_ws && (getenv("OBCA_FAIL_WORKSPACE_ALLOC") != nullptr ||
                          hipMalloc(&h->gm_ws, sizeof(double) * (size_t)h->gm_doubles * (size_t)h->dims.max_batch) != hipSuccess)) {
            (void)hipGetLastError();
            h->gm_ws = nullptr;
            // auto mode with an LDS-resident alternative: run that instead of failing the call (it needs no workspace); an explicit
            // mode 4 / 5 and shapes only the workspace kernels hold report the failure
            if (!(h->mode == 0 && h->mw_ok)) return OBCA_E_NOMEM;
            h->gm_ws_failed = true;
            return obca_solve_batch(h, variant, B, x0, u0, xref, A, b, Ts, term, p, xopt, uopt, ts_opt, status, iters, info, hip_stream);
        }
        L.inst_off = h->inst_off_gm; L.soc_lds = 0;
        L.gm_ws = h->gm_ws; L.gm_stride = h->gm_doubles;
        ObcaLaunch L2 = L;
        if (h->mode == 5 || auto_gm1(h)) { L.two_sided = 0; L2.two_sided = 0; hipLaunchKernelGGL(obca_ipm_kernel_gm1, dim3(B), dim3(64), (size_t)h->lds_bytes_gm, (hipStream_t)hip_stream, L, L2, L2); }
        else hipLaunchKernelGGL(obca_ipm_kernel_gm, dim3(B), dim3(256), (size_t)h->lds_bytes_gm, (hipStream_t)hip_stream, L, L2, L2);
        return hipGetLastError() == hipSuccess ? OBCA_OK : OBCA_E_HIP;
    }
    // one wavefront per instance where the rows fit its registers; four wavefronts (one CU) per instance for bigger
    // shapes that still fit the LDS; the lane kernel for everything else
    // (the choice depends on the SHAPE only, never on the batch size: the answer to an instance must not depend on how many
    // neighbours it was submitted with.  Measured: four wavefronts per instance would shorten launches of B <= 256 by 10-12 %,
    // measured with OBCA_MODE=3 at batch sizes <= 256 -- callers that want that latency ask for it, as the obca() class does)
    const bool mw = h->mode == 3 || (h->mode == 0 && !h->wave_ok && h->mw_ok);
    const bool lane = !mw && (h->mode == 2 || !h->wave_ok);
    if (mw || !lane) {
        // the kernels run the further passes of the start ladder (penalty escalation, next starts) themselves, from their own copy of the descriptor
        if (mw) L.soc_lds = h->soc_lds_mw;
        ObcaLaunch L2 = L;
        if (mw && h->shape_kernel_mw && h->specialise)
            hipLaunchKernelGGL(h->shape_kernel_mw, dim3(B), dim3(256), (size_t)h->lds_bytes_mw, (hipStream_t)hip_stream, L, L2, L2);
        else if (mw)
            hipLaunchKernelGGL(h->R_max <= 768 ? obca_ipm_kernel_mw_r3 : obca_ipm_kernel_mw_r5, dim3(B), dim3(256),
                               (size_t)h->lds_bytes_mw, (hipStream_t)hip_stream, L, L2, L2);
        else if (h->shape_kernel && h->specialise)
            hipLaunchKernelGGL(h->shape_kernel, dim3(B), dim3(64), (size_t)(h->lds_bytes + h->lds_pad), (hipStream_t)hip_stream, L, L2, L2);
        else if (h->R_max <= 256)
            hipLaunchKernelGGL(obca_ipm_kernel_r4, dim3(B), dim3(64), (size_t)(h->lds_bytes + h->lds_pad), (hipStream_t)hip_stream, L, L2, L2);
        else if (h->R_max <= 320)
            hipLaunchKernelGGL(obca_ipm_kernel_r5, dim3(B), dim3(64), (size_t)(h->lds_bytes + h->lds_pad), (hipStream_t)hip_stream, L, L2, L2);
        else
            hipLaunchKernelGGL(obca_ipm_kernel_r6, dim3(B), dim3(64), (size_t)(h->lds_bytes + h->lds_pad), (hipStream_t)hip_stream, L, L2, L2);
    } else {
        if (!h->ws || !h->d_offm) {
            if (!h->ws && hipMalloc(&h->ws, sizeof(double) * (size_t)h->ws_doubles * h->ws_stride) != hipSuccess) { h->ws = nullptr; return OBCA_E_NOMEM; }
            if (!h->d_offm && hipMalloc(&h->d_offm, sizeof(int) * (OBCA_MAX_OBST + 1)) != hipSuccess) { h->d_offm = nullptr; return OBCA_E_NOMEM; }
            if (hipMemcpy(h->d_offm, h->offm, sizeof(int) * (OBCA_MAX_OBST + 1), hipMemcpyHostToDevice) != hipSuccess) return OBCA_E_HIP;
        }
        int ipw = 64;                               // one wave per SIMD (256 CUs x 4) before the waves get fatter
        while (ipw > 1 && (B + ipw - 1) / ipw < 1024) ipw >>= 1;
        hipLaunchKernelGGL(obca_lpi_kernel, dim3((B + ipw - 1) / ipw), dim3(64), 0, (hipStream_t)hip_stream, L, h->ws,
                           (unsigned long long)h->ws_stride, h->d_offm, ipw);
    }
    if (hipGetLastError() != hipSuccess) return OBCA_E_HIP;
    return OBCA_OK;
}

extern "C" const char* obca_strerror(int code) {
    switch (code) {
        case OBCA_OK: return "ok";
        case OBCA_E_INVAL: return "invalid argument or shape beyond compiled limits";
        case OBCA_E_NOMEM: return "out of host memory";
        case OBCA_E_HIP: return "HIP runtime call failed";
        case OBCA_E_LDS: return "instance does not fit in 160 KiB of LDS";
        default: return "unknown error";
    }
}

extern "C" const char* obca_version(void) { return "obca_mpc 0.6 (gfx950)"; }
extern "C" void obca_params_init(obca_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(*p);
}
