// obca_device.h -- structures shared by the HIP kernel and the host side of libobca_mpc.so.
#ifndef OBCA_DEVICE_H
#define OBCA_DEVICE_H

#include <stdint.h>
#include "../../include/obca_mpc.h"

// reference constants
#define OBCA_ACC_MAX0 0.6                       /* src/obca.py:932  */
#define OBCA_ACC_MAX1 0.5235987755982988        /* pi/6, src/obca.py:933 */
#define OBCA_T_MIN 1e-4                         /* src/obca.py:963  */

// IPOPT defaults restated in oracle/ipm_dense.py (same names)
#define OBCA_MU_INIT 0.1
#define OBCA_KAPPA_MU 0.2
#define OBCA_THETA_MU 1.5
#define OBCA_KAPPA_EPS 10.0
#define OBCA_TAU_MIN 0.99
#define OBCA_BOUND_PUSH 1e-2
#define OBCA_BOUND_FRAC 1e-2
#define OBCA_KAPPA_D 1e-5
#define OBCA_KAPPA_SIGMA 1e10
#define OBCA_S_MAX 100.0
#define OBCA_GAMMA_THETA 1e-5
#define OBCA_GAMMA_PHI 1e-8
#define OBCA_DELTA 1.0
#define OBCA_S_THETA 1.1
#define OBCA_S_PHI 2.3
#define OBCA_ETA_PHI 1e-8
#define OBCA_GAMMA_ALPHA 0.05
#define OBCA_THETA_MAX_FACT 1e4
#define OBCA_THETA_MIN_FACT 1e-4
#define OBCA_DELTA_W_MIN 1e-20
#define OBCA_DELTA_W_0 1e-4
#define OBCA_DELTA_W_MAX 1e40
#define OBCA_KAPPA_W_PLUS 8.0
#define OBCA_KAPPA_W_PLUS_BAR 100.0
#define OBCA_KAPPA_W_MINUS (1.0 / 3.0)
#define OBCA_MAX_GRADIENT 100.0
#define OBCA_ACCEPTABLE_ITER 15
#define OBCA_KAPPA_SOC 0.99
#define OBCA_MAX_SOC 4
#define OBCA_RHO_ESCALATION 100.0   /* obca_mpc4 only: one retry with rho x 100 when elastic variables remain */
/* Restart phase (every variant; rule and measurements in oracle/ipm_dense.py:solve): a solve that ended without a feasible
   point is repeated ONCE from the reference window -- poses = xref (first pose x0), inputs by differences clipped to their
   box, free-time problem: the time scale at which the window is driven at this fraction of the speed bound -- with a
   larger initial barrier parameter (IPOPT's restoration phase likewise raises mu to max(mu, ||c||_inf)). */
#define OBCA_RESTART_MU 1.0
/* Patience of the passes BEFORE the restart (only while the restart phase is on): a pass that has not converged after this many
   iterations is abandoned for the restart.  Measured (tools/restart_study.py): solves either converge well below
   it -- N = 5: <= 301 iterations, N = 20: <= 389, N = 74: ~500 per pass -- or crawl at an indefinite point with delta_w ~ 1e3
   until max_iter (3000 for obca_mpc4: 0.27 s on one wavefront), nothing in between. */
#define OBCA_PATIENCE(N) (500 + 10 * (N))
#ifndef OBCA_RESTART_MAX_ITER        /* iteration limit of the restart pass: the restarts that succeed take 16-117 iterations at N <= 20 (C3 gated,
                                        C5; tools/restart_study.py), one that does not would otherwise run to max_iter = 3000 */
#define OBCA_RESTART_MAX_ITER(N) (300 + 10 * (N))
#endif
#define OBCA_WINDOW_SPEED_FRAC 0.9

/* Line-search filter capacity: a function of the problem SHAPE only, so that every kernel that can run a shape (and the
   oracles) stops at the same point when the filter fills up (status OBCA_STATUS_NUMERIC): 64 entries for shapes the
   one-wavefront kernel takes (<= 384 rows: one entry per lane), 128 for larger ones.  IPOPT's filter is unbounded; a
   solve that fills 64 entries is stalling at the barrier floor. */
#define OBCA_FILTER_CAP(R_max) ((R_max) <= 384 ? 64 : 128)

#define OBCA_INST_DOUBLES 64   /* LDS reserved for the per-instance constant block (struct Inst) */
#define OBCA_ZK_DOUBLES(N) (36 * (((N) + 1) / 2) + 42)   /* four-wavefront kernels: forward half of the two-sided Riccati sweep */
/* obca_ipm_kernel_mw_r5 (769 .. 1280 rows, 256 threads): the fifth row slot (rows 1024 .. R_max - 1) lives in LDS behind the
   block above: 15 doubles per row + one dummy element (Rows<-5> in obca_kernel.hip) */
#define OBCA_HYB_DOUBLES(R_max) ((R_max) > 768 ? 15 * (((R_max) > 1024 ? (R_max) - 1024 : 0) + 1) + 1 + 1024 : 0)

/* Fused closed loop: consecutive steps of one rollout per work item.  Measured on C5 (4096 rollouts x 30 steps, per-XCD
   queues): 1 / 2 / 3 / 5 / 6 / 10 steps -> 1.112 / 1.101 / 1.083 / 1.060 / 1.050 / 1.141 s: a longer item keeps the rollout's state
   warm in its CU and needs fewer hand-offs, a too long one leaves the end of the launch to a few workgroups. */
#ifndef OBCA_RO_BLOCK
#define OBCA_RO_BLOCK 6
#endif

struct ObcaWeightsDev { double Q[9], P[9], R1[4], R2[4]; };
/* restart: a second start exists (0 / 1); start: where the FIRST pass begins -- 0 the reference's all-zero cold start (the
   second start is the reference window then), 1 the reference window (the second start is the cold start).  obca_params.restart
   carries both (include/obca_mpc.h): OBCA_OPT_RESTART / OBCA_OPT_START decode it the same way everywhere. */
struct ObcaOptsDev { double tol, rho, feas_tol; int32_t max_iter_free, max_iter_fixed, max_soc, restart, start, pad_; };
#define OBCA_OPT_RESTART(r) (((r) < 0 || (r) == 2) ? 0 : 1)
#define OBCA_OPT_START(r) ((r) >= 1 ? 1 : 0)
struct ObcaParamsDev {
    ObcaWeightsDev free_time, fixed_time;
    double xL[2], xU[2], uL[2], uU[2];
    double gego[4], off, dmin;
    ObcaOptsDev opt;
};

struct ObcaLaunch {
    int32_t B, N, nO, M, n_max, R_max, inst_off;
    int32_t two_sided;     /* four-wavefront kernels: Riccati sweep cut in two halves run by two wavefronts (obca_set_two_sided_sweep) */
    int32_t offm[OBCA_MAX_OBST + 1];
    const int32_t* variant;
    const double *x0, *u0, *xref, *A, *b, *Ts, *term;
    double *xopt, *uopt, *ts_opt;
    int32_t *status, *iters;
    double* info;
    double* prof;          /* [B,20] per-phase cycle counters, only written by -DOBCA_PROFILE builds */
    double* warm_z;        /* [B,n_max] primal vector of the last successful solve (in/out) or NULL: obca_set_warm_start */
    const int32_t* warm_use; /* [B] != 0: start from warm_z shifted by one stage; NULL = every instance          */
    double warm_mu;        /* barrier parameter a warm-started solve begins with                                 */
    double* cert_z;        /* [B,n_max] final primal vector, or NULL: obca_set_certificate_buffers                        */
    double* cert_y;        /* [B,R_max + 2 npair] final multipliers (objective units), rows then rotation equalities, or NULL */
    double* soc_ws;        /* [B, n_max + 2 R_max + 2 npair] scratch of the second-order correction (original direction, corrected
                              residuals); NULL switches the correction off (wave kernels; the lane kernel keeps it in its workspace) */
    double* gm_ws;         /* obca_ipm_kernel_gm (shapes beyond the LDS): per-workgroup slice of the HBM workspace, or NULL        */
    int64_t gm_stride;     /* doubles per slice                                                                                  */
    ObcaParamsDev prm;
};


#ifdef __HIPCC__
// Every entry point that allocates or launches runs on ITS handle's device and leaves the caller's current device alone
// (two handles on different GPUs in one process; torch.cuda.set_device after construction).
struct ObcaDeviceGuard {
    int prev;
    bool ok;
    explicit ObcaDeviceGuard(int dev) : prev(-1), ok(true) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~ObcaDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#endif

// internal (not part of the C ABI): the launch descriptor obca_solve_batch builds, for kernels that embed the solver
int obca_internal_fill_launch(obca_handle* h, const int32_t* variant, int32_t B,
                              const double* x0, const double* u0, const double* xref,
                              const double* A, const double* b, const double* Ts, const double* term,
                              const obca_params* p,
                              double* xopt, double* uopt, double* ts_opt, int32_t* status, int32_t* iters,
                              double* info, ObcaLaunch* out, int64_t* lds_bytes, int* wave_ok);

#endif
