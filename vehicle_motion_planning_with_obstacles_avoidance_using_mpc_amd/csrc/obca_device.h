// obca_device.h -- structures shared by the HIP kernel and the host side of libobca_mpc.so.
#ifndef OBCA_DEVICE_H
#define OBCA_DEVICE_H

#include <math.h>
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
/* obca_mpc4 only: a solve that converged with elastic variables left is repeated from the same start with rho x 100 and, if
   elastic variables still remain, with rho x 1000 (the l1 penalty is exact only above the multipliers, and "rho too small"
   looks like "infeasible").  The second level is what the reference's own open-loop plan of demo9 at N = 50 needs (objective
   8e4: every start ends with elastic variables ~3e-3 up to rho = 1e6 and converges at 1e7 to the plan the reference's figure
   shows, tests/golden/reference_openloop_demo9.json); beyond it (1e8) the solves crawl. */
#ifndef OBCA_N_ESCALATIONS
#define OBCA_N_ESCALATIONS 2
#endif
#define OBCA_RHO_ESCALATION(level) ((level) <= 1 ? 100.0 : 1000.0)    /* level 1, 2 */
/* Start ladder (every variant; rule and measurements in oracle/ipm_dense.py:solve, include/obca_mpc.h: start_order): a solve that
   ended without a feasible point is repeated from the next start of the order.  The three starts:
     OBCA_KIND_X0      all variables 0, Topt = 1, every pose at x0 (IPOPT's first Newton iterate from the reference's start)
     OBCA_KIND_WINDOW  poses = xref (first pose x0), inputs by differences clipped to their box, free-time problem: the time
                       scale at which the window is driven at OBCA_WINDOW_SPEED_FRAC of the speed bound -- with a larger initial
                       barrier parameter (OBCA_RESTART_MU)
     OBCA_KIND_ZEROS   the reference's literal start (src/obca.py:856): all variables 0, Topt = 1 */
#define OBCA_KIND_X0 0
#define OBCA_KIND_WINDOW 1
#define OBCA_KIND_ZEROS 2
/* The dodge rung (fixed-time problems, after every start of the order ended without a feasible point; include/obca_mpc.h: dodge).
   Where the reference window runs head-on into an obstacle the l1 penalty problem has a stationary point that is symmetric about
   the window -- the plan brakes in front of the obstacle, the terminal set and the distance rows share the violation -- and all
   three starts of the order end there although a plan around the obstacle exists (first seen on the reference's demo11 run,
   steps 21-25: tests/test_reference_demo11.py).  Two more passes start from the window moved sideways by OBCA_DODGE_OFFSET metres
   (ramped in over the first OBCA_DODGE_RAMP stages; headings along the moved poses, inputs by differences), to the right
   (OBCA_KIND_DODGE_R) and to the left (OBCA_KIND_DODGE_L) of the direction of travel, with lambda / mu of every (stage, obstacle)
   pair set to the separating half-space of the largest gap (lambda = e_j / ||A_j||, mu from the rotation equalities).  Both run;
   the answer is the feasible one with the lower objective. */
#define OBCA_KIND_DODGE_R 3
#define OBCA_KIND_DODGE_L 4
#define OBCA_DODGE_OFFSET 3.0
#define OBCA_DODGE_RAMP 3
/* obca_mpc6: the rung is tried only with at least this much reach beyond the terminal set (metres; obca_terminal_shortfall < -this): a
   detour of lateral offset d over a path of length L is about 2 d^2 / L longer, so below it no offset that clears anything fits -- and the
   closed loop's terminal set (x0 + 5 m, five steps of exactly 1 m) has 1e-9 m to spare while the car runs straight at full speed */
#define OBCA_DODGE_MIN_SPARE 0.1
/* Second level of the rung (round 6), obca_mpc8 only -- the variant with no fallback behind it (where it fails the reference's closed loop
   stops, src/closed_loop.py:401-413): when neither side of the first level ends feasible, the same two starts again with IPOPT's own
   initial barrier parameter OBCA_MU_INIT instead of OBCA_RESTART_MU.  The dodge starts are nearly feasible with active rows; a barrier
   parameter of 1 pushes the first iterates far enough from them to fall back into the basin of the infeasible stationary point.  Seen
   on C5 world 667 (step 20); then measured on 8192 worlds no rule of this build was looked at on: their only two solver failures
   (worlds 7225, 11188) both end feasible with it (profiles/r06_c5_heldout_failures.json).  Passes of the rung: */
#define OBCA_DODGE_LEVEL2_MU OBCA_MU_INIT
#define OBCA_DODGE_PASSES(variant) ((variant) == 8 ? 4 : 2)
/* transient, kernel-internal: a feasible answer of the first dodge pass waiting for the second one (never returned to a caller) */
#define OBCA_STATUS_DODGE_OK 3
#define OBCA_STATUS_DODGE_ACC 4
/* obca_params.start_order = OBCA_START_DEFAULT (0) means: the reference window first for every variant (window -> x0 -> zeros).
   obca_mpc6 / obca_mpc8 (round 4): several local optima -- measured on 2048 gated instances at N = 20 the window start ends lower than
   x0 on 73-76 % of those both solve and solves 95 % alone against 80 % (profiles/r04_start_quality.txt).  obca_mpc4 (round 5): one
   optimum on every workload measured -- from the window all 8192 headline instances, all 8192 free-time instances of C3 and every
   free-time step of the five reference-held runs end where they end from x0 -- in a third of the iterations (17 against 50 on the
   headline batch, 20 against 64 at N = 20).  Two exceptions keep x0 first for every variant: a caller's warm start stands for x0, and
   a single-start call -- its caller has a fallback of its own and is served best by the start that fails fastest (C5, where the
   terminal set makes obca_mpc6 fail on half of the gated steps: 0.546 s with x0 there, 0.648 s with the window). */
#ifndef OBCA_DEFAULT_ORDER_MPC8
#define OBCA_DEFAULT_ORDER_MPC8 1
#endif
#ifndef OBCA_DEFAULT_ORDER_MPC4
#define OBCA_DEFAULT_ORDER_MPC4 1
#endif
#define OBCA_EFFECTIVE_ORDER(o, variant, warm, single) ((o) != 0 ? (o) : (((warm) || (single)) ? 3 : (variant) == 4 ? OBCA_DEFAULT_ORDER_MPC4 : (variant) == 8 ? OBCA_DEFAULT_ORDER_MPC8 : 1))
/* kind of start s = 0, 1, 2 of the EFFECTIVE order o = 1, 2, 3 (two bits per start): window/x0/zeros, zeros/window/x0, x0/window/zeros */
#define OBCA_START_KIND(o, s) (((((o) == 1) ? 0x21 : ((o) == 2) ? 0x06 : 0x24) >> (2 * (s))) & 3)
/* the caller's optional warm start (obca_set_warm_start) takes the place of the first COLD start of the order */
#define OBCA_WARM_KIND(o) ((o) == 2 ? OBCA_KIND_ZEROS : OBCA_KIND_X0)
/* barrier parameter the window start begins with (IPOPT's restoration phase likewise raises mu to max(mu, ||c||_inf)).  Measured
   in round 4 with 0.1 instead: the gated N = 20 launch 128 instead of 194 ms per 2048 and the same reference-held digits, but the
   window start then no longer finds the plan of SURVEY Appendix C's obca_mpc6 witness (demo1_dyn_mpc6 ends feas = False): kept. */
#ifndef OBCA_RESTART_MU
#define OBCA_RESTART_MU 1.0
#endif
/* Iteration limits of the passes while further starts remain (obca_params.patience / retry_iter; <= 0 selects these):
   the first start's passes are abandoned for the next start after OBCA_PATIENCE iterations.  Measured (round 3, git history of tools/restart_study.py;
   DESIGN.md): solves either converge well below it -- N = 5: <= 301 iterations, N = 20: <= 389, N = 74: ~500 per pass -- or
   crawl at an indefinite point with delta_w ~ 1e3 until max_iter (3000 for obca_mpc4: 0.27 s on one wavefront), nothing in
   between.  Later starts: the ones that succeed take 16-117 iterations from the window at N <= 20 (C3 gated, C5). */
#define OBCA_PATIENCE(N) (500 + 10 * (N))
#define OBCA_RETRY_ITER(N) (300 + 10 * (N))
/* One solve = at most three starts x three penalties (obca_mpc4 that converged with elastic variables left: again from the
   same start with the next penalty of the escalation; the next start begins at the base penalty again). */
#define OBCA_MAX_PASSES (3 * (1 + OBCA_N_ESCALATIONS))
#define OBCA_WINDOW_SPEED_FRAC 0.9
/* WHICH pass's answer an exhausted ladder leaves in the caller's buffers -- status, iterate, objective (round 6; until then: the last
   pass's; rule: oracle/ipm_dense.py:_replaces).  A pass that ends at a feasible point always; otherwise the first one, replaced by a
   later one only if that one CONVERGED (status 2: a stationary point of the penalty problem with elastic variables left -- a statement
   about the problem) where the held one did not (iteration limit, line-search failure, filter full: statements about the solver), or
   if it is the same start's repetition with a raised penalty.  The iteration count stays that of the whole sequence. */
#define OBCA_LADDER_REPLACES(st, start, have, held_st, held_start) \
    (!(have) || (st) == OBCA_STATUS_OK || (st) == OBCA_STATUS_ACCEPTABLE || (st) == OBCA_STATUS_BAD_BOUNDS || \
     ((st) == OBCA_STATUS_INFEASIBLE && ((held_st) != OBCA_STATUS_INFEASIBLE || (held_start) == (start))))

/* Line-search filter capacity: a function of the problem SHAPE only, so that every kernel that can run a shape stops at the
   same point when the filter fills up (status OBCA_STATUS_NUMERIC, answered by the next start of the ladder): 64 entries for
   shapes the one-wavefront kernel takes (<= 384 rows: one entry per lane), 128 for larger ones.  IPOPT's filter is unbounded
   and so is the oracles' (oracle/ipm_dense.py, oracle/obca_oracle.c): on a filter-full instance the kernels move to the next
   start where the oracle keeps iterating -- a product deviation, met by ~1 of 8192 C2 instances. */
#define OBCA_FILTER_CAP(R_max) ((R_max) <= 384 ? 64 : 128)

#define OBCA_INST_DOUBLES 64   /* LDS reserved for the per-instance constant block (struct Inst) */
#define OBCA_ZK_DOUBLES(N) (36 * (((N) + 1) / 2) + 42)   /* four-wavefront kernels: forward half of the two-sided Riccati sweep */
/* obca_ipm_kernel_mw_r5 (769 .. 1280 rows, 256 threads): the fifth row slot (rows 1024 .. R_max - 1) lives in LDS behind the
   block above: 15 doubles per row + one dummy element (Rows<-5> in obca_kernel.hip) */
#define OBCA_HYB_DOUBLES(R_max) ((R_max) > 768 ? 15 * (((R_max) > 1024 ? (R_max) - 1024 : 0) + 1) + 1 + 1024 : 0)

/* Fused closed loop: consecutive steps of one rollout per work item.  Measured on C5 (4096 rollouts x 30 steps, per-XCD
   queues): 1 / 2 / 3 / 5 / 6 / 10 steps -> 1.112 / 1.101 / 1.083 / 1.060 / 1.050 / 1.141 s: a longer item keeps the rollout's state
   warm in its CU and needs fewer hand-offs, a too long one leaves the end of the launch to a few workgroups. */
/* XCDs of an MI355X (one L2 each): queues of the fused closed loop (obca_kernel.hip: rollout_fused_body) */
#define OBCA_RO_XCDS 8
#ifndef OBCA_RO_BLOCK
#define OBCA_RO_BLOCK 6
#endif

/* Sizes that follow from the problem shape (N, number of obstacles nO, half-space rows M) and the LDS carve-up of the
   LDS-resident kernels (csrc/obca_kernel.hip: obca_ipm_body): ONE definition for the host (LDS request, kernel choice) and for
   the kernels' compile-time-shape instantiations (OBCA_SHAPES below).  n_max: variables, R_max: elastic rows (the largest
   over the three variants), inst_off: offset of the per-instance constant block, lds_doubles: doubles of dynamic LDS. */
struct ObcaShapeSizes { int n_max, R_max, inst_off; long long lds_doubles; };
constexpr long long obca_even(long long c) { return (c + 1) & ~1ll; }
constexpr ObcaShapeSizes obca_shape_sizes(int N, int nO, int M) {
    const int N1 = N + 1, np = N1 * nO, MW = OBCA_MAX_EDGES + 6;
    const int n_max = N1 * (3 + M + 4 * nO) + 2 * N + 1;
    const int R_max = 3 + 3 * N + 3 + 2 * N1 + 2 * N + 2 * N + 2 + 2 * np + N1 * M + N1 * 4 * nO;
    long long t = 0;
    t += obca_even(n_max) + obca_even(n_max > 120 ? n_max : 120) + obca_even(5 * N1 + 1) + obca_even(n_max);   // x, dx (also FG / Mall / mall), gf (compact), bx
    t += 5 * obca_even(R_max);                                                                               // y, Einv, gh, Lb, Ub
    t += obca_even(3 * N1 + 3);                                                                              // dy of the soft rows
    t += 4 * obca_even(N1) + 2 * obca_even(2 * np);                                                          // ct, st, ctt, stt; cc, cct
    t += 3 * obca_even(2 * np);                                                                              // nu, dnu, crot
    t += obca_even(N1 * M * 2) + obca_even(N1 * M) + obca_even(3 * N1);                                      // Aobs, bobs, xref
    t += obca_even(36 * N1) + obca_even(8 * N1);                                                             // packed stage blocks, gradients
    {
        const long long nx = obca_even(n_max), nr = obca_even(R_max), ny = (long long)MW * 4 * np;
        t += obca_even(ny > nx + nr ? ny : nx + nr);                                                         // Y, shared with xt and tmp
    }
    t += obca_even(36 * N1 > 12 * np ? 36 * N1 : 12 * np) + obca_even(6 * N1) + obca_even(12 * N1) + obca_even(2 * N1) + obca_even(9 * (N1 + 1));   // Pk (also Sloc), qk, Kk, kapk, Mik
    t += obca_even(32) + obca_even(8);                                                                       // lsv, offm
    const int inst_off = (int)t;
    t += OBCA_INST_DOUBLES;
    return ObcaShapeSizes{n_max, R_max, inst_off, t};
}
/* Scratch of the second-order correction: the original direction (n_max + R_max + 2 npair doubles) and the accumulated residuals of
   the rows (R_max).  Where it costs no occupancy it lives in LDS BEHIND everything else the kernel carves -- the one-wavefront kernels
   keep as many workgroups per CU as without it (at most four: one wavefront per SIMD at 512 registers), the four-wavefront kernels
   (one workgroup per CU) take it while the total stays below the CU's 160 KB -- otherwise in HBM (ObcaLaunch.soc_ws, round 1-5: always;
   measured in round 5 on the headline launch: 5.8 MB of the 8.1 MB written per launch were this scratch).  Offsets in doubles from the
   start of the dynamic LDS, 0 = HBM.  One definition for the host and the compile-time-shape instantiations. */
#define OBCA_LDS_CU_BYTES (160 * 1024)
#define OBCA_LDS_STATIC_BYTES 64              /* static LDS of the solver kernels (ladder state, screen verdict): 32 ... 64 B */
constexpr long long obca_soc_doubles(int N, int nO, int M) {
    return obca_even((long long)obca_shape_sizes(N, nO, M).n_max + 2 * (long long)obca_shape_sizes(N, nO, M).R_max + 2 * (long long)(N + 1) * nO);
}
constexpr int obca_wgs_per_cu(long long dyn_bytes, int cap) {
    return (int)(OBCA_LDS_CU_BYTES / (dyn_bytes + OBCA_LDS_STATIC_BYTES)) < cap ? (int)(OBCA_LDS_CU_BYTES / (dyn_bytes + OBCA_LDS_STATIC_BYTES)) : cap;
}
constexpr int obca_soc_lds_wave(int N, int nO, int M) {          /* one-wavefront kernels (also inside the fused closed loop) */
    const long long base = 8 * obca_shape_sizes(N, nO, M).lds_doubles, with = base + 8 * obca_soc_doubles(N, nO, M);
    return (obca_wgs_per_cu(base, 4) > 0 && obca_wgs_per_cu(with, 4) == obca_wgs_per_cu(base, 4)) ? (int)obca_shape_sizes(N, nO, M).lds_doubles : 0;
}
constexpr int obca_soc_lds_mw(int N, int nO, int M) {            /* four-wavefront LDS kernels: behind the two-sided sweep's storage and the fifth row slot */
    const long long off = obca_shape_sizes(N, nO, M).lds_doubles + OBCA_ZK_DOUBLES(N) + OBCA_HYB_DOUBLES(obca_shape_sizes(N, nO, M).R_max);
    return 8 * (off + obca_soc_doubles(N, nO, M)) + OBCA_LDS_STATIC_BYTES <= OBCA_LDS_CU_BYTES ? (int)off : 0;
}
/* Compile-time-shape instantiations of the one-wavefront kernel: X(N, nO, M).  With the shape known to the compiler every LDS
   offset is an immediate, the stage loops and the index arithmetic (divisions by nO, M, 4 nO, the stage stride) fold, and the
   scalar registers that held ~35 array offsets and the layout are free: measured on C2 29.7 -> 26.5 ms per 8192 solves, every
   output word equal to the generic kernel's (tests/test_gpu_shapes.py).  Listed: every (obstacles, rows) combination the
   reference's nine demo settings produce (src/demo_setting.py) -- static obstacles plus the sensed moving rectangles of four rows
   each: demo1-5 (3, 6) (4, 10); demo6-8 (2, 2) (3, 6) (4, 10); demo9 (5, 14) (6, 18) -- at N_free = N_fix = 5 (the GIF's setting,
   BASELINE) and 6 (src/closed_loop.py:66-67 as checked in), as far as the one-wavefront kernel holds them (<= 384 rows); plus
   (5, 14) = three static obstacles and two sensed rectangles (config C5).  Any other shape runs the generic kernels;
   OBCA_SPECIALISE=0 / obca_set_shape_specialisation(h, 0) forces them.  (The fused closed-loop kernel stays generic: with one
   inlined body per group it gained 3.5 % on C5 and moved 33 GB of scratch per launch instead of 1 GB -- DESIGN.md section 9.) */
#define OBCA_SHAPES(X) X(5, 2, 2) X(5, 3, 6) X(5, 4, 10) X(5, 5, 14) X(5, 6, 18) X(6, 2, 2) X(6, 3, 6) X(6, 4, 10) X(6, 5, 14)
/* ... and of the four-wavefront LDS kernels: the two halves of BASELINE.json's configs[2] (N = 20: free-time against the three
   static obstacles, gated fixed-time against five) */
#define OBCA_MW_SHAPES(X) X(20, 3, 6) X(20, 5, 14)

struct ObcaWeightsDev { double Q[9], P[9], R1[4], R2[4]; };
/* order: obca_params.start_order (validated); nstarts: 1 (single_start) or 3; patience / retry_iter: resolved (> 0) */
struct ObcaOptsDev { double tol, rho, feas_tol; int32_t max_iter_free, max_iter_fixed, max_soc, order, nstarts, patience, retry_iter, dodge, screen, pad_; };
/* the start fields of obca_params -> their resolved form; false: start_order / single_start outside its range.  dodge /
   terminal_screen follow the zero-initialisation rule of obca_params: 0 = the default (on), negative = off */
static inline bool obca_resolve_starts(ObcaOptsDev* o, int start_order, int single_start, int patience, int retry_iter, int N, int dodge = 0, int terminal_screen = 0) {
    if (start_order < OBCA_START_DEFAULT || start_order > OBCA_START_X0_FIRST || single_start < 0 || single_start > 1) return false;
    o->order = start_order;
    o->nstarts = single_start ? 1 : 3;
    o->patience = patience > 0 ? patience : OBCA_PATIENCE(N);
    o->retry_iter = retry_iter > 0 ? retry_iter : OBCA_RETRY_ITER(N);
    o->dodge = dodge >= 0 ? 1 : 0;
    o->screen = terminal_screen >= 0 ? 1 : 0;
    o->pad_ = 0;
    return true;
}
/* Closed-form screen of obca_mpc6 (rule and derivation: oracle/ipm_dense.py:terminal_set_shortfall): by how much no trajectory
   the rows allow can reach the terminal set's x_N >= termx -- the heading of the first step is x0's, the speeds are bounded by
   the input box and, from u0, by the acceleration rows; the margin is what elastic variables of size feas_tol on the rows
   involved can add.  > 0: the solve is not run (status OBCA_STATUS_INFEASIBLE, zero iterations, the x0 start as the iterate). */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline double obca_terminal_shortfall(int N, double Ts, double x0x, double c0 /* cos of x0's heading */, double u0v, double uL0, double uU0, double xU0, double termx, double feas_tol) {
    double vhi = u0v, vlo = u0v, reach = 0.0;
    for (int k = 0; k < N; ++k) {
        vhi = fmin(uU0, vhi + OBCA_ACC_MAX0 * Ts);
        vlo = fmax(uL0, vlo - OBCA_ACC_MAX0 * Ts);
        reach += Ts * (k == 0 ? fmax(vhi * c0, vlo * c0) : fmax(fabs(vhi), fabs(vlo)));
    }
    const double xN = fmin(x0x + reach, xU0);
    const double margin = 2.0 * feas_tol * (N + 2 + N * Ts + Ts * Ts * N * (N + 1) / 2.0);
    return termx - xN - margin;
}
struct ObcaParamsDev {
    ObcaWeightsDev free_time, fixed_time;
    double xL[2], xU[2], uL[2], uU[2];
    double gego[4], off, dmin;
    ObcaOptsDev opt;
};

struct ObcaLaunch {
    int32_t B, N, nO, M, n_max, R_max, inst_off;
    int32_t two_sided;     /* four-wavefront kernels: Riccati sweep cut in two halves run by two wavefronts (obca_set_two_sided_sweep) */
    int32_t soc_lds;       /* scratch of the second-order correction in LDS: offset in doubles (obca_soc_lds_wave / _mw), 0 = in HBM (soc_ws) */
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
/* (the descriptor is filled for the one-wavefront kernels: soc_lds = obca_soc_lds_wave, *lds_bytes includes that scratch) */

#endif
