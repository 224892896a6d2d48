/*
 * obca_mpc.h -- C ABI of the MI355X batched OBCA-MPC solver (libobca_mpc.so).
 *
 * Drop-in boundary.  The reference has no FFI: the hot path is entered through plain Python method
 * calls on `obca()` (reference src/closed_loop.py:22, call sites :117-120, :130-140, :381-398).  Each
 * entry point below therefore names the reference interface it stands behind:
 *
 *   obca_create / obca_destroy   construction of the solver object   (src/closed_loop.py:22  `obca()`)
 *   obca_solve_batch             obca.obca_mpc4 / obca_mpc6 / obca_mpc8 (src/obca.py:828, :1361, :1564),
 *                                B independent calls at once
 *   obca_strerror                the reference never raises across this boundary (bare `except:`,
 *                                src/obca.py:1062-1065); errors are return codes / per-instance status
 *
 * All array arguments are DEVICE pointers (HBM resident, e.g. torch `data_ptr()`), row-major, fp64
 * unless noted; the caller owns every buffer.  Calls are asynchronous on the given HIP stream.
 * No C++ exception crosses this boundary.
 *
 * Threading: handles are independent; calls on ONE handle must be issued from one thread at a time and, when the
 * lane-per-instance kernel runs (it owns a workspace inside the handle), must be ordered on one stream.  Different
 * handles may be used concurrently from different threads and streams.
 */
#ifndef OBCA_MPC_H
#define OBCA_MPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OBCA_MAX_OBST 8      /* obstacles per instance                               */
#define OBCA_MAX_EDGES 4     /* half-spaces per obstacle (vObs[i]-1 in the reference) */

/* problem shape shared by every instance of a handle (reference arguments N, nObs, vObs) */
typedef struct obca_dims {
    int32_t N;                         /* horizon (reference argument N)                          */
    int32_t n_obs;                     /* obstacles passed to the solver (reference nObs)         */
    int32_t m[OBCA_MAX_OBST];          /* half-spaces of obstacle i = vObs[i]-1                   */
    int32_t max_batch;                 /* largest B later passed to obca_solve_batch              */
    int32_t device;                    /* HIP device ordinal                                      */
} obca_dims;

/* cost weights of one call family (reference arguments P, Q, R=[R1,R2]) */
typedef struct obca_weights {
    double Q[9], P[9], R1[4], R2[4];
} obca_weights;

/* everything else the reference passes per call and keeps constant over a rollout */
typedef struct obca_params {
    uint32_t struct_size;              /* sizeof(obca_params) of the header the CALLER was built against: set by
                                          obca_params_init() (or by hand after zero-initialising).  Anything else -- a caller built
                                          against another layout, a struct that was never initialised -- is answered with
                                          OBCA_E_INVAL instead of being read field by field as something it is not          */
    uint32_t reserved_;                /* 0                                                        */
    obca_weights free_time;            /* used by variant 4 (closed_loop.py:77-81)                */
    obca_weights fixed_time;           /* used by variants 6 and 8 (closed_loop.py:94-98)         */
    double xL[2], xU[2];               /* position box (theta is unbounded, obca.py:916)          */
    double uL[2], uU[2];               /* input box                                               */
    double ego[4];                     /* car footprint (closed_loop.py:63)                       */
    double dmin;                       /* clearance (closed_loop.py:64)                           */
    /* interior-point options; <= 0 selects the default in brackets.  The struct MUST be zero-initialised before its fields
       are set (memset / = {0}): every option added later reads 0 as "default". */
    double tol;                        /* [1e-8]  IPOPT tol                                        */
    double rho;                        /* [1e4]   elastic (l1) penalty, unscaled objective units; a free-time solve
                                                   that ends with elastic variables left is repeated from the same
                                                   start with rho x 100 and, if they still remain, with rho x 1000
                                                   (exact-penalty escalation; the next start begins at rho again)  */
    double feas_tol;                   /* [1e-6]  largest elastic variable still called feasible   */
    int32_t max_iter_free;             /* [3000]  IPOPT default, variant 4: bounds EACH pass of a solve (see `patience`) */
    int32_t max_iter_fixed;            /* [1000]  obca.py:1538, variants 6/8: likewise              */
    int32_t max_soc;                   /* [4]     IPOPT max_soc: second-order-correction trials after a rejected first
                                                   trial step; 0 = the default, negative = off              */
    /* The starts of a solve ("start ladder"; rule and measurements: oracle/ipm_dense.py:solve).  A solve
       that ends without a feasible point -- status 2, -1, -2, -3 -- is repeated from the next start of the order, inside the
       same launch, until one start ends feasible or the order is exhausted; iteration and factorisation counts returned are
       those of the whole sequence; status, iterate and info[0..2] are those of the pass that ended feasible.  After an exhausted
       ladder (obca_mpc 0.6; before: the last pass's, whatever it was) they are those of the most informative pass: the FIRST
       one, replaced by a later one only if that one converged to a stationary point of the penalty problem with elastic
       variables left (status 2 -- a statement about the problem) where the held one did not (-1, -2, -3 -- statements about the
       solver), or if it is the same start's repetition with a raised penalty.  So status 2 after an exhausted ladder means
       "some start converged and found no feasible point", -1 / -2 / -3 that none converged.  The three starts:
         x0      every variable 0, Topt = 1 as the reference (src/obca.py:856), every pose at x0 -- the iterate IPOPT's first
                 Newton step reaches from the reference's all-zero start (the initial condition and the dynamics linearised
                 at v = 0 read x_k = x0); uses nothing but x0, like the reference's cold start
         window  the poses of the reference window xref (first pose x0), inputs by differences clipped to their box
         zeros   the reference's literal start: every variable 0, Topt = 1
       No order makes a problem infeasible that another order solves: all three starts are tried in every order. */
    int32_t start_order;               /* [OBCA_START_DEFAULT] one of the OBCA_START_* constants below; anything else:
                                                   OBCA_E_INVAL                                             */
    int32_t single_start;              /* [0]     1 = only the first start of the order (with its penalty escalation) -- what a
                                                   driver asks for where its own fallback follows, as obca_mpc8 follows a
                                                   failed obca_mpc6 in the closed loop (src/closed_loop.py:393-398);
                                                   other values than 0 / 1: OBCA_E_INVAL                    */
    int32_t patience;                  /* [500 + 10 N]  while further starts remain, the FIRST start's passes are abandoned
                                                   for the next start after this many iterations (solves converge far below
                                                   it or crawl until max_iter); never above max_iter_*; with
                                                   single_start = 1 only max_iter_* applies                 */
    int32_t retry_iter;                /* [300 + 10 N]  iteration limit of every later start's passes; never above max_iter_* */
    int32_t dodge;                     /* [on]    0 = the default (on), negative = off.  The last rung of the ladder, obca_mpc6 /
                                                   obca_mpc8 only, after every start of the order (the one start with
                                                   single_start = 1) ended without a feasible point: where the reference window
                                                   runs head-on into an obstacle the penalty problem has a stationary point that is
                                                   symmetric about the window (the plan brakes in front of the obstacle) and all
                                                   three starts end there although a plan around the obstacle exists -- steps 21-25
                                                   of the reference's own demo11 run, where IPOPT drives around.  Two more passes
                                                   start from the window moved 3 m to the right and to the left of the direction of
                                                   travel (ramped in over three stages; lambda, mu on the separating half-space);
                                                   both run, the feasible answer with the lower objective is returned; their
                                                   iterations are added to `iters`.  obca_mpc8 only (obca_mpc 0.6; its failure has no
                                                   fallback behind it): if neither side ends feasible, the same two starts once more
                                                   with IPOPT's own mu_init 0.1 instead of 1 (csrc/obca_device.h: OBCA_DODGE_LEVEL2_MU).  A failed rung leaves the answer the order's
                                                   starts left (see above).                                    */
    int32_t terminal_screen;           /* [on]    0 = the default (on), negative = off.  obca_mpc6 whose terminal set
                                                   x_N >= term[0] no trajectory can reach -- the first step's heading is x0's, the
                                                   speeds are bounded by uL / uU and, from u0, by the acceleration rows; margin for
                                                   elastic variables of size feas_tol on the rows involved -- is not run: status
                                                   OBCA_STATUS_INFEASIBLE, iters 0, xopt = x0 at every stage, uopt = 0,
                                                   info = (0, shortfall in metres, 0, 0).  The reference's closed loop asks for
                                                   x0 + 5 m in N_fix steps of exactly 1 m at full speed, so after the first dodge
                                                   four of five failing obca_mpc6 calls are of this kind (tests/test_terminal_screen.py) */
} obca_params;

/* zero-fills *p and sets struct_size: the one way to start filling an obca_params */
void obca_params_init(obca_params* p);

/* obca_params.start_order */
enum {
    OBCA_START_DEFAULT = 0,            /* window -> x0 -> zeros for every variant.  obca_mpc6 / obca_mpc8 have several local optima, and from the
                                          window the solver ends at the lower one far more often; obca_mpc4 has one optimum on every workload
                                          measured -- all 8192 headline instances, the free-time half of config C3, every free-time step of the
                                          five reference-held runs end where they end from x0 -- and reaches it from the window in a third of the
                                          iterations (csrc/obca_device.h: OBCA_EFFECTIVE_ORDER; x0 first was the obca_mpc4 default of obca_mpc 0.4).
                                          Two exceptions keep x0 first for every variant: single_start = 1 (a caller with its own fallback wants the
                                          start that fails fastest: the closed loop's obca_mpc6 before obca_mpc8) and obca_set_warm_start (the stored
                                          plan stands for x0).  A caller whose reference window is no trajectory (start and goal only: the open-loop
                                          plan) is served better by OBCA_START_X0_FIRST: the x0 start then also runs under `patience` instead of the smaller
                                          `retry_iter` (demo9 at N >= 66 needs 600-1300 iterations from x0) -- the Python mirror's open-loop plan passes it
                                          per call (closedLoop.mpc_openLoop_freeTime). */
    OBCA_START_WINDOW_FIRST = 1,       /* window -> x0 -> zeros, also for single-start and warm-started calls */
    OBCA_START_ZEROS_FIRST = 2,        /* zeros -> window -> x0: the reference's literal start first (the default of
                                          obca_mpc 0.1)                                            */
    OBCA_START_X0_FIRST = 3            /* x0 -> window -> zeros for every variant (the default of obca_mpc 0.2 / 0.3; obca_mpc4's until 0.4) */
};


typedef struct obca_handle obca_handle;

/* per-instance status written by obca_solve_batch */
enum {
    OBCA_STATUS_OK = 0,                /* converged to tol                                        */
    OBCA_STATUS_ACCEPTABLE = 1,        /* IPOPT "acceptable" termination                          */
    OBCA_STATUS_INFEASIBLE = 2,        /* converged but elastic variables remain (feas = False)   */
    OBCA_STATUS_MAXITER = -1,
    OBCA_STATUS_LINESEARCH = -2,
    OBCA_STATUS_NUMERIC = -3,
    OBCA_STATUS_BAD_BOUNDS = -4,
    OBCA_STATUS_SKIPPED = -5,          /* variant[b] == 0: instance not solved, outputs untouched */
    OBCA_STATUS_BAD_VARIANT = -6       /* variant[b] not in {0,4,6,8}, or 6 with term == NULL: not solved */
};

/* return codes */
enum {
    OBCA_OK = 0,
    OBCA_E_INVAL = -22,                /* bad argument / shape beyond compiled limits             */
    OBCA_E_NOMEM = -12,
    OBCA_E_HIP = -5,                   /* a HIP runtime call failed                               */
    OBCA_E_LDS = -28                   /* shape does not fit the LDS kernel (mode 1 only)         */
};

int obca_create(const obca_dims* dims, obca_handle** out);
void obca_destroy(obca_handle* h);

/*
 * Solve B independent NLPs.  variant[b] in {4, 6, 8} selects obca_mpc4 / obca_mpc6 / obca_mpc8;
 * variant[b] == 0 skips instance b (status OBCA_STATUS_SKIPPED, outputs untouched) so that a device-side
 * driver can mask instances without a host round trip.
 *   x0    [B,3]          current pose                     (reference x0)
 *   u0    [B,2]          previous input                   (reference u0)
 *   xref  [B,3,N+1]      reference window                 (reference xref[:, :N+1])
 *   A     [B,N+1,M,2]    obstacle rows per horizon step   (reference AObs; variant 4 reads step 0 only,
 *   b     [B,N+1,M]       obca.py:969; M = sum m[i])      (reference bObs)
 *   Ts    [B]            base sample time                 (reference Ts)
 *   term  [B,3]          xmin, ymin, ymax of the terminal set, variant 6 only (obca.py:1465-1466); may be NULL when
 *                        no instance is variant 6 (a variant-6 instance then gets OBCA_STATUS_BAD_VARIANT)
 * outputs
 *   xopt  [B,3,N+1], uopt [B,2,N], ts_opt [B] (= Topt*Ts for variant 4, Ts otherwise)
 *   status [B] int32, iters [B] int32
 *   info  [B,4] or NULL: objective value, largest elastic variable, final optimality error,
 *                        number of KKT factorisations
 */
int obca_solve_batch(obca_handle* h, const int32_t* variant, int32_t B,
                     const double* x0, const double* u0, const double* xref,
                     const double* A, const double* b, const double* Ts, const double* term,
                     const obca_params* params,
                     double* xopt, double* uopt, double* ts_opt, int32_t* status, int32_t* iters,
                     double* info, void* hip_stream);

/* Optional warm start -- NOT reference behaviour (every solve of the reference is a cold start from zeros, T = 1,
 * src/obca.py:856), off by default, for receding-horizon callers that re-solve a problem one step later.
 * z: device buffer [max_batch, obca_primal_size(dims)] owned by the caller.  While set, every solve that ends
 * converged/acceptable stores its primal vector (poses, inputs, lambda, mu, time scale) there, and every solve of an
 * instance b with use[b] != 0 (use == NULL: all) starts from the stored vector moved one horizon stage forward (last
 * stage repeated) with barrier parameter mu_init instead of 0.1.  Which local optimum is found may differ from the cold
 * start's.  z == NULL switches it off. */
int64_t obca_primal_size(const obca_dims* dims);
int obca_set_warm_start(obca_handle* h, double* z, const int32_t* use, double mu_init);

/* Optional certificate output -- NOT part of the reference's call surface (its callers only read sol.value(x), sol.value(u),
 * src/obca.py:1057-1059), for KKT certificates of the ORIGINAL NLP at the returned point.  While set, every solve stores
 *   z [max_batch, obca_primal_size(dims)]  its final primal vector: per stage k the pose (3), the input (2, k < N),
 *                                          lambda_k (M) and mu_k (4 n_obs); the time scale Topt last (variant 4)
 *   y [max_batch, obca_dual_size(dims)]    the multipliers of the NLP's constraint rows in the objective's own units
 *                                          (the solver's internal objective scaling undone), sign convention
 *                                          grad f + sum_r y_r grad g_r = 0, rows in the order
 *                                            x_0 == x0 (3), dynamics (3N), [x_N == xref_N (3): variant 4],
 *                                            position box (2(N+1)), input box (2N), acceleration rows (2N),
 *                                            [Topt > 0, Topt bounds (2; each stands for the N+1 tied copies): variant 4],
 *                                            [terminal set x, y (2): variant 6], ||A'lambda||^2 <= 1 (per stage and obstacle),
 *                                            distance rows (per stage and obstacle), lambda >= 0 ((N+1) M), mu >= 0 ((N+1) 4 n_obs),
 *                                          followed by the rotation equalities (2 per stage and obstacle).
 * Either pointer may be NULL; both NULL switches it off. */
int64_t obca_dual_size(const obca_dims* dims);
int obca_set_certificate_buffers(obca_handle* h, double* z, double* y);

/* Kernel selection: 0 = auto (default; also env OBCA_MODE): one wavefront per instance when its rows fit the
 * wavefront's registers (<= 384 rows) and its working set one CU's LDS; beyond that, shapes with at most three obstacles run one
 * wavefront per instance with the row state in an HBM workspace owned by the handle (measured 6-33 % faster than four wavefronts
 * there: the stage-serial sweep dominates and four times as many instances are in flight), shapes with more obstacles four
 * wavefronts per instance while the working set still fits the LDS (<= 1280 rows, e.g. N = 20 with five obstacles); else four
 * wavefronts per instance with the row state and every O(rows) array in the HBM workspace and only the O(N) blocks of the
 * stage-serial Riccati sweep in LDS (long horizons: N = 74 with five obstacles has 3976 rows); else (N > ~150) the
 * lane-per-instance kernel.  The choice is a function of the shape only.
 * 1 = one wavefront per instance; 2 = lane-per-instance (64 instances per wavefront, working set in an HBM workspace
 * owned by the handle; any shape); 3 = four wavefronts per instance, LDS resident; 4 = four wavefronts per instance, HBM
 * workspace; 5 = one wavefront per instance, HBM workspace.  Returns OBCA_E_LDS if mode 1 / 3 / 4 / 5 cannot hold the shape. */
int obca_set_mode(obca_handle* h, int mode);

/* Compile-time-shape instantiations.  For the problem shapes the reference's closed-loop driver produces with its nine demo
 * settings (N = 5 or 6; static obstacles plus sensed moving rectangles: (obstacles, rows) = (2, 2) (3, 6) (4, 10) (5, 14) (6, 18))
 * and for the two halves of the N = 20 benchmark configuration (list: csrc/obca_device.h OBCA_SHAPES, OBCA_MW_SHAPES) the library
 * holds an instantiation of the kernel with the shape as compile-time constants -- same code, same arithmetic, every output word
 * equal (tests/test_gpu_shapes.py), 5-15 % shorter launches.  obca_solve_batch uses it whenever the handle's shape is one of them
 * and the kernel selection above would run the corresponding generic kernel; on = 0
 * (environment at obca_create: OBCA_SPECIALISE=0) forces the generic kernel.  obca_shape_is_specialised: 1 if launches of this
 * handle use an instantiation. */
int obca_set_shape_specialisation(obca_handle* h, int on);
int obca_shape_is_specialised(const obca_handle* h);

/* Four-wavefront kernels (OBCA_MODE 3, and auto mode for shapes whose rows do not fit one wavefront's registers): the
 * Riccati sweep over the stages can be cut at stage ~0.45 N into a backward half (cost-to-go) and a forward half
 * (cost-to-arrive) that two wavefronts run at the same time, meeting in one 6 x 6 solve.  Same Newton step up to
 * roundoff (measured in the kernel: 1e-11..1e-10 of the step's size typically); the one-sided sweep rounds exactly like
 * the one-wavefront kernels.  on = -1 (default): two-sided exactly where the one-wavefront kernels cannot run the
 * shape, so that every shape both kernel families can run gives bit-identical results in both; 0: never; 1: always.
 * Environment override at obca_create: OBCA_TWO_SIDED=-1|0|1. */
int obca_set_two_sided_sweep(obca_handle* h, int on);

/* Diagnostic: device buffer [max_batch,20] receiving per-phase shader-clock totals of each instance.
 * Only builds compiled with -DOBCA_PROFILE write to it; NULL (the default) disables it. */
void obca_set_profile_buffer(obca_handle* h, double* prof);

/* bytes of LDS one instance needs in the wave-per-instance kernels (> 163840: only the lane kernel runs it); the
 * four-wavefront kernels ask for 8 * (36 * ((N + 1) / 2) + 42) bytes more (forward half of their two-sided Riccati sweep),
 * beyond 768 rows another 8 * (15 * (max(rows - 1024, 0) + 1) + 1025) (fifth row slot and row values, csrc/obca_device.h) */
int64_t obca_lds_bytes(const obca_dims* dims);

/* ------------------------------------------------------------------------------------------------------
 * Device-resident closed loop: B receding-horizon rollouts advanced in lock-step without leaving the GPU.
 *
 * Stands behind the body of the reference's `closedLoop.closed_loop_mpc4` loop (src/closed_loop.py:345-432):
 * update_obstacle (:445-486), sensor (:591-629), update_reference_trajectory (:502-528), the fixed-time
 * reference preparation (:360-374 with update_path(allAviable=1) :570-587), rebuild_lObs + obstacle_H_Represent
 * for the moving rectangles (src/demo_setting.py:457-473, src/model_obstacle.py:37-102), the variant dispatch
 * with the mpc6 -> mpc8 fallback (:380-398) and the state advance (:400-432).  Quirks kept: q7 (Ts overwritten
 * after a fixed-time step), q8 (vertex lists of present obstacles are not filtered by the lidar gate), cold start
 * every solve, stop after max_steps (30) steps.  N_free = N, N_fix = N_fix (default: equal, the reference's 6/6).
 *
 * Static obstacles are passed as their half-space rows (host side: obstacle_H_Represent); moving obstacles as the
 * reference's 11-tuple [cx, cy, theta, length, width, speed, end_x, end_y, end_theta, t_start, t_end] followed by
 * cos(theta), sin(theta) as the host evaluated them (13 doubles), so that the exact `==` branch tests of
 * obstacle_H_Represent see the same numbers as the reference.
 */
#define OBCA_MAX_DYN 4

typedef struct obca_rollout_dims {
    int32_t N;                         /* horizon of both the free-time and the fixed-time problem */
    int32_t n_static;                  /* static obstacles                                          */
    int32_t m_static[OBCA_MAX_OBST];   /* their half-space counts                                   */
    int32_t n_dyn;                     /* moving rectangles per rollout, 0..OBCA_MAX_DYN             */
    int32_t path_max;                  /* padded length of the reference path                       */
    int32_t batch;                     /* rollouts                                                  */
    int32_t max_steps;                 /* 30 in the reference (src/closed_loop.py:431)              */
    int32_t device;
    int32_t N_fix;                     /* horizon of the fixed-time problem (reference N_fix); 0 = N.  Must be a multiple of
                                          N (the reference resamples the reference plan by int(N_fix/N_free),
                                          src/closed_loop.py:570-587) with N_fix - 5 <= N (its shift-in of the previous plan,
                                          :363-364, reads N_fix - 5 + 1 columns of a free-time plan; the reference raises
                                          IndexError beyond that, e.g. at 6/12)                       */
} obca_rollout_dims;

typedef struct obca_rollouts obca_rollouts;

/* rollout flags */
enum { OBCA_RUN = 0, OBCA_DONE_GOAL = 1, OBCA_DONE_CAP = 2, OBCA_DONE_FAILED = 3 };

int obca_rollouts_create(const obca_rollout_dims* dims, obca_rollouts** out);
void obca_rollouts_destroy(obca_rollouts* r);

/* (Re)start all rollouts.  Device pointers: start [B,3], goal [B,2], path [B,3,path_max] (A* reference,
 * row-major x/y/yaw), path_len [B] int32, static_A [B,Ms,2], static_b [B,Ms], dyn [B,n_dyn,13].
 * Ts0 = the reference's self.Ts (0.1), sense_dis = setting.senseDis (10). */
int obca_rollouts_reset(obca_rollouts* r, const double* start, const double* goal, const double* path,
                        const int32_t* path_len, const double* static_A, const double* static_b, const double* dyn,
                        double Ts0, double sense_dis, const obca_params* params, void* hip_stream);

/* One iteration of the loop body for every rollout still running: harness kernel, one solve launch per problem
 * shape (variant 4 on the static obstacles; variant 6, then 8 where 6 failed, per number of sensed obstacles),
 * state advance.  Asynchronous; no host synchronisation inside. */
int obca_rollouts_step(obca_rollouts* r, void* hip_stream);

/* n_steps iterations for every rollout.  Default (mode 0): when every problem shape fits the wave kernel, ONE launch
 * of a persistent kernel (harness on lane 0, solves on the wave) whose workgroups -- one per SIMD -- take (round, rollout)
 * items (a round = six consecutive steps) from a device-side queue: every rollout has done round r before any starts
 * round r + 1, rollouts advance independently instead of in lock step (one expensive solve does not hold the batch back),
 * and the launch does not end with a few long rollouts on an otherwise idle GPU.  There is one queue per XCD (rollout b
 * belongs to queue b % 8 and is only handled by workgroups running on that XCD, so its state is handed on inside the XCD's
 * L2 without a write-back), followed by one pass of a global queue that skips every item already done.  Results are
 * identical to n_steps calls of obca_rollouts_step.  OBCA_ROLLOUT_QUEUE at obca_rollouts_create: 2 (default) as described,
 * 1 the global queue only (hand-offs through HBM with agent-scope release / acquire), 0 one workgroup per rollout for all
 * its steps.  Mode 1 forces the lock-step launches. */
int obca_rollouts_run(obca_rollouts* r, int32_t n_steps, void* hip_stream);
int obca_rollouts_set_mode(obca_rollouts* r, int mode);
/* The queue mode obca_rollouts_run uses (2, 1 or 0 as above).  2 is only offered where the device reports eight XCCs
 * (hipDeviceAttributeNumberOfXccs; MI355X): the per-XCD queues identify an L2 by HW_REG_XCC_ID & 7.  Elsewhere the default is 1. */
int obca_rollouts_queue_mode(const obca_rollouts* r);
/* Diagnostic (-DOBCA_RO_STATS builds): per persistent workgroup [wait, work (10 ns units), items, end clock], n <= 16384 ints to host */
int obca_rollouts_debug_stats(obca_rollouts* r, int32_t* out, int n);
/* Test hook: runs ONLY the harness part of a step (obstacle advance, lidar gate, reference window, fixed-time preparation,
 * half-space rows of the moving rectangles -- everything before the solve) for every rollout, after setting its step counter
 * to k, its inherited step length to Ts_opt and (x0_host != NULL) its pose to x0_host[3]; then copies what the harness handed
 * the solver of group g (= number of sensed moving obstacles) to HOST buffers: variant [B] (0: the rollout is not in this
 * group), A [B,N_g+1,M_g,2], b [B,N_g+1,M_g] with M_g = static rows + 4 g, N_g = N (g = 0) or N_fix.  Synchronises.  The
 * rollout state is left as the harness left it (moving obstacles advanced): obca_rollouts_reset before running on. */
int obca_rollouts_debug_harness(obca_rollouts* r, int32_t k, double Ts_opt, const double* x0_host, int32_t g,
                                int32_t* variant, double* A, double* b, void* hip_stream);

/* Optional, NOT reference behaviour (see obca_set_warm_start): a step whose problem shape equals the previous step's
 * starts from the previous plan moved one stage forward with barrier parameter mu_init.  Call before
 * obca_rollouts_reset; enable = 0 restores the reference's cold starts. */
int obca_rollouts_set_warm_start(obca_rollouts* r, int enable, double mu_init);

/* Copy state and history to caller-owned DEVICE buffers (any may be NULL): x_closed [B,max_steps+1,3],
 * u_closed [B,max_steps,2], T_closed [B,max_steps], x_openloop [B,max_steps,3,max(N,N_fix)+1] (a free-time
 * plan fills its first N+1 columns), variant_hist [B,max_steps]
 * int32 (4/6/8 as solved, 0 = no step), iters_hist [B,max_steps] int32, status_hist [B,max_steps] int32 (solver
 * status of the step's last solve), dyn_hist [B,max_steps,n_dyn,4]
 * (cx, cy, present, sensed), steps [B] int32 (successful steps), flags [B] int32. */
int obca_rollouts_read(obca_rollouts* r, double* x_closed, double* u_closed, double* T_closed, double* x_openloop,
                       int32_t* variant_hist, int32_t* iters_hist, int32_t* status_hist, double* dyn_hist, int32_t* steps,
                       int32_t* flags, void* hip_stream);

/* ------------------------------------------------------------------------------------------------------
 * Batched global planner: the reference's grid A* (src/a_star.py:16-102: 8-connected, Euclidean cost and
 * heuristic, open list ordered by (f, (row, col)), its neighbour order and re-queue rules) followed by
 * rebuild_path (:137-147) and create_reference_path (:189-200), one rollout per GPU lane.
 *   grid  [B,rows,cols] uint8, 1 = occupied (setting.org_gridMap);  start, goal [B,2] int32 as (row, col)
 *         = (pose_y, pose_x) like the reference's call (src/closed_loop.py:28-30);  rows*cols <= 65535
 *   yaw9  HOST pointer, 9 doubles: arctan2(dy, dx) for dy, dx in {-1,0,1} at index (dy+1)*3+(dx+1), as the
 *         caller's libm evaluates them (the reference uses numpy's)
 *   path  [B,3,path_max] x / y / yaw rows, padded with the last point; path_len [B]: points, or
 *         -1 no route, -2 open list overflow, -3 path_max too small
 *   workspace: device buffer of obca_astar_workspace_bytes(B, rows, cols) bytes.
 * path / path_len are what obca_rollouts_reset takes. */
int64_t obca_astar_workspace_bytes(int32_t B, int32_t rows, int32_t cols);
int obca_astar_batch(const uint8_t* grid, int32_t B, int32_t rows, int32_t cols, const int32_t* start,
                     const int32_t* goal, const double* yaw9, int32_t path_max, double* path, int32_t* path_len,
                     void* workspace, int64_t workspace_bytes, void* hip_stream);

/* Occupancy grids of B worlds on the device -- the reference's mapModel.shape2grid (src/model_map.py:21-56; vertex
 * re-ordering :88-101 and the division by the resolution :58-71 included): boxes [B,K,4] = (xmin, ymin, xmax, ymax) of each
 * obstacle polygon in world units (an entry with xmin > xmax or NaN is padding), grid [B,rows,cols] uint8 (1 = occupied),
 * rows = int((map_y - 1)/resolution) + 1, cols likewise (src/model_map.py:17).  The grid is what obca_astar_batch takes. */
int obca_rasterise_batch(const double* boxes, int32_t B, int32_t K, double resolution, int32_t rows, int32_t cols,
                         uint8_t* grid, void* hip_stream);

const char* obca_strerror(int code);
/* "obca_mpc 0.6 (gfx950)": 0.6 = the answer of an exhausted start ladder is the most informative pass's (obca_params: the starts of a
 * solve), the second-order correction's scratch in LDS where it costs no occupancy; 0.5 = obca_params.struct_size (first member; obca_params_init), the dodge rung and the terminal-set screen,
 * OBCA_START_DEFAULT = the window first for obca_mpc4 too, kernel mode 5;
 * 0.2 = the start ladder (start_order / single_start / patience / retry_iter replace restart); 0.3 = second
 * level of the penalty escalation, compile-time-shape instantiations, obca_rollouts_queue_mode; 0.4 = OBCA_START_DEFAULT per variant
 * (OBCA_START_X0_FIRST moved from 0 to 3) */
const char* obca_version(void);

#ifdef __cplusplus
}
#endif
#endif
