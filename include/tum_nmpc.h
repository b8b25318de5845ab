/*
 * include/tum_nmpc.h -- C-ABI of libtumnmpc.so, the MI355X-native batched SQP-RTI solver.
 *
 * Drop-in boundary: the reference drives its solver through acados_template.AcadosOcpSolver
 * (a ctypes wrapper around the generated libacados_ocp_solver_<name>.so). Every entry point
 * below replaces one AcadosOcpSolver method the reference calls; the reference call sites are
 * cited per function (paths relative to bzarr/TUM-CONTROL). All functions take plain pointers
 * and sizes; no torch / C++ types cross the boundary.
 *
 * New relative to acados: a leading BATCH dimension. A capsule owns `batch` independent OCP
 * instances (perturbed-x0 / sigma-point / Monte-Carlo / weight-sweep fan-outs); one wavefront
 * solves one instance. Host buffers passed to set/get hold `nb` consecutive instances
 * [b0, b0+nb), `stride` doubles apart (stride 0 on a set = broadcast one record to all nb).
 *
 * Conventions: doubles everywhere; matrices column-major like acados; every set/get COPIES
 * (caller owns host buffers, capsule owns device memory); return 0 on success, non-zero on
 * error with the message available from tum_ocp_last_error(). One host thread per capsule.
 */
#ifndef TUM_NMPC_H
#define TUM_NMPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define TUM_NX 8          /* [posx,posy,yaw,vlong,vlat,yawrate,delta_f,a]  pred_model_dynamic_stm_pacejka.py:80-93 */
#define TUM_NU 2          /* [jerk, steering_rate]                                                        :96-98   */
#define TUM_NY 6          /* cost_y_expr = [x0,x1,wrap(yaw),vlong,u]        NMPC_STM_acados_settings.py:51 */
#define TUM_NYE 4
#define TUM_N_MAX 56      /* horizon limit of this build: N <= 40 (five 16-wide MFMA tiles of condensed variables) on every kernel
                             variant, 41..48 (six tiles) on the pipeline variant only (nominal and coupled SNMPC OCP), 49..56 (seven
                             tiles, round 6: Tp = 4.0 s at Ts_MPC = 0.08 s is N = 50) with a diagonal W (nominal, R2 and coupled SNMPC OCP) */
#define TUM_ALL_STAGES (-1)

/* acados return codes the callers test (NMPC_class.py:183-206, main.py:59-61) */
#define TUM_SUCCESS 0
#define TUM_QP_FAILURE 4

typedef struct tum_ocp tum_ocp;           /* opaque capsule (acados: nlp_solver_capsule) */

/* Problem description = what NMPC_STM_acados_settings.py:16-245 bakes into the generated solver. */
typedef struct tum_ocp_desc {
    int N;                 /* shooting intervals, ocp.dims.N                     :34  */
    int nsub;              /* sim_method_num_steps (ERK4 sub-steps)              :240 */
    double dt;             /* Tf / N                                             :230 */
    int batch;             /* number of independent OCP instances                     */
    int device;            /* HIP device ordinal                                      */
    /* single-track / Pacejka constants, Config/EDGAR/veh_params_pred.yaml, pacejka_params.yaml,
       pred_model_dynamic_stm_pacejka.py:33-46 */
    double lf, lr, m, Iz, ro, S, Cd;
    double Bf, Cf, Df, Ef, Br, Cr, Dr, Er;
    double g, fr0, fr1, fr4;
    double acc_min;        /* braking limit used as ax_max when a < 0            NMPC_STM_acados_settings.py:74 */
    /* velocity-dependent gg limits, Config/EDGAR/ggv.csv via NMPC_class.py:322-335 */
    int n_ggv;
    double ggv_v[16], ggv_ax[16], ggv_ay[16];
    /* QP solver options (acados: qp_solver_iter_max :232, HPIPM tolerances json:904-950) */
    int qp_iter_max;
    double qp_tol_stat, qp_tol_ineq, qp_tol_comp;
    double qp_mu0;         /* initial complementarity target of the interior point method (default 0.05) */
    double qp_t0;          /* floor of the initial constraint residuals (default 0.05) */
    int store_qp_in;       /* keep A_k,B_k,b_k of the last linearisation for tum_ocp_get_from_qp_in */
    /* acados: qp_solver_warm_start (1 in Stochastic_NMPC/SNMPC_acados_settings.py:307, unset = 0 for the nominal solver). != 0: in a
     * sequence of solves the interior point method of an instance starts from the multipliers and violation slacks its previous QP
     * ended with -- when that QP converged and the new problem is close to it (at most 16 row sides changed their activity, no new
     * violation above 0.1; otherwise it starts cold); tum_ocp_cold_start / tum_ocp_reset forget them -- pushed back into the interior and
     * re-centred to the complementarity target qp_warm_mu (0: the default 1e-2). The QP solution is the same to the solver's
     * tolerances; 9 % fewer interior point iterations over the reference's logged closed loops (profiles/r05_ipm_iterations.txt).
     * The Python binding switches it on by default for every controller -- for the nominal and R2 solvers a documented deviation from
     * the reference's setting ("Interior point warm start"). */
    int qp_warm_start;
    double qp_warm_mu;
    /* the proximity gate of the warm start: an instance starts warm only when at most qp_warm_flips row sides changed their activity
     * against the previous QP (0: the default 16; < 0: no gate -- a development aid, measured to stall solves at the iteration cap on
     * jumping sequences, profiles/r05_warm_gate.txt) and no new violation exceeds qp_warm_viol (0: the default 0.1). */
    int qp_warm_flips;
    double qp_warm_viol;
} tum_ocp_desc;

/* AcadosOcpSolver(ocp, json_file=..., generate=..., build=...)   NMPC_STM_acados_settings.py:243 */
tum_ocp *tum_ocp_create(const tum_ocp_desc *desc);
void tum_ocp_free(tum_ocp *c);
const char *tum_ocp_last_error(void);
int tum_ocp_batch(const tum_ocp *c);
int tum_ocp_horizon(const tum_ocp *c);

/* acados_solver.set(stage, field, value)   NMPC_class.py:172,178,254; SNMPC_class.py:124,130
 * field: "x" (8), "u" (2), "yref" (6 for stage<N, 4 for stage N); on an SNMPC capsule also
 * "p" (L*ns + 2 values = [A_pce.flatten(), risk_parameter, stop_flag], SNMPC_class.py:124,185,193; b0 = 0, nb = batch,
 * stride 0: the parameter vector is shared by the batch). A_pce and the risk parameter are shared by all stages (the
 * reference sends every stage the same values; a differing risk parameter is an error at the next solve); the stop flags
 * must be 0 on the stages < uph and 1 from stage uph on (SNMPC_class.py:103-104) and DEFINE the uncertainty propagation
 * horizon of the next solve (any other pattern is an error at the next solve).
 * stage == TUM_ALL_STAGES: v holds all stages back to back ("x": (N+1)*8, "u": N*2, "yref": (N+1)*6
 * with the terminal record padded to 6). */
int tum_ocp_set(tum_ocp *c, int stage, const char *field, const double *v, int len, int b0, int nb, int stride);
/* acados_solver.get(stage, field)          NMPC_class.py:193,198; Reduced_Robustified_NMPC_class.py:280,298
 * field: "x", "u" (and the soft-constraint slacks of the last QP: "sl", "su", 3 per stage in acados order). */
int tum_ocp_get(tum_ocp *c, int stage, const char *field, double *v, int len, int b0, int nb, int stride);

/* acados_solver.constraints_set(stage, field, value)   NMPC_class.py:111-112,245-246;
 * Reduced_Robustified_NMPC_class.py:335-336,359,362-365
 * stage 0 "lbx"/"ubx" (8 values) = initial state x0 (initial-value embedding);
 * stage>=1 "lbx"/"ubx" (1 value) = steering-angle bound; "lbu"/"ubu" (1) = steering-rate bound;
 * "lh"/"uh" (1) = bounds of the gg-circle constraint. */
int tum_ocp_constraints_set(tum_ocp *c, int stage, const char *field, const double *v, int len, int b0, int nb, int stride);

/* acados_solver.cost_set(stage, field, value)   NMPC_class.py:295-317
 * "W": ny*ny (stage<N: 36, stage N: 16) column-major. Any matrix, as in acados (its symmetric part is what the cost sees and what is kept). The
 * reference only installs blockdiag(Q,R) with diagonal Q, R: a capsule whose W have always been diagonal stores the diagonal and condenses with the
 * kernels of the headline; the first W with an off-diagonal entry switches the capsule to a full 6 x 6 per stage (nominal / R2 OCP on the pipeline;
 * the condensing then runs as the six-wavefront kernel's full-W instantiation whatever the batch size; refused by the coupled SNMPC OCP, whose cost
 * rows come from the prologue kernel, and by the development build's fused kernel). Per STAGE, as in acados: the reference sets every stage in a loop (NMPC_class.py:294-296), and a caller may
 * give every stage its own weights. stage == TUM_ALL_STAGES with a 6 x 6 W: all the stages 0..N-1 in one call. (The development
 * build's fused kernel reads stage 0's W for all stages < N; the shipped pipeline honours every stage.)
 * "zl","zu","Zl","Zu": 1 value at stage 0 [sbu], 3 at stages 1..N-1 [sbu,sbx,sh], 2 at stage N [sbx,sh]. */
int tum_ocp_cost_set(tum_ocp *c, int stage, const char *field, const double *v, int len, int b0, int nb, int stride);

/* status = acados_solver.solve()   NMPC_class.py:183, SNMPC_class.py:198, Reduced_Robustified_NMPC_class.py:264
 * One SQP real-time iteration for every instance. Returns the max status over the batch
 * (0 ok, 4 QP failure); per-instance values via tum_ocp_get_stats("status"). Synchronous.
 * The reference's LITERAL per-step call sequence on a small capsule (batch x (N+1) x 8 <= 32768 doubles, N <= 63) -- N+1 x
 * set(j,"yref"), constraints_set(0,"lbx"|"ubx"), solve(), get(0,"u"), N x get(j,"x"), get_cost(), 3 x get_stats, NMPC_class.py:169-206,
 * 243-246 -- costs ONE device round trip: the per-stage "yref" setters and the stage-0 "lbx"/"ubx" land in a pinned shadow the capsule
 * owns (with a map of the stages touched) and go up in one kernel in front of the solve; the solve ends with one kernel that writes
 * u0 / cost / status / qp_iter and the whole iterate into pinned slabs; get "x" / "u", get_cost and get_stats "qp_iter" / "status" are
 * then host copies until something changes the iterate (a set "x"/"u", reset, cold start, device upload, another solve). Same numbers as
 * the one-call step (tum_ocp_step_async) and as the uncached getters, bit for bit (tests/test_gpu_call_sequence.py). On such a capsule
 * the solve is timed by the device's wall clock (get_stats "time_tot") and the events around the interior point kernel are left out
 * (get_stats "time_ipm" needs tum_ocp_set_kernel(c, "time-ipm")). */
int tum_ocp_solve(tum_ocp *c);
/* Asynchronous flavour for benchmarking / pipelining: enqueue on the capsule's stream, no host sync. */
int tum_ocp_solve_async(tum_ocp *c);
int tum_ocp_synchronize(tum_ocp *c);

/* acados_solver.get_cost()   NMPC_class.py:202 */
int tum_ocp_get_cost(tum_ocp *c, double *out, int b0, int nb);
/* acados_solver.get_stats(field)   NMPC_class.py:203-205
 * "time_tot" -> 1 double, device seconds of the last solve (HIP events on the capsule's stream)
 * "sqp_iter" -> nb ints (always 1), "qp_iter" -> nb ints, "status" -> nb ints, "qp_status" -> nb ints
 * "res" -> nb x 3 doubles (stat, ineq, comp residuals of the last QP). */
int tum_ocp_get_stats(tum_ocp *c, const char *field, void *out, int b0, int nb);
/* acados_solver.reset()   NMPC_class.py:251 -- zero the iterate of every instance */
int tum_ocp_reset(tum_ocp *c);
/* acados_solver.get_from_qp_in(stage, "A")   Reduced_Robustified_NMPC_class.py:295
 * "A" (64, column-major 8x8), "B" (16, column-major 8x2), "b" (8) of the LAST linearisation. */
int tum_ocp_get_from_qp_in(tum_ocp *c, int stage, const char *field, double *out, int len, int b0, int nb, int stride);

/* Device-side plumbing (no reference counterpart): use an external HIP stream (e.g. torch's current
 * stream) and copy results device-to-device into caller-owned HBM (for the RCCL gather). */
int tum_ocp_set_stream(tum_ocp *c, void *hip_stream);
/* field: "u0" (nb x 2), "x1" (nb x 8), "cost" (nb), "X" (nb x (N+1)*8), "U" (nb x N*2),
 * "status" / "qp_iter" (nb int32), "summary" (nb x 5 doubles: u0[2], cost, status, qp_iter -- the slab of the rooted gather) */
int tum_ocp_get_device(tum_ocp *c, const char *field, void *dev_dst, int b0, int nb);
/* Results on the HOST without stalling the stream (no reference counterpart; the caller it serves reads u0 / pred_X / cost /
 * status after every solve, NMPC_class.py:193-206). tum_ocp_results_async enqueues, on the capsule's stream and behind the
 * solve already enqueued there: the result summary packed on the device (5 doubles per instance: u0[2], cost, status,
 * qp_iter) and its copy into a PINNED host slab the capsule owns -- with_iterate != 0: also the whole iterate X ((N+1)*8
 * doubles per instance) and U (N*2) -- then records an event. It returns at once: a caller that keeps several capsules in
 * flight (streaming.SolverRing) enqueues the next batches while this one's results cross PCIe.
 * A capsule owns TWO sets of slabs and events, used in turn: up to two requests may be outstanding, so a caller can enqueue the
 * next batch and its request BEFORE it reads the previous batch's results (the stream then never runs dry waiting for the host).
 * tum_ocp_results_wait blocks until the event of the OLDEST outstanding request has passed and hands out its pinned slabs
 * (valid until the second-next tum_ocp_results_async on this capsule; X / U are null when that request was made without the
 * iterate). A third request without a wait in between is an error. */
int tum_ocp_results_async(tum_ocp *c, int with_iterate);
int tum_ocp_results_wait(tum_ocp *c, const double **summary, const double **X, const double **U);
/* number of result requests outstanding on this capsule (0, 1 or 2; tum_ocp_step_async makes one): a synchronous caller drains what
 * an earlier, abandoned request left behind before it trusts tum_ocp_results_wait to deliver ITS solve (solver.py: step) */
int tum_ocp_results_outstanding(const tum_ocp *c);
/* One control step of a HOST-driven loop in one call (what NMPC_class.py:163-241 does with 2 + (N+1) setters, solve() and
 * 2 N + 5 getters, each a synchronous round trip): x0 (nb = batch instances x 8; SNMPC capsules: as tum_ocp_put_device "x0") and
 * yref (batch x (N+1) x 6) -- either may be null: the capsule keeps what it has -- are copied into PINNED staging memory the capsule
 * owns and uploaded on its stream, one SQP-RTI is enqueued behind them and a results request (tum_ocp_results_async) behind
 * the solve. Returns after enqueuing; tum_ocp_results_wait delivers summary / X / U. The caller's x0 / yref buffers are free on
 * return. Two steps may be outstanding, like two result requests. Small batches: one kernel reads the staging area, one writes
 * summary and iterate into the pinned slabs, and the step's device time (get_stats "time_tot") is read from the device's wall clock
 * by these two kernels -- no copy command and no event on the stream; get_stats "time_ipm" is not available after a step. One instance, N = 38, warm: 0.34 -> 0.15 ms per control step
 * of the mirrored controller class. */
int tum_ocp_step_async(tum_ocp *c, const double *x0, const double *yref, int with_iterate);
/* The other direction: per-instance inputs from caller-owned DEVICE memory (asynchronous D2D on the capsule's stream).
 * field: "x0" (nb x 8, = constraints_set(0,"lbx")), "yref" (nb x (N+1)*6), "X" (nb x (N+1)*8), "U" (nb x N*2). */
int tum_ocp_put_device(tum_ocp *c, const char *field, const void *dev_src, int b0, int nb);
/* Zero-copy flavour of the above for "x0" / "yref" (whole batch): the capsule USES the caller's device array as its own -- kernels read
 * it in place, setters and the device closed loop write through to it -- until dev_ptr = NULL hands the capsule's own array back. The
 * memory must stay valid, and unchanged by others, while solves that use it are in flight. For callers that rotate batches resident in
 * HBM (bench.py --bind-inputs). Measured (profiles/r05_bind_inputs.txt): the copy kernels of tum_ocp_put_device cost 1.2 % on one stream
 * and disappear behind another capsule's solve with three capsules in flight (4.037 against 4.032 M solves/s). */
int tum_ocp_bind_device(tum_ocp *c, const char *field, void *dev_ptr);
/* cold start every instance on the device: X_k = x0 for all k, U = 0 (acados create / reset + set x;
 * NMPC_class.py:250-254) using the x0 already uploaded with constraints_set(0,"lbx"). */
int tum_ocp_cold_start(tum_ocp *c);
/* Scheduling of the instances onto wavefronts (no reference counterpart). longest_first = 1 (default): workgroup i solves
 * the instance with the i-th largest IPM iteration count of the PREVIOUS solve, so the few long instances of a batch start
 * first instead of leaving the GPU idle at the end (a batch is only a few rounds of resident wavefronts); 0: natural order.
 * Results do not depend on the schedule. Environment override at create time: TUM_NMPC_SCHEDULE=natural. */
int tum_ocp_set_schedule(tum_ocp *c, int longest_first);
/* Kernel variant of the solve (no reference counterpart). The shipped library has ONE: "pipeline" (= "auto", the default):
 * linearise / condense / interior point / expand as four kernels, each at its own occupancy, handing over through an
 * L2-resident workspace; the coupled SNMPC OCP runs its prologue / epilogue kernels around it. get_stats("time_ipm") reports
 * the interior point kernel. The development build (libtumnmpc_dev.so, tests and experiments only) adds the two other
 * implementations the pipeline is held against: "fused" (round 1's single kernel, N <= 40, uph <= 31) and "pipeline4" (the
 * pipeline with the four-wavefront interior point kernel); the shipped library refuses these names. Environment override at
 * create time: TUM_NMPC_KERNEL.
 * Two further names choose the PROLOGUE of a coupled SNMPC capsule without touching the rest: "prologue-mfma" (the column
 * recursions of the samples as v_mfma_f64_4x4x4_4b products; n_samples <= 10, where it is the library's own choice) and
 * "prologue-passes" (the column-slot / pass kernels of rounds 1-3, the only ones for n_samples > 10): two implementations of
 * the same hand-over the tests hold against each other.
 * Small batches are bound by the length of ONE pass of a kernel, not by throughput; for them the library launches wider forms of
 * the first two kernels of the pipeline, and four more names pin the choice (tests, A/B runs):
 *   "lin-eight-lanes" / "lin-lane-per-stage"      linearisation with eight lanes per (instance, stage) -- one sensitivity column per
 *       lane, the tyre chains of the model split over a DPP quad -- or one; default: eight while batch x (N+1) x 8 lanes are one
 *       round of wavefronts (batch <= 199 at N = 40). b_k identical, A_k / B_k to 3e-15 relative (FMA contraction).
 *   "cond-six-wavefronts" / "cond-one-wavefront"  condensing with a workgroup of six (N > 40: seven) wavefronts per OCP -- column
 *       recursion with four lanes per column, Hessian tiles dealt to four wavefronts, gradient on two -- or one wavefront;
 *       default: the workgroup while batch <= 256 (one per CU). Bit-identical results. The nominal OCP only.
 *   "loop-fork" / "loop-serial"   device closed loops (tum_sim_run) of the nominal OCP on the latency path: the linearisation of a
 *       control step's solve runs BESIDE the planner of that step, on a side stream of the simulation (the Runge-Kutta pass needs
 *       the iterate, not the reference; the four residuals of the cost are formed by the condensing kernel) -- or behind it as in
 *       a plain solve. Default: serial (the fork is bit-identical and measured slower: the cross-stream dependencies cost more
 *       than the overlap gains). Environment: TUM_SIM_FORK = 0 | 1.
 * "auto" hands all these choices back to the library. Environment overrides (read once): TUM_LIN_COLS, TUM_COND_WIDE = 0 | 1.
 * Two names are timing options, not kernels: "time-ipm" keeps the two HIP events around the interior point kernel on EVERY solve
 * (get_stats "time_ipm"), also where the library leaves them out -- steps, and synchronous solves of small capsules: every event on the
 * stream is a gap of several microseconds between two kernels; "no-time-ipm" (default) hands that back. */
int tum_ocp_set_kernel(tum_ocp *c, const char *name);
/* last kernel launch time in milliseconds (HIP events on the launch stream) */
double tum_ocp_last_kernel_ms(tum_ocp *c);
/* debug: one solve, then the condensed QP of instance b as the condensing kernel handed it to the interior point kernel
 * (N <= 40): out = [H 80x80 | q 80 | steering-angle row and gg row of every stage, 2N x 80 | their constants 2N]; see
 * csrc/tum_nmpc.hip */
int tum_ocp_debug_dump(tum_ocp *c, int b, double *out, int len);

/* ---- the small kernels either side of the solve ---------------------------------
 * Scenario fan-out of the initial state (Stochastic_NMPC/stochastic_mpc_utils.py:78-91 compute_x0dist as a
 * batch axis): batch must equal P*(S+1); instance p*(S+1) is pose p, instance p*(S+1)+s is pose p + offs[s-1].
 * pose: P x 8, offs: S x 8 (host). Writes lbx_0 = ubx_0 of every instance. */
int tum_ocp_set_x0_fanout(tum_ocp *c, const double *pose, const double *offs, int P, int S);
/* PCE moments over each scenario group (Stochastic_NMPC/SNMPC_acados_settings.py:116-133): coefficients
 * c = A v with the L x S least-squares PCE matrix A (row-major, host), mean = c_0, var = sum_{k>=1} c_k^2, for
 * every component of field "x" (8) / "u" (2) at `stage` of the current iterate. mean, var: P x m (host). */
int tum_pce_moments(tum_ocp *c, const char *field, int stage, const double *A, int L, int S, double *mean, double *var);
/* The same reduction without a host round trip: tum_pce_attach keeps A (L x S, row-major, host) on the device,
 * tum_pce_moments_device enqueues the reduction on the capsule's stream and writes P x m doubles each into caller-owned
 * DEVICE buffers (the slab of the rooted gather of BASELINE config 3). */
int tum_pce_attach(tum_ocp *c, const double *A, int L, int S);
int tum_pce_moments_device(tum_ocp *c, const char *field, int stage, double *mean_dev, double *var_dev);
/* The coupled SNMPC OCP (Stochastic_NMPC/SNMPC_acados_settings.py:19-320 with the DISCRETE stacked dynamics
 * of Stochastic_NMPC/pred_model_dynamic_disc.py:121-220): turns the capsule (created with nsub = 1) into the solver the
 * reference builds at SNMPC_acados_settings.py:318 and calls at SNMPC_class.py:198. The stacked state is the nominal
 * copy followed by `ns` sample copies (nx = 8 (ns+1)); Apce (L x ns, row-major, host) and `uph` replace the per-stage
 * parameter vector p = [A_pce.flatten(), risk_parameter, stop_flag] (SNMPC_class.py:103-104,124: stop_flag = 1 from
 * stage uph on); gamma is the chance-constraint level (kappa = sqrt((1-gamma)/gamma), SNMPC_acados_settings.py:187).
 * Afterwards set/get "x" and constraints_set "lbx"/"ubx" at stage 0 also accept 8 (ns+1) values, cold_start copies the
 * stacked x0 to every stage (SNMPC_class.py:126-127), and solve runs prologue + fused kernel + epilogue. The cost acts
 * on the nominal copy with |v| as the speed row; the gg limits are looked up at |v|. 0 <= uph <= N (the reference ran
 * uph = N; beyond 31 stages the sample columns no longer fit one wavefront: pipeline kernels only). 1 <= ns <= 32 and
 * 1 <= L <= 32 (MPC_params.yaml's n_samples / expansion_degree, SNMPC_class.py:78-94; beyond 16 of either: pipeline kernels
 * only, and ns x uph bounded by the prologue's 128 KiB of LDS -- uph <= 28 / 26 / 18 / 16 at 17 / 20 / 24 / 32 samples --
 * otherwise refused here). */
int tum_ocp_snmpc_attach(tum_ocp *c, int ns, int L, const double *Apce, int uph, double gamma);
int tum_ocp_snmpc_samples(const tum_ocp *c);   /* ns, or 0 for a nominal capsule */
/* Offsets of the sample initial conditions from the nominal one (compute_x0dist, Stochastic_NMPC/stochastic_mpc_utils.py:78-91;
 * SNMPC_class.py:259-264 recomputes x0_samples from x0 before every solve): ns x 8 (host). Once registered, an 8-value
 * lbx_0 / ubx_0 -- and the device closed loop (tum_sim_*), whose state estimator writes the nominal x0 -- fans out to the
 * sample copies on the device at every solve; a stacked 8 (ns+1) lbx_0 switches back to explicit samples. */
int tum_ocp_snmpc_set_offsets(tum_ocp *c, const double *offs);
/* R2NMPC constraint tightening after a solve (Reduced_Robustified_NMPC_class.py:286-366): propagates
 * Sigma_{k+1} = A_k Sigma_k A_k' + B W B' with the A_k of the last linearisation (needs store_qp_in) and rewrites the
 * capsule's lbx/ubx (steering angle) and uh (gg circle) of stages 1..N-1 for the NEXT solve.
 * Sigma0, BWB: 8x8 row-major (host); backoff (optional, host): batch x N x 2 (steering, gg) back-offs. */
int tum_ocp_r2_backoff(tum_ocp *c, const double *Sigma0, const double *BWB, int uph,
                       double delta_min, double delta_max, double uh_nom, double *backoff);
/* The same tightening as part of EVERY solve (Reduced_Robustified_NMPC_class.py:276-378 runs it after each successful acados
 * call, and counts it in the reported solver time, :379-381): after the SQP-RTI kernel the back-off kernel rewrites the bounds
 * for the next solve on the capsule's stream; instances whose solve failed keep their bounds. Needs store_qp_in. This is what
 * lets the robustified controller run in the device closed loop (tum_sim_*). uph = 0 detaches. */
int tum_ocp_r2_attach(tum_ocp *c, const double *Sigma0, const double *BWB, int uph,
                      double delta_min, double delta_max, double uh_nom);
/* Snapshot / restore (asynchronous) of ALL per-stage bounds on the device: the tightening rewrites them after every solve,
 * a sweep or a benchmark that restarts from the nominal problem puts them back without a host copy. */
int tum_ocp_bounds_snapshot(tum_ocp *c);
int tum_ocp_bounds_restore(tum_ocp *c);
/* read back a bound installed with constraints_set / r2_backoff (one value per instance) */
int tum_ocp_constraints_get(tum_ocp *c, int stage, const char *field, double *v, int b0, int nb);

/* ---- the producer and the consumer of the solve, on the device --------------------------
 * PlannerEmulator(ref_traj, pose, n_points, Tp, loop_circuit)   Utils/MPC_sim_utils.py:137-194, called from
 * main.py:52-54 / get_baseline_performances.py:105: stateless batched form. track: n_track x 4 row-major
 * [pos_x,pos_y,ref_yaw,ref_v]; pose: P x 2; ref_out: P x n_points x 4; closest_out (optional): P. All host pointers. */
int tum_planner_emulate(const double *track, int n_track, const double *pose, int P, int n_points, double Tp,
                        int loop_circuit, double *ref_out, int *closest_out, int device);

/* A batch of closed loops kept in HBM: planner -> solve -> plant step + state estimation
 * (main.py:48-78; Utils/SimulationMode_main_class.py:106-156 sim_step simMode 0 / StateEstimation;
 * Vehicle_Simulator/sim_model_dynamic_stm_pacejka.py:137-195 + VehicleSimulator.py:73-77: RK4 with n_elem elements over Ts).
 * windows: the 8 moving-average lengths (1..4). log_capacity: control steps of logs kept on the device (0 = none). */
typedef struct tum_sim tum_sim;
tum_sim *tum_sim_create(tum_ocp *c, const double *track, int n_track, double Tp, int loop_circuit, double Ts, int n_elem,
                        const int *windows, int log_capacity);
void tum_sim_free(tum_sim *s);
/* x_sim: batch x 7 plant states, x_mpc: batch x 8 controller states (host); resets the estimator and the step counter */
int tum_sim_set_state(tum_sim *s, const double *x_sim, const double *x_mpc, int cold_start);
/* Disturbance realisation played back by the loop (Utils/SimulationMode_main_class.py:121-143: simulate_disturbances /
 * simulate_state_estimation, as with disturbance_playback; the draws of Utils/MPC_sim_utils.py:54-86 are made on the host): w_deriv is
 * added to the state derivatives of the plant for the whole control step (Vehicle_Simulator/sim_model_dynamic_stm_pacejka.py:196) in a
 * SECOND plant step from the same state -- the true state stays the undisturbed step, as in the reference --, e_est is added to that
 * disturbed state before the estimator. n_steps x batch x 7 doubles each (host, step-major), either may be null; control steps beyond
 * n_steps run undisturbed; n_steps = 0 removes the realisation. tum_sim_set_state restarts the playback. */
int tum_sim_set_disturbances(tum_sim *s, const double *w_deriv, const double *e_est, int n_steps);
int tum_sim_plan(tum_sim *s);        /* yref of every instance from its pose (async on the capsule's stream) */
/* plant step with (x1[7], u0[1]) of the iterate, estimator -> next x0 (async). An instance whose solve FAILED (status != 0)
 * is then treated as main.py:59-61 treats it (MPC.reintialize_solver(x_next)): its iterate is cold-started at the state the
 * failed solve started from (sample copies included), an R2 capsule gets its nominal bounds back. */
int tum_sim_advance(tum_sim *s);
/* nsteps x (plan, solve, advance), then synchronises; chunks of 25 steps are captured once into a hipGraph and replayed
 * (tum_sim_get "graph_steps" tells whether the capture succeeded) */
int tum_sim_run(tum_sim *s, int nsteps);
int tum_sim_steps(const tum_sim *s);
/* field: "x_sim" (B*7), "x_mpc" (B*8), "pose" (B*2), "ref0" (B*4), "closest" (B), "graph_steps" (1); logs with the npz schema of
 * Utils/Logging_Plotting.py:357-372, step-major: "CiLX" ((steps+1)*B*7), "MPC_SimX" ((steps+1)*B*8), "simU" (steps*B*2),
 * "simREF" (steps*B*4), "simSolverDebug" (steps*B*5: cost, 0, sqp_iter, qp_iter, status). len must match. */
int tum_sim_get(tum_sim *s, const char *field, double *out, long long len);

/* development aid: one solve with in-kernel phase timers; out = batch x 12 shader-cycle counters
 * [linearise, condense, ipm-residuals, M assembly, Cholesky, rhs, tri-solves, row updates, (iteration tail), expand+cost] */
int tum_ocp_profile_phases(tum_ocp *c, long long *out);

#ifdef __cplusplus
}
#endif
#endif /* TUM_NMPC_H */
