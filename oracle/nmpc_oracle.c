/*
 * oracle/nmpc_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked by the product).
 *
 * Plain-C FP64 restatement of ONE acados SQP-RTI step of the TUM-CONTROL nominal
 * NMPC (single-track + Pacejka), i.e. of what `acados_solver.solve()` computes at
 *   Model_Predictive_Controller/Nominal_NMPC/NMPC_class.py:183
 * for the OCP defined in
 *   Model_Predictive_Controller/Nominal_NMPC/NMPC_STM_acados_settings.py:16-245
 * with the dynamics of
 *   Prediction_Models/pred_model_dynamic_stm_pacejka.py:118-177.
 *
 * The arithmetic itself lives in un-vendored third-party code (acados + BLASFEO +
 * HPIPM, version unpinned by the reference; casadi==3.5.5 for AD/codegen), so this
 * file restates the PUBLISHED algorithm of that stack:
 *   ERK4 x num_steps with forward sensitivities  -> (Phi, A, B)
 *   NONLINEAR_LS cost + GAUSS_NEWTON Hessian      -> diagonal stage Hessians
 *   BGH constraints, two-sided soft               -> linearised rows + slack penalties
 *   FULL_CONDENSING                               -> dense QP in dU (x0 eliminated)
 *   dense primal-dual interior point (Mehrotra)   -> dU, slacks
 *   full step (FIXED_STEP, step length 1)
 * Parity is pinned against the reference's logged acados outputs
 * (Learning_To_Adapt/SafeRL_WMPC/_baseline/F/...npz, see tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Everything here is deliberately the dumbest dense formulation (no tiles, no
 * structure exploitation): it is the checker, the HIP solver is the product.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <malloc.h>

#define NX 8
#define NU 2
#define NMAXH 64                 /* max horizon */
#define NVMAX (NU * NMAXH)       /* condensed variables */
#define MMAX  (3 * NMAXH)        /* constraint rows: N box + N delta + N h */

/* ------------------------------------------------------------------ model */
typedef struct {
    /* Config/EDGAR/veh_params_pred.yaml:3-10, pacejka_params.yaml:3-12 */
    double lf, lr, m, Iz, ro, S, Cd;
    double Bf, Cf, Df, Ef, Br, Cr, Dr, Er;
    double g, fr0, fr1, fr4;          /* pred_model_dynamic_stm_pacejka.py:38-46 */
    double acc_min;                   /* veh_params: acc_min (negative) */
    int    n_ggv;                     /* Config/EDGAR/ggv.csv */
    double ggv_v[16], ggv_ax[16], ggv_ay[16];
} stm_model;

/* forward-mode dual numbers over the 5 "active" states (vl, vt, r, delta, a) */
#define ND 5
typedef struct { double v; double d[ND]; } dual;

static dual dc(double c) { dual r; r.v = c; for (int i = 0; i < ND; i++) r.d[i] = 0.0; return r; }
static dual dvar(double v, int i) { dual r = dc(v); r.d[i] = 1.0; return r; }
static dual dadd(dual a, dual b) { dual r; r.v = a.v + b.v; for (int i = 0; i < ND; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
static dual dsub(dual a, dual b) { dual r; r.v = a.v - b.v; for (int i = 0; i < ND; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
static dual dmul(dual a, dual b) { dual r; r.v = a.v * b.v; for (int i = 0; i < ND; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static dual ddiv(dual a, dual b) { dual r; r.v = a.v / b.v; for (int i = 0; i < ND; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
static dual dscale(dual a, double s) { dual r; r.v = a.v * s; for (int i = 0; i < ND; i++) r.d[i] = a.d[i] * s; return r; }
static dual dchain(dual a, double fv, double fp) { dual r; r.v = fv; for (int i = 0; i < ND; i++) r.d[i] = fp * a.d[i]; return r; }
static dual dsin(dual a) { return dchain(a, sin(a.v), cos(a.v)); }
static dual dcos(dual a) { return dchain(a, cos(a.v), -sin(a.v)); }
static dual datan(dual a) { return dchain(a, atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
static dual dsqrt(dual a) { double s = sqrt(a.v); return dchain(a, s, 0.5 / s); }
static dual dasin(dual a) { return dchain(a, asin(a.v), 1.0 / sqrt(1.0 - a.v * a.v)); }
/* fmax(fmin(a, hi), lo) with CasADi's convention: derivative of the taken branch */
static dual dclip(dual a, double lo, double hi) { if (a.v > hi) return dc(hi); if (a.v < lo) return dc(lo); return a; }

/* xdot = f(x,u) and its Jacobians. Jx is 8x8 row-major, Ju 8x2 row-major.
 * pred_model_dynamic_stm_pacejka.py:118-177 */
static void stm_f_jac(const stm_model *p, const double *x, const double *u,
                      double *xdot, double *Jx, double *Ju)
{
    const double psi = x[2];
    dual vl = dvar(x[3], 0), vt = dvar(x[4], 1), r = dvar(x[5], 2), de = dvar(x[6], 3), a = dvar(x[7], 4);

    /* rolling resistance :118-124 */
    dual vkmh = dscale(dsqrt(dadd(dmul(vl, vl), dmul(vt, vt))), 3.6);
    dual w = dscale(vkmh, 0.01);
    dual w2 = dmul(w, w);
    dual fr = dadd(dadd(dc(p->fr0), dscale(w, p->fr1)), dscale(dmul(w2, w2), p->fr4));
    const double Fz_f = p->m * p->lr * p->g / (p->lf + p->lr);
    const double Fz_r = p->m * p->lf * p->g / (p->lf + p->lr);
    /* longitudinal forces :133-142 (banking = 0, F_braking = 0) */
    dual Fx_f = dscale(fr, -Fz_f);
    dual Fx_r = dsub(dscale(a, p->m), dscale(fr, Fz_r));
    dual Faero = dscale(dmul(vl, vl), 0.5 * p->ro * p->S * p->Cd);
    /* slip angles :148-149 */
    dual al_f, al_r;
    if (x[3] > 0.001) {
        al_f = dsub(de, datan(ddiv(dadd(vt, dscale(r, p->lf)), vl)));
        al_r = datan(ddiv(dsub(dscale(r, p->lr), vt), vl));
    } else {
        al_f = dc(0.0); al_r = dc(0.0);
    }
    /* Pacejka :154-155 */
    dual bf = dscale(al_f, p->Bf), br = dscale(al_r, p->Br);
    dual Fy_f_lat = dscale(dsin(dscale(datan(dsub(bf, dscale(dsub(bf, datan(bf)), p->Ef))), p->Cf)), p->Df);
    dual Fy_r_lat = dscale(dsin(dscale(datan(dsub(br, dscale(dsub(br, datan(br)), p->Er))), p->Cr)), p->Dr);
    /* combined slip :158-163 */
    const double Fmax_f = sqrt(Fz_f * Fz_f + (p->Cf * Fz_f) * (p->Cf * Fz_f));
    const double Fmax_r = sqrt(Fz_r * Fz_r + (p->Cr * Fz_r) * (p->Cr * Fz_r));
    dual Gy_f = dclip(dscale(Fx_f, 1.0 / Fmax_f), -0.98, 0.98);
    dual Gy_r = dclip(dscale(Fx_r, 1.0 / Fmax_r), -0.98, 0.98);
    dual Fy_f = dmul(Fy_f_lat, dcos(dasin(Gy_f)));
    dual Fy_r = dmul(Fy_r_lat, dcos(dasin(Gy_r)));
    /* state derivatives :167-177 */
    dual sd = dsin(de), cd = dcos(de);
    dual vld = dadd(dscale(dadd(dsub(dsub(Fx_r, Faero), dmul(Fy_f, sd)), dmul(Fx_f, cd)), 1.0 / p->m), dmul(vt, r));
    dual front = dadd(dmul(Fy_f, cd), dmul(Fx_f, sd));
    dual vtd = dsub(dscale(dadd(Fy_r, front), 1.0 / p->m), dmul(vl, r));
    dual rd = dscale(dsub(dscale(front, p->lf), dscale(Fy_r, p->lr)), 1.0 / p->Iz);

    const double c = cos(psi), s = sin(psi);
    xdot[0] = x[3] * c - x[4] * s;
    xdot[1] = x[3] * s + x[4] * c;
    xdot[2] = x[5];
    xdot[3] = vld.v; xdot[4] = vtd.v; xdot[5] = rd.v;
    xdot[6] = u[1];
    xdot[7] = u[0];
    if (Jx) {
        memset(Jx, 0, sizeof(double) * NX * NX);
        Jx[0 * NX + 2] = -x[3] * s - x[4] * c; Jx[0 * NX + 3] = c; Jx[0 * NX + 4] = -s;
        Jx[1 * NX + 2] = x[3] * c - x[4] * s;  Jx[1 * NX + 3] = s; Jx[1 * NX + 4] = c;
        Jx[2 * NX + 5] = 1.0;
        for (int j = 0; j < ND; j++) {
            Jx[3 * NX + 3 + j] = vld.d[j];
            Jx[4 * NX + 3 + j] = vtd.d[j];
            Jx[5 * NX + 3 + j] = rd.d[j];
        }
    }
    if (Ju) {
        memset(Ju, 0, sizeof(double) * NX * NU);
        Ju[6 * NU + 1] = 1.0;   /* delta_f_dot = steering_rate */
        Ju[7 * NU + 0] = 1.0;   /* a_dot = jerk */
    }
}

/* exported for tests: plain f and Jacobians */
void oracle_stm_f(const stm_model *p, const double *x, const double *u, double *xdot, double *Jx, double *Ju)
{
    stm_f_jac(p, x, u, xdot, Jx, Ju);
}

/* One shooting interval: ERK4, nsub steps of h = dt/nsub, with forward sensitivities
 * propagated through the same scheme (acados sim_erk with sens_forw; options at
 * NMPC_STM_acados_settings.py:238-240). A = dPhi/dx (8x8), B = dPhi/du (8x2). */
void oracle_rk4_sens(const stm_model *p, const double *x0, const double *u, double dt, int nsub,
                     double *xn, double *A, double *B)
{
    double x[NX], S[NX * (NX + NU)];         /* S = [dx/dx0 | dx/du], row-major 8x10 */
    const int NS = NX + NU;
    memcpy(x, x0, sizeof(x));
    memset(S, 0, sizeof(S));
    for (int i = 0; i < NX; i++) S[i * NS + i] = 1.0;
    const double h = dt / nsub;
    const double cc[4] = {0.0, 0.5, 0.5, 1.0};
    const double bb[4] = {1.0 / 6, 2.0 / 6, 2.0 / 6, 1.0 / 6};
    for (int step = 0; step < nsub; step++) {
        double k[4][NX], K[4][NX * (NX + NU)];
        for (int st = 0; st < 4; st++) {
            double xs[NX], Ss[NX * (NX + NU)], Jx[NX * NX], Ju[NX * NU];
            for (int i = 0; i < NX; i++) xs[i] = x[i] + (st ? cc[st] * h * k[st - 1][i] : 0.0);
            for (int i = 0; i < NX * NS; i++) Ss[i] = S[i] + (st ? cc[st] * h * K[st - 1][i] : 0.0);
            stm_f_jac(p, xs, u, k[st], Jx, Ju);
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NS; j++) {
                    double acc = (j >= NX) ? Ju[i * NU + (j - NX)] : 0.0;
                    for (int l = 0; l < NX; l++) acc += Jx[i * NX + l] * Ss[l * NS + j];
                    K[st][i * NS + j] = acc;
                }
        }
        for (int i = 0; i < NX; i++)
            for (int st = 0; st < 4; st++) x[i] += h * bb[st] * k[st][i];
        for (int i = 0; i < NX * NS; i++)
            for (int st = 0; st < 4; st++) S[i] += h * bb[st] * K[st][i];
    }
    memcpy(xn, x, sizeof(x));
    for (int i = 0; i < NX; i++) {
        for (int j = 0; j < NX; j++) A[i * NX + j] = S[i * NS + j];
        for (int j = 0; j < NU; j++) B[i * NU + j] = S[i * NS + NX + j];
    }
}

/* piecewise-linear table (casadi.interpolant 'linear'; NMPC_class.py:322-335),
 * linear extrapolation from the end segments (flat for the shipped ggv.csv). */
static void interp_lin(int n, const double *xs, const double *ys, double x, double *y, double *dy)
{
    int i = 0;
    while (i < n - 2 && x >= xs[i + 1]) i++;
    double sl = (ys[i + 1] - ys[i]) / (xs[i + 1] - xs[i]);
    *y = ys[i] + sl * (x - xs[i]);
    *dy = sl;
}

/* combined_acc_limits == 2 (circle): h = (a/ax)^2 + (vl*r/ay)^2
 * NMPC_STM_acados_settings.py:70-74,108-119 */
void oracle_h(const stm_model *p, const double *x, double *h, double *gh)
{
    double ax, dax, ay, day;
    interp_lin(p->n_ggv, p->ggv_v, p->ggv_ax, x[3], &ax, &dax);
    interp_lin(p->n_ggv, p->ggv_v, p->ggv_ay, x[3], &ay, &day);
    if (x[7] < 0.0) { ax = -p->acc_min; dax = 0.0; }
    const double alat = x[3] * x[5];
    const double nlon = x[7] / ax, nlat = alat / ay;
    *h = nlon * nlon + nlat * nlat;
    if (gh) {
        for (int i = 0; i < NX; i++) gh[i] = 0.0;
        gh[3] = 2.0 * nlat * (x[5] / ay - alat / (ay * ay) * day) - 2.0 * nlon * x[7] / (ax * ax) * dax;
        gh[5] = 2.0 * nlat * x[3] / ay;
        gh[7] = 2.0 * nlon / ax;
    }
}

/* yaw wrapped to [0, 2pi): NMPC_STM_acados_settings.py:41-42 */
static double wrap_yaw(double yaw)
{
    double y = fmod(yaw, 2.0 * M_PI);
    if (y < 0.0) y += 2.0 * M_PI;
    return y;
}

/* ------------------------------------------------------------------ dense QP IPM */
typedef struct {
    int iter_max;
    double tol_stat, tol_ineq, tol_comp;
    double mu0;           /* initial complementarity target */
    double t0;            /* floor of the initial constraint residuals t */
    double reg;           /* primal regularisation added to diag(M) */
    int iter_force;       /* tests only: > 0 = run exactly this many iterations, whatever the termination test says (the HIP
                           * kernel tracks its linear residuals, this code recomputes them: an instance whose test sits on the
                           * tolerance edge can stop one iteration apart; forced to the kernel's count the two follow the same
                           * path and are compared at full tolerance). 0 = off */
    /* --- iteration-count experiments (round 5; all 0 = the shipped method). Studied by scripts/study/ipm_iterations.py,
     * outcome in profiles/r05_ipm_iterations.txt and DESIGN.md */
    int warm;             /* > 0: start from the multipliers / violation slacks of the PREVIOUS QP of this OCP (the reference's SNMPC
                           * sets qp_solver_warm_start = 1, SNMPC_acados_settings.py:307), pushed back into the interior and re-centred
                           * to warm_mu; the variants 1.. differ in how (see qp_ipm) */
    double warm_mu;       /* complementarity target of a warm start */
    int split;            /* 1: separate step lengths for the primal (v, t, s) and the dual (lam, mu) variables (HPIPM's split_step, which its
                           * SPEED modes use and BALANCE -- the reference's mode -- does not) */
    int vstart;           /* 1: primal start at the unconstrained minimiser v = -H^-1 q (one extra factorisation of H and one back-solve) instead
                           * of v = 0; 2: only when the start v = 0 is far from stationary (a cold start), |q|_inf > vstart_q */
    double vstart_q;
    int ncorr;            /* Gondzio multiple centrality correctors per iteration (extra back-solves on the same factorisation) */
    double corr_beta_lo, corr_beta_hi, corr_dalpha, corr_gain;   /* their usual constants: 0.1, 10, 0.1 (0.3), 1.01 */
    long solves, factorisations, backsolves;                     /* work counters of the study (accumulated by qp_ipm) */
    /* how far the previous QP's final point is from the new problem (computed when a warm start is offered, study + safeguard):
     * [0] max lam_prev * margin_new over the rows (old multipliers on rows that are inactive now), [1] the largest new violation beyond
     * the old slack, [2] max |lam_prev|, [3] the number of rows whose activity differs */
    double warm_meas[4];
    double warm_gate[2];  /* study: a warm start is used only if warm_meas[0] <= warm_gate[0] and warm_meas[1] <= warm_gate[1] (0: no gate) */
    int warm_flips;       /* safeguard (shipped: IPM_WARM_FLIPS_DEFAULT, together with warm_gate[1] = IPM_WARM_VIOL_DEFAULT): the warm start is used
                           * only if at most this many row sides changed their activity between the previous QP's solution and the new problem
                           * (warm_meas[3]) and no row is violated by more than warm_gate[1] beyond its old slack (warm_meas[1]); < 0: no gate.
                           * Consecutive QPs of a closed loop differ in 0-2 row sides (99 % of the logged solves pass), a real-time iteration
                           * sequence whose initial state jumps differs in dozens -- there the stale multipliers cost iterations (8.55 against
                           * 8.09 cold) and can stall the method at its iteration cap; gated: 7.91 (scripts/study/warm_gate.py) */
    int warm_used;        /* out: did the last QP start warm */
    /* study: the corrector pass of an iteration is SKIPPED (the affine step is taken at 0.995 of its step to the boundary) when the predictor
     * already asks for next to no centring: sigma <= skip_sigma and its step to the boundary >= skip_amax (0 / 0 = never: the shipped method).
     * Saves a back-solve with its row phases, a quarter of an iteration in the work model */
    double skip_sigma, skip_amax;
    long skipped;
} ipm_opts;

typedef struct {
    int iter; int status; /* 0 ok, 1 max iter, 2 min step, 3 nan */
    double res_stat, res_ineq, res_comp;
} ipm_info;

/* Cholesky M = L L^T in place (lower), returns 0 ok */
static int chol_lower(int n, double *M, int ld)
{
    for (int j = 0; j < n; j++) {
        double d = M[j * ld + j];
        for (int k = 0; k < j; k++) d -= M[j * ld + k] * M[j * ld + k];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        M[j * ld + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = M[i * ld + j];
            for (int k = 0; k < j; k++) s -= M[i * ld + k] * M[j * ld + k];
            M[i * ld + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(int n, const double *L, int ld, double *b)
{
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * ld + k] * b[k];
        b[i] = s / L[i * ld + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[k * ld + i] * b[k];
        b[i] = s / L[i * ld + i];
    }
}

/*
 * min 1/2 v'Hv + q'v + sum_i [ zl_i sl_i + 1/2 Zl_i sl_i^2 + zu_i su_i + 1/2 Zu_i su_i^2 ]
 * s.t. l_i - sl_i <= c_i'v + d_i <= u_i + su_i,  sl, su >= 0        (i = 0..m-1)
 *
 * Per row and side (eps = +1 lower, -1 upper):
 *   t = eps*(c'v + d - bound) + s >= 0  (multiplier lam),   s >= 0 (multiplier mu)
 * Infeasible-start Mehrotra predictor-corrector; the slack/multiplier blocks are
 * eliminated so that each iteration factorises the nv x nv matrix H + C' Gamma C
 * (what HPIPM's dense IPM does after removing the soft-constraint slacks).
 * Side index: 0 = lower, 1 = upper; arrays are [2*m] with side-major layout.
 */
static long g_work[3];          /* study counters over all OCPs (OpenMP-safe): QPs, factorisations, back-solves */
void oracle_global_work(long *out, int reset)
{
    for (int i = 0; i < 3; i++) { out[i] = g_work[i]; if (reset) g_work[i] = 0; }
}
#define IPM_SO_ALPHA_MIN 0.1    /* second-order correction only if the affine step length reaches this */
/* Warm start of the interior point method in a sequence of real-time iterations (adopted in round 5 after the study of
 * scripts/study/ipm_iterations.py, profiles/r05_ipm_iterations.txt: -9 % interior point work over the 285 948 logged warm solves, the
 * parity gate better than with the cold start; the reference's SNMPC solver runs with qp_solver_warm_start = 1,
 * SNMPC_acados_settings.py:307): variant 5 of qp_ipm's initial point at the complementarity target below, used whenever the previous
 * QP of the same OCP converged; a cold start (no previous QP) is untouched. oracle_set_ipm_experiment(o, 0, ...) switches it off. */
#define IPM_WARM_DEFAULT 5
#define IPM_WARM_MU_DEFAULT 1e-2
#define IPM_WARM_FLIPS_DEFAULT 16
#define IPM_WARM_VIOL_DEFAULT 0.1
static void qp_ipm(int nv, int m, const double *H, const double *q, const double *C, const double *d,
                   const double *lb, const double *ub,
                   const double *zl, const double *zu, const double *Zl, const double *Zu,
                   ipm_opts *opt, double *v, double *s_out, double *lam_out, ipm_info *info,
                   const double *s_warm, const double *lam_warm)
{
    const int M2 = 2 * m;
    double *s = calloc(8 * M2 + 8, sizeof(double));
    double *t = s + M2, *lam = t + M2, *mu = lam + M2;
    double *ds = mu + M2, *dt = ds + M2, *dlam = dt + M2, *dmu = dlam + M2;
    double *z = malloc(sizeof(double) * 10 * M2), *Z = z + M2, *bnd = Z + M2, *eps = bnd + M2;
    double *rs = eps + M2, *rt = rs + M2, *rc1 = rt + M2, *rc2 = rc1 + M2, *gam = rc2 + M2, *rho = gam + M2;
    double *Mx = malloc(sizeof(double) * nv * nv);
    double *rv = malloc(sizeof(double) * 4 * nv), *dv = rv + nv, *rhs = dv + nv, *e = malloc(sizeof(double) * 2 * m);
    double *cdv = e + m;

    for (int i = 0; i < m; i++) {
        z[i] = zl[i]; Z[i] = Zl[i]; bnd[i] = lb[i]; eps[i] = 1.0;
        z[m + i] = zu[i]; Z[m + i] = Zu[i]; bnd[m + i] = ub[i]; eps[m + i] = -1.0;
    }
    /* --- initial point: v = 0, t consistent with the slack, multipliers mu0 / (.) */
    for (int j = 0; j < nv; j++) v[j] = 0.0;
    long n_fact = 0, n_back = 0;
    double *e0 = NULL;
    {
        double qmax = 0.0;
        for (int j = 0; j < nv; j++) if (fabs(q[j]) > qmax) qmax = fabs(q[j]);
        if (opt->vstart == 1 || (opt->vstart == 2 && qmax > opt->vstart_q)) {
            memcpy(Mx, H, sizeof(double) * nv * nv);
            if (!chol_lower(nv, Mx, nv)) {
                for (int j = 0; j < nv; j++) v[j] = -q[j];
                chol_solve(nv, Mx, nv, v);
                e0 = malloc(sizeof(double) * m);
                for (int i = 0; i < m; i++) { double acc = 0.0; for (int j = 0; j < nv; j++) acc += C[i * nv + j] * v[j]; e0[i] = acc; }
                n_fact++; n_back++; opt->factorisations++; opt->backsolves++;
            }
        }
    }
    int warm = (opt->warm > 0 && s_warm && lam_warm) ? opt->warm : 0;
    if (warm) {
        double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
        for (int k = 0; k < M2; k++) {
            int i = k % m;
            double r0 = eps[k] * (d[i] + (e0 ? e0[i] : 0.0) - bnd[k]);
            double a = lam_warm[k] * (r0 > 0.0 ? r0 : 0.0), b = -r0 - s_warm[k];
            if (a > m0) m0 = a;
            if (b > m1) m1 = b;
            if (lam_warm[k] > m2) m2 = lam_warm[k];
            if ((lam_warm[k] > 1e-3) != (r0 + s_warm[k] < 1e-3)) m3 += 1.0;
        }
        opt->warm_meas[0] = m0; opt->warm_meas[1] = m1; opt->warm_meas[2] = m2; opt->warm_meas[3] = m3;
        if ((opt->warm_gate[0] > 0.0 && m0 > opt->warm_gate[0]) || (opt->warm_gate[1] > 0.0 && m1 > opt->warm_gate[1])) warm = 0;
        if (opt->warm_flips >= 0 && m3 > (double)opt->warm_flips) warm = 0;
    }
    opt->warm_used = warm != 0;
    for (int k = 0; k < M2; k++) {
        int i = k % m;
        double r0 = eps[k] * (d[i] + (e0 ? e0[i] : 0.0) - bnd[k]);       /* >= 0 : satisfied at the starting v */
        /* slack-equation-feasible start: the violation slack starts where s*z = mu0, its multiplier from
         * z + Z s - lam - mu = 0 (floored), the constraint residual t keeps a floor t0. Cuts the iteration
         * count of the plain s = t = sqrt(mu0) start by ~35 % cold and ~55 % warm on the reference's problems. */
        double mu0 = opt->mu0, t0 = opt->t0;
        if (warm) { mu0 = opt->warm_mu; t0 = opt->warm_mu; }
        double s0 = mu0 / (z[k] > 1e-6 ? z[k] : 1e-6);
        if (warm >= 2 && s_warm[k] > s0) s0 = s_warm[k];          /* a violation slack the last QP ended with stays */
        s[k] = s0;
        t[k] = r0 + s0;
        if (t[k] < t0) t[k] = t0;
        lam[k] = mu0 / t[k];
        if (warm == 5 || warm == 6) {
            /* the last multiplier is trusted only as far as the row is still near its bound in the NEW problem: kept within a factor
             * kcap of the centred value mu0 / t (a row that left its bound gets a small multiplier again instead of a badly centred pair) */
            const double kcap = (warm == 5) ? 10.0 : 100.0;
            double lw = lam_warm[k];
            if (lw > kcap * lam[k]) lw = kcap * lam[k];
            if (lw > lam[k]) {
                double cap = 0.99 * (z[k] + Z[k] * s0); if (lw > cap) lw = cap;
                lam[k] = lw;
                double tc = mu0 / lam[k];
                if (t[k] < tc) t[k] = tc;
            }
        } else
        if (warm >= 2 && lam_warm[k] > lam[k]) {
            /* the last QP's multiplier of a row that was active: kept, and the row pushed off the boundary to where the pair is
             * centred at mu0 (t lam = mu0) unless the row is further inside anyway */
            lam[k] = lam_warm[k];
            if (warm >= 3) { double cap = 0.99 * (z[k] + Z[k] * s0); if (lam[k] > cap) lam[k] = cap; }
            double tc = mu0 / lam[k];
            if (warm == 2 || warm == 3) { if (t[k] < tc) t[k] = tc; }
            else if (warm == 4) { if (r0 + s0 < tc) t[k] = tc; }
        }
        double ms = z[k] + Z[k] * s0 - lam[k];
        if (ms < 1e-2 * mu0 / s0) ms = 1e-2 * mu0 / s0;
        mu[k] = ms;
    }
    opt->solves++;
    free(e0);
    int it = 0, status = 1;
    double res_stat = 0, res_ineq = 0, res_comp = 0;
    double qn = 1.0;
    for (int j = 0; j < nv; j++) if (fabs(q[j]) > qn) qn = fabs(q[j]);
    for (it = 0; ; it++) {
        /* residuals */
        for (int i = 0; i < m; i++) {
            double acc = d[i];
            for (int j = 0; j < nv; j++) acc += C[i * nv + j] * v[j];
            e[i] = acc;
        }
        for (int j = 0; j < nv; j++) {
            double acc = q[j];
            for (int l = 0; l < nv; l++) acc += H[j * nv + l] * v[l];
            for (int i = 0; i < m; i++) acc -= C[i * nv + j] * (lam[i] - lam[m + i]);
            rv[j] = acc;
        }
        res_stat = 0; res_ineq = 0; res_comp = 0;
        double gap = 0.0;
        for (int j = 0; j < nv; j++) if (fabs(rv[j]) > res_stat) res_stat = fabs(rv[j]);
        for (int k = 0; k < M2; k++) {
            int i = k % m;
            rs[k] = z[k] + Z[k] * s[k] - lam[k] - mu[k];
            rt[k] = t[k] - eps[k] * (e[i] - bnd[k]) - s[k];
            if (fabs(rs[k]) > res_stat) res_stat = fabs(rs[k]);
            if (fabs(rt[k]) > res_ineq) res_ineq = fabs(rt[k]);
            gap += t[k] * lam[k] + s[k] * mu[k];
            if (t[k] * lam[k] > res_comp) res_comp = t[k] * lam[k];
            if (s[k] * mu[k] > res_comp) res_comp = s[k] * mu[k];
        }
        gap /= (2.0 * M2);
        if (!(res_stat == res_stat) || !(gap == gap)) { status = 3; break; }
        if (opt->iter_force > 0) { if (it >= opt->iter_force) { status = 0; break; } }
        else if (res_stat <= opt->tol_stat * qn && res_ineq <= opt->tol_ineq && res_comp <= opt->tol_comp) { status = 0; break; }
        if (it >= opt->iter_max) { status = 1; break; }

        /* --- factorise M = H + sum gamma c c' */
        for (int k = 0; k < M2; k++) {
            double Ds = Z[k] + mu[k] / s[k];
            gam[k] = 1.0 / (t[k] / lam[k] + 1.0 / Ds);
        }
        memcpy(Mx, H, sizeof(double) * nv * nv);
        for (int j = 0; j < nv; j++) Mx[j * nv + j] += opt->reg;
        for (int i = 0; i < m; i++) {
            double g = gam[i] + gam[m + i];
            const double *ci = C + i * nv;
            for (int j = 0; j < nv; j++) {
                if (ci[j] == 0.0) continue;
                double gc = g * ci[j];
                for (int l = 0; l <= j; l++) Mx[j * nv + l] += gc * ci[l];
            }
        }
        if (chol_lower(nv, Mx, nv)) { status = 3; break; }
        opt->factorisations++; n_fact++;

        double sigma = 0.0, mu_aff = 0.0, alpha = 1.0, so = 1.0, alpha_p = 1.0, alpha_d = 1.0;
        double amax_acc = 1.0, tau_c = 0.0;          /* (centrality correctors: step to the boundary of the accepted direction, its target) */
        double *sav = NULL;
        for (int pass = 0; pass < 2 + opt->ncorr; pass++) {
            opt->backsolves++; n_back++;
            /* complementarity residuals: predictor (tau = 0) then corrector */
            for (int k = 0; k < M2; k++) {
                if (pass == 0) { rc1[k] = t[k] * lam[k]; rc2[k] = s[k] * mu[k]; }
                else if (pass >= 2) {
                    /* Gondzio's multiple centrality correctors: at the trial point of an enlarged step the complementarity products
                     * are projected into [beta_lo, beta_hi] x target; the projection error corrects the right-hand side */
                    double at = amax_acc + opt->corr_dalpha; if (at > 1.0) at = 1.0;
                    double lo = opt->corr_beta_lo * tau_c, hi = opt->corr_beta_hi * tau_c;
                    double p1 = (t[k] + at * dt[k]) * (lam[k] + at * dlam[k]), p2 = (s[k] + at * ds[k]) * (mu[k] + at * dmu[k]);
                    double c1 = (p1 < lo) ? lo - p1 : (p1 > hi ? hi - p1 : 0.0), c2 = (p2 < lo) ? lo - p2 : (p2 > hi ? hi - p2 : 0.0);
                    if (c1 < -hi) c1 = -hi;
                    if (c2 < -hi) c2 = -hi;
                    rc1[k] -= c1; rc2[k] -= c2;
                }
                else {
                    /* centring target, floored so the products settle just below tol_comp instead of
                     * collapsing to 0 (which would blow up gamma = lam/t and the conditioning of M) */
                    double tau = sigma * gap;
                    if (tau < 0.1 * opt->tol_comp) tau = 0.1 * opt->tol_comp;
                    tau_c = tau;
                    rc1[k] = t[k] * lam[k] + so * dt[k] * dlam[k] - tau;
                    rc2[k] = s[k] * mu[k] + so * ds[k] * dmu[k] - tau;
                }
                double Ds = Z[k] + mu[k] / s[k];
                rho[k] = -rt[k] + rc1[k] / lam[k] - (rs[k] + rc2[k] / s[k]) / Ds;
            }
            if (pass >= 2) {          /* keep the accepted direction: a corrector that does not lengthen the step is dropped */
                if (!sav) sav = malloc(sizeof(double) * (4 * M2 + nv + 2 * M2));
                memcpy(sav, dt, sizeof(double) * M2); memcpy(sav + M2, ds, sizeof(double) * M2);
                memcpy(sav + 2 * M2, dlam, sizeof(double) * M2); memcpy(sav + 3 * M2, dmu, sizeof(double) * M2);
                memcpy(sav + 4 * M2, dv, sizeof(double) * nv);
            }
            for (int j = 0; j < nv; j++) rhs[j] = -rv[j];
            for (int i = 0; i < m; i++) {
                double w = gam[i] * rho[i] - gam[m + i] * rho[m + i];
                const double *ci = C + i * nv;
                for (int j = 0; j < nv; j++) rhs[j] -= ci[j] * w;
            }
            memcpy(dv, rhs, sizeof(double) * nv);
            chol_solve(nv, Mx, nv, dv);
            for (int i = 0; i < m; i++) {
                double acc = 0.0;
                for (int j = 0; j < nv; j++) acc += C[i * nv + j] * dv[j];
                cdv[i] = acc;
            }
            double amax = 1.0, amax_p = 1.0, amax_d = 1.0;
            int blk = -1, blkw = 0;
            for (int k = 0; k < M2; k++) {
                int i = k % m;
                double Ds = Z[k] + mu[k] / s[k];
                dlam[k] = -gam[k] * (eps[k] * cdv[i] + rho[k]);
                ds[k] = (dlam[k] - rs[k] - rc2[k] / s[k]) / Ds;
                dmu[k] = (-rc2[k] - mu[k] * ds[k]) / s[k];
                dt[k] = (-rc1[k] - t[k] * dlam[k]) / lam[k];
                if (dt[k] < 0.0 && -t[k] / dt[k] < amax) { amax = -t[k] / dt[k]; blk = k; blkw = 0; }
                if (ds[k] < 0.0 && -s[k] / ds[k] < amax) { amax = -s[k] / ds[k]; blk = k; blkw = 1; }
                if (dlam[k] < 0.0 && -lam[k] / dlam[k] < amax) { amax = -lam[k] / dlam[k]; blk = k; blkw = 2; }
                if (dmu[k] < 0.0 && -mu[k] / dmu[k] < amax) { amax = -mu[k] / dmu[k]; blk = k; blkw = 3; }
                if (dt[k] < 0.0 && -t[k] / dt[k] < amax_p) amax_p = -t[k] / dt[k];
                if (ds[k] < 0.0 && -s[k] / ds[k] < amax_p) amax_p = -s[k] / ds[k];
                if (dlam[k] < 0.0 && -lam[k] / dlam[k] < amax_d) amax_d = -lam[k] / dlam[k];
                if (dmu[k] < 0.0 && -mu[k] / dmu[k] < amax_d) amax_d = -mu[k] / dmu[k];
            }
            if (getenv("TUM_ORACLE_TRACE") && getenv("TUM_ORACLE_TRACE")[0] == '2') {
                int kx = 0; double px = 0; int which = 0;
                for (int k = 0; k < M2; k++) { if (t[k] * lam[k] > px) { px = t[k] * lam[k]; kx = k; which = 0; } if (s[k] * mu[k] > px) { px = s[k] * mu[k]; kx = k; which = 1; } }
                fprintf(stderr, "   pass %d amax %.4f blocked by k=%d (row %d side %d) var %d: t %.3e s %.3e lam %.3e mu %.3e | dt %.3e ds %.3e dlam %.3e dmu %.3e || max product k=%d (row %d side %d) %s: t %.3e s %.3e lam %.3e mu %.3e dt %.3e ds %.3e dlam %.3e dmu %.3e Z %.3e z %.3e\n",
                        pass, amax, blk, blk < 0 ? -1 : blk % m, blk < 0 ? -1 : blk / m, blkw, blk < 0 ? 0 : t[blk], blk < 0 ? 0 : s[blk], blk < 0 ? 0 : lam[blk], blk < 0 ? 0 : mu[blk],
                        blk < 0 ? 0 : dt[blk], blk < 0 ? 0 : ds[blk], blk < 0 ? 0 : dlam[blk], blk < 0 ? 0 : dmu[blk],
                        kx, kx % m, kx / m, which ? "s*mu" : "t*lam", t[kx], s[kx], lam[kx], mu[kx], dt[kx], ds[kx], dlam[kx], dmu[kx], Z[kx], z[kx]);
            }
            if (pass == 0) {
                mu_aff = 0.0;
                const double ap0 = opt->split ? amax_p : amax, ad0 = opt->split == 1 ? amax_d : (opt->split == 2 ? amax_p : amax);
                for (int k = 0; k < M2; k++) {
                    double ln = lam[k] + ad0 * dlam[k], mn = mu[k] + ad0 * dmu[k];
                    if (opt->split == 2) { if (ln < 0.0) ln = 0.0; if (mn < 0.0) mn = 0.0; }
                    mu_aff += (t[k] + ap0 * dt[k]) * ln + (s[k] + ap0 * ds[k]) * mn;
                }
                mu_aff /= (2.0 * M2);
                double ratio = mu_aff / gap;
                sigma = ratio * ratio * ratio;
                /* Safeguard: when the affine step is blocked almost immediately (a badly centred soft-constraint pair at
                 * the kink of its L1 penalty), its second-order terms are a wild extrapolation and the corrector can cycle
                 * between the two sides of the kink for all 50 iterations (seen on 7 of 286 000 logged closed-loop solves;
                 * acados' own logs show iteration-cap hits in the same loops). Drop them for this iteration: the step becomes
                 * a plain centring step and the next iteration proceeds normally. */
                so = (amax < IPM_SO_ALPHA_MIN) ? 0.0 : 1.0;
                if (opt->skip_amax > 0.0 && sigma <= opt->skip_sigma && amax >= opt->skip_amax) {
                    alpha = (amax >= 1.0) ? 1.0 : 0.995 * amax;
                    alpha_p = alpha_d = alpha;
                    opt->skipped++;
                    break;
                }
            } else if (pass == 1) {
                alpha = 0.995 * amax; if (amax >= 1.0) alpha = 1.0; if (alpha > 1.0) alpha = 1.0;
                alpha_p = (amax_p >= 1.0) ? 1.0 : 0.995 * amax_p; alpha_d = (amax_d >= 1.0) ? 1.0 : 0.995 * amax_d;
                amax_acc = amax;
                if (amax >= 1.0) break;          /* a full step needs no corrector */
            } else {
                if (amax >= opt->corr_gain * amax_acc) {
                    amax_acc = amax;
                    alpha = 0.995 * amax; if (amax >= 1.0) alpha = 1.0; if (alpha > 1.0) alpha = 1.0;
                    if (amax >= 1.0) break;
                } else {          /* rejected: back to the accepted direction */
                    memcpy(dt, sav, sizeof(double) * M2); memcpy(ds, sav + M2, sizeof(double) * M2);
                    memcpy(dlam, sav + 2 * M2, sizeof(double) * M2); memcpy(dmu, sav + 3 * M2, sizeof(double) * M2);
                    memcpy(dv, sav + 4 * M2, sizeof(double) * nv);
                    break;
                }
            }
        }
        free(sav);
        if (getenv("TUM_ORACLE_TRACE"))
            fprintf(stderr, "ipm it %2d stat %.2e ineq %.2e comp %.2e gap %.2e sigma %.2e alpha %.4f\n", it, res_stat, res_ineq, res_comp, gap, sigma, alpha);
        if (alpha < 1e-12) { status = 2; break; }
        if (!opt->split) alpha_p = alpha_d = alpha;
        for (int j = 0; j < nv; j++) v[j] += alpha_p * dv[j];
        if (opt->split == 2) {
            /* the step length is the PRIMAL one for everything; a multiplier that this step would take through zero keeps the
             * fraction 0.005 of its value instead (the fraction-to-the-boundary rule per component): a vanishing multiplier of a
             * row that leaves its bound no longer shortens the step of all the others */
            for (int k = 0; k < M2; k++) {
                t[k] += alpha_p * dt[k]; s[k] += alpha_p * ds[k];
                double ln = lam[k] + alpha_p * dlam[k], mn = mu[k] + alpha_p * dmu[k];
                lam[k] = (ln < 0.005 * lam[k]) ? 0.005 * lam[k] : ln;
                mu[k] = (mn < 0.005 * mu[k]) ? 0.005 * mu[k] : mn;
            }
        } else
        for (int k = 0; k < M2; k++) {
            t[k] += alpha_p * dt[k]; s[k] += alpha_p * ds[k];
            lam[k] += alpha_d * dlam[k]; mu[k] += alpha_d * dmu[k];
        }
    }
    {
#ifdef _OPENMP
#pragma omp atomic
#endif
        g_work[0] += 1;
#ifdef _OPENMP
#pragma omp atomic
#endif
        g_work[1] += n_fact;
#ifdef _OPENMP
#pragma omp atomic
#endif
        g_work[2] += n_back;
    }
    if (s_out) memcpy(s_out, s, sizeof(double) * M2);
    if (lam_out) memcpy(lam_out, lam, sizeof(double) * M2);
    info->iter = it; info->status = status;
    info->res_stat = res_stat; info->res_ineq = res_ineq; info->res_comp = res_comp;
    free(s); free(z); free(Mx); free(rv); free(e);
}

/* ------------------------------------------------------------------ the OCP */
typedef struct {
    int N, nsub;
    double dt;
    stm_model model;
    ipm_opts ipm;
    /* iterate */
    double X[(NMAXH + 1) * NX], U[NMAXH * NU];
    /* problem data */
    double x0[NX];                       /* lbx_0 = ubx_0 */
    double yref[(NMAXH + 1) * 6];        /* stage N uses the first 4 */
    double W[(NMAXH + 1) * 6];           /* diagonal of W per stage (stage N: first 4) */
    /* a full (symmetric) W per stage, 6 x 6 row-major (stage N: its leading 4 x 4), used instead of the diagonal when full_w != 0:
     * acados' cost_set(i, 'W', W) takes any matrix (NMPC_class.py:290-296); the reference only ever installs a diagonal one */
    double Wf[(NMAXH + 1) * 36];
    double full_w;
    double lbu[NMAXH], ubu[NMAXH];       /* steering-rate box, stages 0..N-1 */
    double lbx[NMAXH + 1], ubx[NMAXH + 1]; /* delta box, stages 1..N */
    double lh[NMAXH + 1], uh[NMAXH + 1];   /* stages 1..N */
    /* slack penalties, canonical slot order (bu, bx, h) per stage */
    double zl[(NMAXH + 1) * 3], zu[(NMAXH + 1) * 3], Zl[(NMAXH + 1) * 3], Zu[(NMAXH + 1) * 3];
    /* outputs of the last solve */
    double A[NMAXH * NX * NX], B[NMAXH * NX * NU], b[NMAXH * NX];
    double sl[MMAX], su[MMAX];           /* slack values of the last QP, row order below */
    double lam[2 * MMAX];
    double cost;
    int qp_iter, status;
    double res[3];
    int have_qp;                         /* sl / su / lam hold a converged QP of this OCP (warm start experiments) */
} oracle_ocp;

oracle_ocp *oracle_create(int N, double dt, int nsub)
{
    if (N < 1 || N > NMAXH) return NULL;
    oracle_ocp *o = calloc(1, sizeof(oracle_ocp));
    o->N = N; o->dt = dt; o->nsub = nsub;
    o->ipm.iter_max = 50; o->ipm.tol_stat = 1e-8; o->ipm.tol_ineq = 1e-8; o->ipm.tol_comp = 1e-8;
    o->ipm.mu0 = 0.05; o->ipm.t0 = 0.05; o->ipm.reg = 0.0;
    o->ipm.corr_beta_lo = 0.1; o->ipm.corr_beta_hi = 10.0; o->ipm.corr_dalpha = 0.1; o->ipm.corr_gain = 1.01;
    o->ipm.warm = IPM_WARM_DEFAULT; o->ipm.warm_mu = IPM_WARM_MU_DEFAULT; o->ipm.warm_flips = IPM_WARM_FLIPS_DEFAULT; o->ipm.warm_gate[1] = IPM_WARM_VIOL_DEFAULT;
    return o;
}
/* the multipliers of the last QP no longer belong to this OCP's next solve (cold start, reset) */
void oracle_forget_qp(oracle_ocp *o) { o->have_qp = 0; }
/* iteration-count experiments (ipm_opts): warm-start variant, its complementarity target, number of centrality correctors */
void oracle_set_ipm_vstart(oracle_ocp *o, int vstart, double qthr) { o->ipm.vstart = vstart; o->ipm.vstart_q = qthr; }
void oracle_set_ipm_split(oracle_ocp *o, int split) { o->ipm.split = split; }
void oracle_set_ipm_skip(oracle_ocp *o, double sigma_thr, double amax_thr) { o->ipm.skip_sigma = sigma_thr; o->ipm.skip_amax = amax_thr; }
void oracle_set_warm_gate(oracle_ocp *o, double g0, double g1) { o->ipm.warm_gate[0] = g0; o->ipm.warm_gate[1] = g1; }
void oracle_set_warm_flips(oracle_ocp *o, int flips) { o->ipm.warm_flips = flips; }
void oracle_warm_meas(oracle_ocp *o, double *out) { for (int i = 0; i < 4; i++) out[i] = o->ipm.warm_meas[i]; out[4] = o->ipm.warm_used; }
void oracle_set_ipm_experiment(oracle_ocp *o, int warm, double warm_mu, int ncorr, double dalpha)
{
    o->ipm.warm = warm; if (warm_mu > 0) o->ipm.warm_mu = warm_mu; o->ipm.ncorr = ncorr; if (dalpha > 0) o->ipm.corr_dalpha = dalpha;
}
void oracle_work_counters(oracle_ocp *o, long *out, int reset)
{
    out[0] = o->ipm.solves; out[1] = o->ipm.factorisations; out[2] = o->ipm.backsolves;
    if (reset) o->ipm.solves = o->ipm.factorisations = o->ipm.backsolves = 0;
}
void oracle_free(oracle_ocp *o) { free(o); }
size_t oracle_sizeof(void) { return sizeof(oracle_ocp); }

/* field access by name for the ctypes wrapper (pointer into the struct + length) */
double *oracle_field(oracle_ocp *o, const char *name, int *len)
{
    const int N = o->N;
#define F(nm, ptr, n) if (!strcmp(name, nm)) { *len = (n); return (ptr); }
    F("X", o->X, (N + 1) * NX) F("U", o->U, N * NU) F("x0", o->x0, NX)
    F("yref", o->yref, (N + 1) * 6) F("W", o->W, (N + 1) * 6) F("Wf", o->Wf, (N + 1) * 36) F("full_w", &o->full_w, 1)
    F("lbu", o->lbu, N) F("ubu", o->ubu, N)
    F("lbx", o->lbx, N + 1) F("ubx", o->ubx, N + 1) F("lh", o->lh, N + 1) F("uh", o->uh, N + 1)
    F("zl", o->zl, (N + 1) * 3) F("zu", o->zu, (N + 1) * 3) F("Zl", o->Zl, (N + 1) * 3) F("Zu", o->Zu, (N + 1) * 3)
    F("A", o->A, N * NX * NX) F("B", o->B, N * NX * NU) F("b", o->b, N * NX)
    F("sl", o->sl, 3 * N) F("su", o->su, 3 * N) F("lam", o->lam, 6 * N)
    F("cost", &o->cost, 1) F("res", o->res, 3)
    F("ipm_tol", &o->ipm.tol_stat, 3) F("ipm_mu0", &o->ipm.mu0, 1) F("ipm_t0", &o->ipm.t0, 1) F("ipm_reg", &o->ipm.reg, 1)
#undef F
    *len = 0; return NULL;
}
void oracle_set_model(oracle_ocp *o, const stm_model *m) { o->model = *m; }
void oracle_set_iter_max(oracle_ocp *o, int it) { o->ipm.iter_max = it; }
void oracle_set_iter_force(oracle_ocp *o, int it) { o->ipm.iter_force = it; }
int oracle_qp_iter(const oracle_ocp *o) { return o->qp_iter; }
int oracle_status(const oracle_ocp *o) { return o->status; }

/* total cost at the current iterate with the given slack values
 * (acados ocp_nlp_eval_cost: stage terms scaled by dt, terminal unscaled) */
static double eval_cost(const oracle_ocp *o)
{
    const int N = o->N;
    double c = 0.0;
    for (int k = 0; k <= N; k++) {
        const double sc = (k < N) ? o->dt : 1.0;
        const double *x = o->X + k * NX, *yr = o->yref + k * 6, *W = o->W + k * 6;
        double y[6] = {x[0], x[1], wrap_yaw(x[2]), x[3], 0, 0};
        int ny = 4;
        if (k < N) { y[4] = o->U[k * NU]; y[5] = o->U[k * NU + 1]; ny = 6; }
        double acc = 0.0;
        if (o->full_w != 0.0) {
            const double *Wk = o->Wf + k * 36;
            for (int i = 0; i < ny; i++) for (int j = 0; j < ny; j++) acc += 0.5 * Wk[i * 6 + j] * (y[i] - yr[i]) * (y[j] - yr[j]);
        } else
        for (int i = 0; i < ny; i++) { double r = y[i] - yr[i]; acc += 0.5 * W[i] * r * r; }
        c += sc * acc;
    }
    /* slack terms: rows [bu_0..bu_{N-1} | (bx_k, h_k) k=1..N] */
    for (int i = 0; i < 3 * N; i++) {
        int k, slot;
        if (i < N) { k = i; slot = 0; } else { k = 1 + (i - N) / 2; slot = 1 + (i - N) % 2; }
        const double sc = (k < N) ? o->dt : 1.0;
        c += sc * (o->zl[k * 3 + slot] * o->sl[i] + 0.5 * o->Zl[k * 3 + slot] * o->sl[i] * o->sl[i]);
        c += sc * (o->zu[k * 3 + slot] * o->su[i] + 0.5 * o->Zu[k * 3 + slot] * o->su[i] * o->su[i]);
    }
    return c;
}

/* optional dump of the condensed QP of the next oracle_solve calls (tests only):
 * layout [H nv*nv | q nv | C m*nv | d m | g (N+1)*8] */
static double *g_dbg = NULL;
void oracle_set_debug(double *buf) { g_dbg = buf; }

/* One SQP real-time iteration. Returns acados-like status. */
int oracle_solve(oracle_ocp *o)
{
    const int N = o->N, nv = NU * N, m = 3 * N;
    const double dt = o->dt;
    double *G = calloc((size_t)(N + 1) * NX * nv, sizeof(double));   /* G_k: 8 x nv each */
    double *g = calloc((size_t)(N + 1) * NX, sizeof(double));
    double *H = calloc((size_t)nv * nv, sizeof(double)), *q = calloc(nv, sizeof(double));
    double *C = calloc((size_t)m * nv, sizeof(double)), *d = calloc(m, sizeof(double));
    double *lb = calloc(6 * m, sizeof(double)), *ub = lb + m, *zl = ub + m, *zu = zl + m, *Zl = zu + m, *Zu = Zl + m;
    double *v = calloc(nv, sizeof(double));

    /* 1. linearise dynamics */
    for (int k = 0; k < N; k++) {
        double xn[NX];
        oracle_rk4_sens(&o->model, o->X + k * NX, o->U + k * NU, dt, o->nsub, xn, o->A + k * 64, o->B + k * 16);
        for (int i = 0; i < NX; i++) o->b[k * NX + i] = xn[i] - o->X[(k + 1) * NX + i];
    }
    /* 4./5. initial-value embedding and condensing: dx_k = G_k v + g_k */
    for (int i = 0; i < NX; i++) g[i] = o->x0[i] - o->X[i];
    for (int k = 0; k < N; k++) {
        const double *A = o->A + k * 64, *B = o->B + k * 16;
        double *Gn = G + (size_t)(k + 1) * NX * nv, *Gk = G + (size_t)k * NX * nv;
        for (int i = 0; i < NX; i++) {
            for (int j = 0; j < NU * k; j++) {
                double acc = 0.0;
                for (int l = 0; l < NX; l++) acc += A[i * NX + l] * Gk[l * nv + j];
                Gn[i * nv + j] = acc;
            }
            for (int j = 0; j < NU; j++) Gn[i * nv + NU * k + j] = B[i * NU + j];
            double acc = o->b[k * NX + i];
            for (int l = 0; l < NX; l++) acc += A[i * NX + l] * g[k * NX + l];
            g[(k + 1) * NX + i] = acc;
        }
    }
    /* 2. Gauss-Newton cost: y = [x0,x1,wrap(x2),x3,u0,u1], J = selection */
    for (int k = 0; k <= N; k++) {
        const double sc = (k < N) ? dt : 1.0;
        const double *x = o->X + k * NX, *yr = o->yref + k * 6, *W = o->W + k * 6;
        const double *Gk = G + (size_t)k * NX * nv;
        double y[4] = {x[0], x[1], wrap_yaw(x[2]), x[3]};
        if (o->full_w != 0.0) {
            /* full W: rows of the output Jacobian in the condensed variables -- J_r = row r of G_k (r < 4), J_{4+r} = unit row of input
             * r of stage k -- and H += J' (sc W) J, q += J' (sc W) res, the dumbest way */
            const double *Wk = o->Wf + k * 36;
            const int ny = (k < N) ? 6 : 4;
            double res[6];
            for (int r = 0; r < 4; r++) res[r] = y[r] - yr[r] + g[k * NX + r];
            if (k < N) for (int r = 0; r < NU; r++) res[4 + r] = o->U[k * NU + r] - yr[4 + r];
            const int ncol = (k < N) ? NU * (k + 1) : NU * k;       /* columns the stage's rows reach */
            for (int j = 0; j < ncol; j++) {
                double Jj[6];
                for (int r = 0; r < 4; r++) Jj[r] = (j < NU * k) ? Gk[r * nv + j] : 0.0;
                for (int r = 0; r < NU; r++) Jj[4 + r] = (k < N && j == NU * k + r) ? 1.0 : 0.0;
                double WJ[6];
                for (int a = 0; a < ny; a++) { double t = 0.0; for (int c2 = 0; c2 < ny; c2++) t += sc * Wk[a * 6 + c2] * Jj[c2]; WJ[a] = t; }
                for (int a = 0; a < ny; a++) q[j] += WJ[a] * res[a];
                for (int l = 0; l <= j; l++) {
                    double t = 0.0;
                    for (int a = 0; a < 4; a++) t += WJ[a] * ((l < NU * k) ? Gk[a * nv + l] : 0.0);
                    for (int r = 0; r < NU; r++) if (k < N && l == NU * k + r) t += WJ[4 + r];
                    H[j * nv + l] += t;
                }
            }
            continue;
        }
        for (int r = 0; r < 4; r++) {
            const double w = sc * W[r];
            const double res = y[r] - yr[r] + g[k * NX + r];     /* residual incl. the constant part of dx */
            const double *Gr = Gk + r * nv;
            for (int j = 0; j < NU * k; j++) {
                if (Gr[j] == 0.0) continue;
                q[j] += w * res * Gr[j];
                for (int l = 0; l <= j; l++) H[j * nv + l] += w * Gr[j] * Gr[l];
            }
        }
        if (k < N)
            for (int r = 0; r < NU; r++) {
                const double w = sc * W[4 + r];
                H[(NU * k + r) * nv + NU * k + r] += w;
                q[NU * k + r] += w * (o->U[k * NU + r] - yr[4 + r]);
            }
    }
    for (int j = 0; j < nv; j++) for (int l = 0; l < j; l++) H[l * nv + j] = H[j * nv + l];
    /* 3. constraints. Row order: [bu_k k=0..N-1 | (bx_k, h_k) k=1..N] */
    for (int k = 0; k < N; k++) {
        const double sc = dt;
        C[k * nv + NU * k + 1] = 1.0;
        d[k] = o->U[k * NU + 1];
        lb[k] = o->lbu[k]; ub[k] = o->ubu[k];
        zl[k] = sc * o->zl[k * 3]; zu[k] = sc * o->zu[k * 3]; Zl[k] = sc * o->Zl[k * 3]; Zu[k] = sc * o->Zu[k * 3];
    }
    for (int k = 1; k <= N; k++) {
        const double sc = (k < N) ? dt : 1.0;
        const double *x = o->X + k * NX;
        const double *Gk = G + (size_t)k * NX * nv;
        int ib = N + 2 * (k - 1), ih = ib + 1;
        for (int j = 0; j < NU * k; j++) C[ib * nv + j] = Gk[6 * nv + j];
        d[ib] = x[6] + g[k * NX + 6];
        lb[ib] = o->lbx[k]; ub[ib] = o->ubx[k];
        double h, gh[NX];
        oracle_h(&o->model, x, &h, gh);
        double acc = h;
        for (int l = 0; l < NX; l++) acc += gh[l] * g[k * NX + l];
        d[ih] = acc;
        for (int j = 0; j < NU * k; j++) {
            double a2 = 0.0;
            for (int l = 0; l < NX; l++) a2 += gh[l] * Gk[l * nv + j];
            C[ih * nv + j] = a2;
        }
        lb[ih] = o->lh[k]; ub[ih] = o->uh[k];
        for (int s = 1; s <= 2; s++) {
            int i = (s == 1) ? ib : ih;
            zl[i] = sc * o->zl[k * 3 + s]; zu[i] = sc * o->zu[k * 3 + s];
            Zl[i] = sc * o->Zl[k * 3 + s]; Zu[i] = sc * o->Zu[k * 3 + s];
        }
    }
    if (g_dbg) {
        double *p = g_dbg;
        memcpy(p, H, sizeof(double) * nv * nv); p += nv * nv;
        memcpy(p, q, sizeof(double) * nv); p += nv;
        memcpy(p, C, sizeof(double) * m * nv); p += m * nv;
        memcpy(p, d, sizeof(double) * m); p += m;
        memcpy(p, g, sizeof(double) * (N + 1) * NX);
    }
    /* 6. QP */
    double *sall = calloc(2 * m, sizeof(double));
    ipm_info info;
    double *swarm = NULL, *lwarm = NULL;
    if (o->ipm.warm > 0 && o->have_qp) {          /* the previous QP's violation slacks and multipliers (side-major like sall) */
        swarm = malloc(sizeof(double) * 4 * m); lwarm = swarm + 2 * m;
        memcpy(swarm, o->sl, sizeof(double) * m); memcpy(swarm + m, o->su, sizeof(double) * m);
        memcpy(lwarm, o->lam, sizeof(double) * 2 * m);
    }
    qp_ipm(nv, m, H, q, C, d, lb, ub, zl, zu, Zl, Zu, &o->ipm, v, sall, o->lam, &info, swarm, lwarm);
    free(swarm);
    memcpy(o->sl, sall, sizeof(double) * m);
    memcpy(o->su, sall + m, sizeof(double) * m);
    o->qp_iter = info.iter;
    o->have_qp = (info.status == 0);
    o->res[0] = info.res_stat; o->res[1] = info.res_ineq; o->res[2] = info.res_comp;
    /* acados: QP max-iter is not a failure; NaN / min-step -> 4 */
    o->status = (info.status == 0 || info.status == 1) ? 0 : 4;
    /* 7. full step */
    if (o->status == 0) {
        for (int k = 0; k <= N; k++) {
            const double *Gk = G + (size_t)k * NX * nv;
            for (int i = 0; i < NX; i++) {
                double acc = g[k * NX + i];
                for (int j = 0; j < NU * k; j++) acc += Gk[i * nv + j] * v[j];
                o->X[k * NX + i] += acc;
            }
        }
        for (int j = 0; j < nv; j++) o->U[j] += v[j];
    }
    o->cost = eval_cost(o);
    free(G); free(g); free(H); free(q); free(C); free(d); free(lb); free(v); free(sall);
    return o->status;
}

/* batch helper for the cpu_baseline leg: solve `nb` independent OCPs that share the
 * model / weights / bounds of `tmpl`, each from a cold start (X_k = x0, U = 0).
 * x0: nb x 8, yref: nb x (N+1) x 6. Outputs u0 (nb x 2), X1 (nb x 8), stats (nb x 3: cost, qp_iter, status).
 * OpenMP over instances when compiled with -fopenmp. */
void oracle_solve_batch_cold_forced(const oracle_ocp *tmpl, int nb, const double *x0, const double *yref,
                                    double *u0, double *X1, double *stats, int nthreads, const int *force);
void oracle_solve_batch_cold(const oracle_ocp *tmpl, int nb, const double *x0, const double *yref,
                             double *u0, double *X1, double *stats, int nthreads)
{
    oracle_solve_batch_cold_forced(tmpl, nb, x0, yref, u0, X1, stats, nthreads, NULL);
}
/* the same with the interior point iteration count of every instance imposed (force[b] > 0; see ipm_opts.iter_force) */
void oracle_solve_batch_cold_forced(const oracle_ocp *tmpl, int nb, const double *x0, const double *yref,
                                    double *u0, double *X1, double *stats, int nthreads, const int *force)
{
    (void)nthreads;
    /* keep the per-solve work arrays (a few 100 KB) in the thread arenas: without this every solve
     * mmap()s/munmap()s them and the threads serialise on the process address-space lock */
    mallopt(M_MMAP_THRESHOLD, 64 << 20);
    mallopt(M_TRIM_THRESHOLD, 256 << 20);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (int b = 0; b < nb; b++) {
        oracle_ocp *o = malloc(sizeof(oracle_ocp));
        memcpy(o, tmpl, sizeof(oracle_ocp));
        const int N = o->N;
        memcpy(o->x0, x0 + (size_t)b * NX, sizeof(double) * NX);
        memcpy(o->yref, yref + (size_t)b * (N + 1) * 6, sizeof(double) * (N + 1) * 6);
        for (int k = 0; k <= N; k++) memcpy(o->X + k * NX, o->x0, sizeof(double) * NX);
        memset(o->U, 0, sizeof(double) * N * NU);
        if (force) o->ipm.iter_force = force[b];
        oracle_solve(o);
        u0[b * 2] = o->U[0]; u0[b * 2 + 1] = o->U[1];
        memcpy(X1 + (size_t)b * NX, o->X + NX, sizeof(double) * NX);
        stats[b * 3] = o->cost; stats[b * 3 + 1] = o->qp_iter; stats[b * 3 + 2] = o->status;
        free(o);
    }
}

/* ================================================================== coupled SNMPC OCP
 *
 * One acados SQP-RTI step of the reference's stochastic NMPC, i.e. of `acados_solver.solve()` at
 *   Model_Predictive_Controller/Stochastic_NMPC/SNMPC_class.py:198
 * for the OCP of Stochastic_NMPC/SNMPC_acados_settings.py:19-320 with the DISCRETE dynamics of
 *   Stochastic_NMPC/pred_model_dynamic_disc.py:121-220.
 * State: the nominal copy followed by n_s sample copies of the 8 single-track states (nx = 8 (n_s+1));
 * ONE shared input. Per stage k the parameter vector carries the PCE matrix A (L x n_s, row-major,
 * SNMPC_class.py:124) and stop_flag_k (1 from the uncertainty propagation horizon on, SNMPC_class.py:103-104):
 *   stop_flag = 0: sample i advances by ONE RK4 step of length Ts (pred_model_dynamic_disc.py:185-203),
 *                  nominal_next = sum_i A[0,i] sample_next_i                     (:208-210)
 *                  h = E + kappa sqrt(Var) of the sample values of the gg circle (SNMPC_acados_settings.py:116-133,187)
 *   stop_flag = 1: samples are frozen, the nominal copy takes its own RK4 step, h = h(nominal copy).
 * Cost on the nominal copy with |v| as the speed row (:153-154); bounds as in the nominal OCP (:208-218),
 * no h row at stage 0 (dims nh_0 = 0, acados_ocp_SNMPC.json:704-740).
 * Deliberately the generic dense formulation: 8(n_s+1) x 8(n_s+1) stage matrices, dense condensing.
 * SOLVER OUTPUTS PARITY-UNPINNED: the reference holds no SNMPC solver outputs with a known configuration (ACC24 logs:
 * weights / lateral model of that campaign not recorded). What IS pinned (tests/test_snmpc.py):
 *  - the model functions -- stacked discrete dynamics, cost output, chance constraint -- against the reference's own
 *    expressions as exported in acados_ocp_SNMPC.json (CasADi text evaluated at random points by tests/golden/make_golden.py:
 *    dynamics to the 6 printed digits of the constants, the chance constraint to 2e-16), and the problem data against the
 *    same file (dimensions, weights, penalties, bounds, options);
 *  - the solve against the golden-pinned nominal restatement above (stop_flag = 1 everywhere), and its rows against finite
 *    differences of an independent rollout.
 */
#define NSMAX 32                 /* max samples */
#define NLMAX 32                 /* max PCE terms */
#define NXS   (NX * (NSMAX + 1))

typedef struct {
    int N, ns, L;
    double dt, kappa;
    stm_model model;
    ipm_opts ipm;
    double X[(NMAXH + 1) * NXS], U[NMAXH * NU];      /* X: stage-major, (ns+1)*8 per stage */
    double x0[NXS];
    double yref[(NMAXH + 1) * 6], W[(NMAXH + 1) * 6];
    double lbu[NMAXH], ubu[NMAXH], lbx[NMAXH + 1], ubx[NMAXH + 1], lh[NMAXH + 1], uh[NMAXH + 1];
    double zl[(NMAXH + 1) * 3], zu[(NMAXH + 1) * 3], Zl[(NMAXH + 1) * 3], Zu[(NMAXH + 1) * 3];
    double Apce[NLMAX * NSMAX];                      /* L x ns, row-major */
    double stop[NMAXH + 1];
    double sl[MMAX], su[MMAX], lam[2 * MMAX];
    double hval[NMAXH + 1];                          /* constraint value per stage at the linearisation point */
    double cost;
    int qp_iter, status;
    double res[3];
    int have_qp;                                     /* sl / su / lam hold a converged QP of this OCP (interior point warm start) */
} snmpc_ocp;

snmpc_ocp *snmpc_create(int N, double dt, int ns, int L, double gamma)
{
    if (N < 1 || N > NMAXH || ns < 1 || ns > NSMAX || L < 1 || L > NLMAX) return NULL;
    snmpc_ocp *o = calloc(1, sizeof(snmpc_ocp));
    o->N = N; o->dt = dt; o->ns = ns; o->L = L;
    o->kappa = sqrt((1.0 - gamma) / gamma);          /* SNMPC_acados_settings.py:187 */
    o->ipm.iter_max = 50; o->ipm.tol_stat = 1e-8; o->ipm.tol_ineq = 1e-8; o->ipm.tol_comp = 1e-8;
    o->ipm.mu0 = 0.05; o->ipm.t0 = 0.05; o->ipm.reg = 0.0;
    o->ipm.warm = IPM_WARM_DEFAULT; o->ipm.warm_mu = IPM_WARM_MU_DEFAULT; o->ipm.warm_flips = IPM_WARM_FLIPS_DEFAULT; o->ipm.warm_gate[1] = IPM_WARM_VIOL_DEFAULT;
    return o;
}
void snmpc_forget_qp(snmpc_ocp *o) { o->have_qp = 0; }
void snmpc_set_ipm_warm(snmpc_ocp *o, int warm, double warm_mu) { o->ipm.warm = warm; if (warm_mu > 0) o->ipm.warm_mu = warm_mu; }
void snmpc_free(snmpc_ocp *o) { free(o); }
void snmpc_set_model(snmpc_ocp *o, const stm_model *m) { o->model = *m; }
void snmpc_set_iter_max(snmpc_ocp *o, int it) { o->ipm.iter_max = it; }
int snmpc_qp_iter(const snmpc_ocp *o) { return o->qp_iter; }
int snmpc_status(const snmpc_ocp *o) { return o->status; }

double *snmpc_field(snmpc_ocp *o, const char *name, int *len)
{
    const int N = o->N, nxs = NX * (o->ns + 1);
#define F(nm, ptr, n) if (!strcmp(name, nm)) { *len = (n); return (ptr); }
    F("X", o->X, (N + 1) * nxs) F("U", o->U, N * NU) F("x0", o->x0, nxs)
    F("yref", o->yref, (N + 1) * 6) F("W", o->W, (N + 1) * 6)
    F("lbu", o->lbu, N) F("ubu", o->ubu, N)
    F("lbx", o->lbx, N + 1) F("ubx", o->ubx, N + 1) F("lh", o->lh, N + 1) F("uh", o->uh, N + 1)
    F("zl", o->zl, (N + 1) * 3) F("zu", o->zu, (N + 1) * 3) F("Zl", o->Zl, (N + 1) * 3) F("Zu", o->Zu, (N + 1) * 3)
    F("Apce", o->Apce, o->L * o->ns) F("stop", o->stop, N + 1) F("kappa", &o->kappa, 1)
    F("sl", o->sl, 3 * N) F("su", o->su, 3 * N) F("lam", o->lam, 6 * N) F("hval", o->hval, N + 1)
    F("cost", &o->cost, 1) F("res", o->res, 3)
    F("ipm_tol", &o->ipm.tol_stat, 3) F("ipm_mu0", &o->ipm.mu0, 1) F("ipm_t0", &o->ipm.t0, 1) F("ipm_reg", &o->ipm.reg, 1)
#undef F
    *len = 0; return NULL;
}

/* gg circle with the limits looked up at |v| (SNMPC_acados_settings.py:60-67,100-113) */
void oracle_h_vabs(const stm_model *p, const double *x, double *h, double *gh)
{
    const double vabs = sqrt(x[3] * x[3] + x[4] * x[4]);
    double ax, dax, ay, day;
    interp_lin(p->n_ggv, p->ggv_v, p->ggv_ax, vabs, &ax, &dax);
    interp_lin(p->n_ggv, p->ggv_v, p->ggv_ay, vabs, &ay, &day);
    if (x[7] < 0.0) { ax = -p->acc_min; dax = 0.0; }
    const double alat = x[3] * x[5];
    const double nlon = x[7] / ax, nlat = alat / ay;
    *h = nlon * nlon + nlat * nlat;
    if (gh) {
        for (int i = 0; i < NX; i++) gh[i] = 0.0;
        /* d/d|v| through the two tables, then |v| -> (vl, vt) */
        const double dv = -2.0 * nlat * alat / (ay * ay) * day - 2.0 * nlon * x[7] / (ax * ax) * dax;
        const double ivl = (vabs > 0.0) ? x[3] / vabs : 0.0, ivt = (vabs > 0.0) ? x[4] / vabs : 0.0;
        gh[3] = 2.0 * nlat * x[5] / ay + dv * ivl;
        gh[4] = dv * ivt;
        gh[5] = 2.0 * nlat * x[3] / ay;
        gh[7] = 2.0 * nlon / ax;
    }
}

/* constraint value of one stage and its gradient w.r.t. the full stacked state (nxs) */
static double snmpc_h(const snmpc_ocp *o, int k, const double *x, double *grad)
{
    const int ns = o->ns, L = o->L, nxs = NX * (ns + 1);
    for (int i = 0; i < nxs; i++) grad[i] = 0.0;
    if (o->stop[k] == 1.0) {
        double h;
        oracle_h_vabs(&o->model, x, &h, grad);
        return h;
    }
    double hs[NSMAX], ghs[NSMAX][NX], c[NLMAX];
    for (int i = 0; i < ns; i++) oracle_h_vabs(&o->model, x + NX * (i + 1), &hs[i], ghs[i]);
    for (int l = 0; l < L; l++) {
        c[l] = 0.0;
        for (int i = 0; i < ns; i++) c[l] += o->Apce[l * ns + i] * hs[i];
    }
    double var = 0.0;
    for (int l = 1; l < L; l++) var += c[l] * c[l];
    const double sd = sqrt(var);
    for (int i = 0; i < ns; i++) {
        double w = o->Apce[i];                                   /* d mean / d h_i */
        if (sd > 0.0) {
            double acc = 0.0;
            for (int l = 1; l < L; l++) acc += c[l] * o->Apce[l * ns + i];
            w += o->kappa * acc / sd;
        }
        for (int r = 0; r < NX; r++) grad[NX * (i + 1) + r] = w * ghs[i][r];
    }
    return c[0] + o->kappa * sd;
}

/* the stacked DISCRETE dynamics of one stage (pred_model_dynamic_disc.py:170-212): successor state and, if asked for,
 * the dense stage matrices A (nxs x nxs) and B (nxs x 2), both expected zero-filled */
static void snmpc_step(const snmpc_ocp *o, int stop, const double *x, const double *u, double *xn, double *A, double *B)
{
    const int ns = o->ns, nxs = NX * (ns + 1);
    const double dt = o->dt;
    double Ai[NSMAX + 1][64], Bi[NSMAX + 1][16];
    for (int i = 1; i <= ns; i++) {
        if (!stop) {
            oracle_rk4_sens(&o->model, x + NX * i, u, dt, 1, xn + NX * i, Ai[i], Bi[i]);
            if (A)
                for (int r = 0; r < NX; r++) {
                    for (int c = 0; c < NX; c++) A[(NX * i + r) * nxs + NX * i + c] = Ai[i][r * NX + c];
                    for (int c = 0; c < NU; c++) B[(NX * i + r) * NU + c] = Bi[i][r * NU + c];
                }
        } else {
            for (int r = 0; r < NX; r++) { xn[NX * i + r] = x[NX * i + r]; if (A) A[(NX * i + r) * nxs + NX * i + r] = 1.0; }
        }
    }
    if (stop) {
        oracle_rk4_sens(&o->model, x, u, dt, 1, xn, Ai[0], Bi[0]);
        if (A)
            for (int r = 0; r < NX; r++) {
                for (int c = 0; c < NX; c++) A[r * nxs + c] = Ai[0][r * NX + c];
                for (int c = 0; c < NU; c++) B[r * NU + c] = Bi[0][r * NU + c];
            }
    } else {
        /* nominal_next = first row of A_pce times the sample successors */
        for (int r = 0; r < NX; r++) {
            double acc = 0.0;
            for (int i = 1; i <= ns; i++) acc += o->Apce[i - 1] * xn[NX * i + r];
            xn[r] = acc;
            if (A)
                for (int i = 1; i <= ns; i++) {
                    const double a = o->Apce[i - 1];
                    for (int c = 0; c < NX; c++) A[r * nxs + NX * i + c] = a * Ai[i][r * NX + c];
                    for (int c = 0; c < NU; c++) B[r * NU + c] += a * Bi[i][r * NU + c];
                }
        }
    }
}

/* model functions of one stage at a given point, for the checks against the reference's exported expressions
 * (acados_ocp_SNMPC.json model.disc_dyn_expr / cost_y_expr / con_h_expr; tests/golden/snmpc_expr.npz):
 * xn = f_disc(x, u, p), y = cost_y_expr(x, u) (6 entries), h = con_h_expr(x, p); stop = the stage's stop_flag */
void snmpc_eval(snmpc_ocp *o, double stop, const double *x, const double *u, double *xn, double *y, double *h)
{
    double grad[NXS];
    snmpc_step(o, stop == 1.0, x, u, xn, NULL, NULL);
    y[0] = x[0]; y[1] = x[1]; y[2] = wrap_yaw(x[2]); y[3] = sqrt(x[3] * x[3] + x[4] * x[4]); y[4] = u[0]; y[5] = u[1];
    const double keep = o->stop[0];
    o->stop[0] = stop;
    *h = snmpc_h(o, 0, x, grad);
    o->stop[0] = keep;
}

static double snmpc_eval_cost(const snmpc_ocp *o)
{
    const int N = o->N, nxs = NX * (o->ns + 1);
    double c = 0.0;
    for (int k = 0; k <= N; k++) {
        const double sc = (k < N) ? o->dt : 1.0;
        const double *x = o->X + (size_t)k * nxs, *yr = o->yref + k * 6, *W = o->W + k * 6;
        double y[6] = {x[0], x[1], wrap_yaw(x[2]), sqrt(x[3] * x[3] + x[4] * x[4]), 0, 0};
        int ny = 4;
        if (k < N) { y[4] = o->U[k * NU]; y[5] = o->U[k * NU + 1]; ny = 6; }
        double acc = 0.0;
        for (int i = 0; i < ny; i++) { double r = y[i] - yr[i]; acc += 0.5 * W[i] * r * r; }
        c += sc * acc;
    }
    for (int i = 0; i < 3 * N; i++) {
        int k, slot;
        if (i < N) { k = i; slot = 0; } else { k = 1 + (i - N) / 2; slot = 1 + (i - N) % 2; }
        const double sc = (k < N) ? o->dt : 1.0;
        c += sc * (o->zl[k * 3 + slot] * o->sl[i] + 0.5 * o->Zl[k * 3 + slot] * o->sl[i] * o->sl[i]);
        c += sc * (o->zu[k * 3 + slot] * o->su[i] + 0.5 * o->Zu[k * 3 + slot] * o->su[i] * o->su[i]);
    }
    return c;
}

static double *g_sdbg = NULL;    /* [H nv*nv | q nv | C m*nv | d m] of the next snmpc_solve calls (tests only) */
void snmpc_set_debug(double *buf) { g_sdbg = buf; }

int snmpc_solve(snmpc_ocp *o)
{
    const int N = o->N, ns = o->ns, nxs = NX * (ns + 1), nv = NU * N, m = 3 * N;
    const double dt = o->dt;
    double *Ab = calloc((size_t)N * nxs * nxs, sizeof(double));      /* stage matrices, dense */
    double *Bb = calloc((size_t)N * nxs * NU, sizeof(double));
    double *bb = calloc((size_t)N * nxs, sizeof(double));
    double *G = calloc((size_t)(N + 1) * nxs * nv, sizeof(double));
    double *g = calloc((size_t)(N + 1) * nxs, sizeof(double));
    double *H = calloc((size_t)nv * nv, sizeof(double)), *q = calloc(nv, sizeof(double));
    double *C = calloc((size_t)m * nv, sizeof(double)), *d = calloc(m, sizeof(double));
    double *lb = calloc(6 * m, sizeof(double)), *ub = lb + m, *zl = ub + m, *zu = zl + m, *Zl = zu + m, *Zu = Zl + m;
    double *v = calloc(nv, sizeof(double));
    double *row = calloc(nv, sizeof(double));

    /* 1. discrete dynamics and their Jacobians */
    for (int k = 0; k < N; k++) {
        double xn[NXS];
        snmpc_step(o, o->stop[k] == 1.0, o->X + (size_t)k * nxs, o->U + k * NU, xn,
                   Ab + (size_t)k * nxs * nxs, Bb + (size_t)k * nxs * NU);
        for (int i = 0; i < nxs; i++) bb[(size_t)k * nxs + i] = xn[i] - o->X[(size_t)(k + 1) * nxs + i];
    }
    /* 4./5. condensing, dx_k = G_k v + g_k */
    for (int i = 0; i < nxs; i++) g[i] = o->x0[i] - o->X[i];
    for (int k = 0; k < N; k++) {
        const double *A = Ab + (size_t)k * nxs * nxs, *B = Bb + (size_t)k * nxs * NU;
        double *Gn = G + (size_t)(k + 1) * nxs * nv, *Gk = G + (size_t)k * nxs * nv;
        for (int i = 0; i < nxs; i++) {
            double acc = bb[(size_t)k * nxs + i];
            for (int l = 0; l < nxs; l++) {
                const double a = A[i * nxs + l];
                if (a == 0.0) continue;
                acc += a * g[(size_t)k * nxs + l];
                for (int j = 0; j < NU * k; j++) Gn[(size_t)i * nv + j] += a * Gk[(size_t)l * nv + j];
            }
            for (int j = 0; j < NU; j++) Gn[(size_t)i * nv + NU * k + j] = B[i * NU + j];
            g[(size_t)(k + 1) * nxs + i] = acc;
        }
    }
    /* 2. Gauss-Newton cost on the nominal copy, y = [x, y, wrap(yaw), |v|, u] */
    for (int k = 0; k <= N; k++) {
        const double sc = (k < N) ? dt : 1.0;
        const double *x = o->X + (size_t)k * nxs, *yr = o->yref + k * 6, *W = o->W + k * 6;
        const double *Gk = G + (size_t)k * nxs * nv, *gk = g + (size_t)k * nxs;
        const double vabs = sqrt(x[3] * x[3] + x[4] * x[4]);
        const double y[4] = {x[0], x[1], wrap_yaw(x[2]), vabs};
        for (int r = 0; r < 4; r++) {
            const double w = sc * W[r];
            double res = y[r] - yr[r];
            if (r < 3) {
                res += gk[r];
                for (int j = 0; j < nv; j++) row[j] = Gk[(size_t)r * nv + j];
            } else {
                const double cl = (vabs > 0.0) ? x[3] / vabs : 0.0, ct = (vabs > 0.0) ? x[4] / vabs : 0.0;
                res += cl * gk[3] + ct * gk[4];
                for (int j = 0; j < nv; j++) row[j] = cl * Gk[(size_t)3 * nv + j] + ct * Gk[(size_t)4 * nv + j];
            }
            for (int j = 0; j < NU * k; j++) {
                if (row[j] == 0.0) continue;
                q[j] += w * res * row[j];
                for (int l = 0; l <= j; l++) H[j * nv + l] += w * row[j] * row[l];
            }
        }
        if (k < N)
            for (int r = 0; r < NU; r++) {
                const double w = sc * W[4 + r];
                H[(NU * k + r) * nv + NU * k + r] += w;
                q[NU * k + r] += w * (o->U[k * NU + r] - yr[4 + r]);
            }
    }
    for (int j = 0; j < nv; j++) for (int l = 0; l < j; l++) H[l * nv + j] = H[j * nv + l];
    /* 3. constraints, rows [bu_k k=0..N-1 | (bx_k, h_k) k=1..N] */
    for (int k = 0; k < N; k++) {
        C[k * nv + NU * k + 1] = 1.0;
        d[k] = o->U[k * NU + 1];
        lb[k] = o->lbu[k]; ub[k] = o->ubu[k];
        zl[k] = dt * o->zl[k * 3]; zu[k] = dt * o->zu[k * 3]; Zl[k] = dt * o->Zl[k * 3]; Zu[k] = dt * o->Zu[k * 3];
    }
    double *gh = calloc(nxs, sizeof(double));
    for (int k = 1; k <= N; k++) {
        const double sc = (k < N) ? dt : 1.0;
        const double *x = o->X + (size_t)k * nxs;
        const double *Gk = G + (size_t)k * nxs * nv, *gk = g + (size_t)k * nxs;
        const int ib = N + 2 * (k - 1), ih = ib + 1;
        for (int j = 0; j < NU * k; j++) C[ib * nv + j] = Gk[(size_t)6 * nv + j];
        d[ib] = x[6] + gk[6];
        lb[ib] = o->lbx[k]; ub[ib] = o->ubx[k];
        const double h = snmpc_h(o, k, x, gh);
        o->hval[k] = h;
        double acc = h;
        for (int l = 0; l < nxs; l++) acc += gh[l] * gk[l];
        d[ih] = acc;
        for (int l = 0; l < nxs; l++) {
            if (gh[l] == 0.0) continue;
            for (int j = 0; j < NU * k; j++) C[ih * nv + j] += gh[l] * Gk[(size_t)l * nv + j];
        }
        lb[ih] = o->lh[k]; ub[ih] = o->uh[k];
        for (int s = 1; s <= 2; s++) {
            const int i = (s == 1) ? ib : ih;
            zl[i] = sc * o->zl[k * 3 + s]; zu[i] = sc * o->zu[k * 3 + s];
            Zl[i] = sc * o->Zl[k * 3 + s]; Zu[i] = sc * o->Zu[k * 3 + s];
        }
    }
    if (g_sdbg) {
        double *p = g_sdbg;
        memcpy(p, H, sizeof(double) * nv * nv); p += nv * nv;
        memcpy(p, q, sizeof(double) * nv); p += nv;
        memcpy(p, C, sizeof(double) * m * nv); p += m * nv;
        memcpy(p, d, sizeof(double) * m);
    }
    /* 6. QP */
    double *sall = calloc(2 * m, sizeof(double));
    ipm_info info;
    double *swarm = NULL, *lwarm = NULL;
    if (o->ipm.warm > 0 && o->have_qp) {          /* (as in oracle_solve) */
        swarm = malloc(sizeof(double) * 4 * m); lwarm = swarm + 2 * m;
        memcpy(swarm, o->sl, sizeof(double) * m); memcpy(swarm + m, o->su, sizeof(double) * m);
        memcpy(lwarm, o->lam, sizeof(double) * 2 * m);
    }
    qp_ipm(nv, m, H, q, C, d, lb, ub, zl, zu, Zl, Zu, &o->ipm, v, sall, o->lam, &info, swarm, lwarm);
    free(swarm);
    memcpy(o->sl, sall, sizeof(double) * m);
    memcpy(o->su, sall + m, sizeof(double) * m);
    o->qp_iter = info.iter;
    o->have_qp = (info.status == 0);
    o->res[0] = info.res_stat; o->res[1] = info.res_ineq; o->res[2] = info.res_comp;
    o->status = (info.status == 0 || info.status == 1) ? 0 : 4;
    /* 7. full step on all copies */
    if (o->status == 0) {
        for (int k = 0; k <= N; k++) {
            const double *Gk = G + (size_t)k * nxs * nv;
            for (int i = 0; i < nxs; i++) {
                double acc = g[(size_t)k * nxs + i];
                for (int j = 0; j < NU * k; j++) acc += Gk[(size_t)i * nv + j] * v[j];
                o->X[(size_t)k * nxs + i] += acc;
            }
        }
        for (int j = 0; j < nv; j++) o->U[j] += v[j];
    }
    o->cost = snmpc_eval_cost(o);
    free(Ab); free(Bb); free(bb); free(G); free(g); free(H); free(q); free(C); free(d); free(lb); free(v);
    free(row); free(gh); free(sall);
    return o->status;
}
