// nmpc_device.hpp -- CDNA4 (gfx950) device code of the batched SQP-RTI solver.
//
// One wavefront (64 lanes, one 64-thread workgroup) solves one OCP instance:
//   phase 1  linearise   lane = shooting stage: ERK4 x nsub with hand-derived forward sensitivities
//                        of the single-track/Pacejka ODE (pred_model_dynamic_stm_pacejka.py:118-177)
//   phase 2  condense    lane = column of G (dx_k = G_k dU + g_k); the Gauss-Newton Hessian is a SYRK
//                        over the 4 cost rows of every stage, accumulated with v_mfma_f64_16x16x4_f64
//                        into 15 register-resident 16x16 tiles (upper triangle of the 80x80 H)
//   phase 3  dense IPM   Mehrotra predictor-corrector on the condensed soft-constrained QP; per
//                        iteration  M = H + C' Gamma C  (MFMA SYRK into the packed lower triangle in
//                        LDS), LDL' over 4-column micro-panels held as row-panel register tiles (MFMA
//                        rank-4 trailing updates), in-place inverses of the five 16x16 diagonal blocks,
//                        then two block substitutions on 16x16 register tiles: a vector block is kept
//                        replicated over the four DPP rows, the four partial sums of a row meet in a
//                        quad_sum of gfx950 row swaps (v_permlane16_swap / v_permlane32_swap) and the
//                        new unknowns reach the lanes that need them through ds_bpermute gathers.
//                        The row state (s, t, lam, mu and residuals of the six soft rows of a stage)
//                        stays in registers, two slots x two sides per lane
//   phase 4  expand      dx trajectory, full step, cost at the new iterate
// Everything between the initial loads and the final stores lives in LDS / registers.
//
// Reference semantics: NMPC_STM_acados_settings.py:16-245 (OCP), SURVEY.md Appendix B (one RTI call).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace tum {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NX = 8, NU = 2;
constexpr int NMAX = 40;                    // max horizon of this build
constexpr int NT = 5;                       // 16-wide MFMA tiles per dimension
constexpr int NVP = 16 * NT;                // 80 (padded) condensed variables
constexpr int NTT = NT * (NT + 1) / 2;      // 15 upper-triangular tiles
constexpr int ABS = 53;                     // LDS stride (doubles) of one stage's linearisation record
constexpr int LPK = NVP * (NVP + 1) / 2;    // 3240 packed lower-triangular entries

__host__ __device__ constexpr int tidx(int K, int I) { return K * NT - K * (K - 1) / 2 + (I - K); }   // K <= I
__host__ __device__ constexpr int lpk(int i, int j) { return i * (i + 1) / 2 + j; }                    // i >= j
// packed start of row R0 + r for a compile-time R0 and a small lane value r whose r (r + 1) / 2 is at hand: one 24-bit multiply-add
// (lpk on a lane value costs a full-width integer multiply -- quarter rate -- and a signed halving)
__device__ __forceinline__ int lpk_row(int R0, int r, int tri_r) { return R0 * (R0 + 1) / 2 + (int)__umul24(R0, r) + tri_r; }
// packed gg-constraint rows: stage s (1..N) owns one row (grad h' G_s) of 2s entries.
// (The steering-angle rows need no storage: delta is a pure integrator of the steering rate, so
//  row s of that block is dt on the odd (steering-rate) columns < 2s and 0 elsewhere.)
__host__ __device__ constexpr int hoff(int s) { return s * (s - 1); }
// the same rows as MFMA operands: (chunk c = rows 4c+1..4c+4, tile column T) pairs with c >= 2T, 30 of them
constexpr int NCHV = 30;
__host__ __device__ constexpr int chidx(int c, int T) { return (T == 0 ? 0 : T == 1 ? 8 : T == 2 ? 14 : T == 3 ? 18 : 20) + c; }

// LDS carve (offsets in doubles). FOUR workgroups per CU (one wavefront on every SIMD) need <= 40 KiB each:
// only the KKT matrix, the packed gg rows and a handful of vectors stay in LDS during the interior point
// loop. Everything that is alive only before the loop (linearisation records, condensing scratch, the
// iterate) is aliased into those regions; the records are parked in HBM/L2 (KArgs::ws) while the IPM runs
// and come back for the final expansion; the IPM row state lives in registers; the pivots d_j sit on the
// diagonal of the (unit-lower) factor.
constexpr int O_PEN = 0;                              // 36                    slack penalties [class][slot][zl,zu,Zl,Zu]
//   (first on purpose: the diagonal-block substitution reads up to 15 doubles in front of a packed row with a zero
//    multiplier; in front of row 0 that lands here, on finite data)
constexpr int O_M = O_PEN + 36;                       // LPK                   KKT matrix / L D L' factor (d on the diagonal)
//   aliased into the M region (dead before the first KKT assembly, reloaded after the last):
constexpr int O_AB = O_M;                             //     NMAX*ABS    compact (Sp,S,b) per stage
constexpr int O_STAGE = O_AB + NMAX * ABS;            //     4 x NVP     staging rows for the H SYRK
constexpr int O_RES = O_STAGE + 4 * NVP;              //     (NMAX+1)*4  y - yref of the 4 state cost rows
constexpr int O_GH = O_RES + (NMAX + 1) * 4;          //     (NMAX+1)*4  (gh3, gh5, gh7, h) per stage
constexpr int O_D = O_GH + (NMAX + 1) * 4;            //     2*NMAX      constant term of the general rows
constexpr int O_G = O_D + 2 * NMAX;                   //     (NMAX+1)*8  g_k (constant part of dx_k)
constexpr int O_CH = O_M + LPK;                       // NMAX*(NMAX+1)         packed h rows
//   aliased into the h-row region (before condensing writes it / after the IPM):
constexpr int O_X = O_CH;                             //     (NMAX+1)*8  iterate X
constexpr int O_U1 = O_X + (NMAX + 1) * NX;           //     NVP         iterate U (after the IPM)
constexpr int O_GAMH = O_CH + NMAX * (NMAX + 1);      // NMAX                  gamma of the h rows
constexpr int O_WH = O_GAMH;                          //   (same buffer: gamma while assembling, weights while forming the rhs)
constexpr int O_WB = O_WH + NMAX;                     // NMAX                  box-row scalars (gamma / weights)
constexpr int O_SFX = O_WB + NMAX;                    // NMAX+2                suffix sums over the steering-angle rows
constexpr int O_XD = O_SFX;                           //   (before the IPM: steering angle of every stage of the iterate)
constexpr int O_DV = O_SFX + NMAX + 2;                // NVP                   broadcast copy of a v-space vector
constexpr int O_U0 = O_DV;                            //   (before the IPM: iterate U)
constexpr int O_DUMMY = O_DV + NVP;                   // 1                     target of masked-off stores
constexpr int LDS_DOUBLES = O_DUMMY + 1;
constexpr int LDS_BYTES = LDS_DOUBLES * 8;
constexpr int WS_DOUBLES = NMAX * ABS;                // per-instance HBM workspace (parked linearisation records)
static_assert(O_G + (NMAX + 1) * NX <= O_CH, "aliased condensing scratch must fit inside the KKT matrix region");
static_assert(LDS_BYTES * 4 <= 160 * 1024, "four workgroups per CU");

struct Model {
    double lf, lr, inv_m, inv_Iz, m, ka;            // ka = 0.5*ro*S*Cd
    double Bf, Cf, Df, Ef, Br, Cr, Dr, Er;
    double Fz_f, Fz_r, invFmax_f, invFmax_r;
    double fr0, fr1, fr4;
    double ax_brake;                                  // -acc_min
    int n_ggv;
    double ggv_v[16], ggv_ax[16], ggv_ay[16];
};

struct KArgs {
    int N, nsub, batch, flags;                        // flags: 1 store_qp_in, 2 debug dump, 4 phase timers
    double dt;
    int iter_max;
    double tol_stat, tol_ineq, tol_comp, mu0, t0, reg;
    Model mp;
    double *X, *U;                                    // iterate, [b][(N+1)*8], [b][N*2]
    const double *x0, *yref, *W, *pen, *bnd;          // [b][8], [b][(N+1)*6], [b][10], [b][36], [b][6][N+1]
    double *cost, *res, *slack;                       // [b], [b][3], [b][6N]
    int *status, *qp_iter, *qp_status;                // [b]
    double *qpin;                                     // [b][N][88]  (A 64 | B 16 | b 8), row-major
    double *dbg;                                      // debug dump, instance 0.. (flags&2)
    int dbg_stride;
    long long *prof;                                  // [b][12] phase cycle counters (flags&4)
    double *ws;                                       // [b][WS_DOUBLES] linearisation records parked during the IPM
    const int *order;                                 // [batch] workgroup -> instance map (longest-first schedule) or null
    // coupled SNMPC OCP only (nmpc_rti_kernel<., true>, snmpc_kernels.hpp)
    int uph;                                          // uncertainty propagation horizon: stages 1..uph come from `pro`
    const double *pro;                                // [b][uph][sn_pro_stage(uph)] G_nom,s | chance-constraint row of stage s
    double *dv;                                       // [b][NVP] QP solution for the epilogue kernel
    // warm start of the interior point method (pipeline kernels): per instance the multipliers of the last QP in the layout of `slack`,
    // then 1.0 where that QP converged / 0.0 (one pointer: it is live across the whole kernel); the complementarity target of a warm
    // start (0: always cold)
    double *qp_lam;                                   // [b][6N + 2]
    double warm_mu;
    // safeguard of the warm start: used only when at most warm_flips row sides changed their activity between the previous QP's solution
    // and the new problem and no row is violated by more than warm_viol beyond its old slack (warm_flips < 0: always)
    int warm_flips; double warm_viol;
    // a full symmetric W per stage (round 6; acados' cost_set(i, 'W', W) takes any matrix, NMPC_class.py:290-296): [b][N+1][36] row-major, stage N its
    // leading 4 x 4 -- or null: the diagonal in W above. The pipeline then condenses with cond_wide_kernel<., false, true>.
    const double *Wf;
};

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ double rl(double v, int lane)   // broadcast lane `lane` (wave-uniform index)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// ---- wavefront scans / reductions on DPP (row_shr 1,2,4,8 then row_bcast 15/31: the gfx9 wave64 scan
//      sequence). No LDS round trips, unlike __shfl: a 64-lane scan costs ~12 DPP moves + 6 ops.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double ident, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// (row_shr steps with bound_ctrl: a lane whose source is outside its row of 16 receives 0 -- the identity of both scans used
//  here, sums and maxima of non-negative values -- without a v_mov that pre-loads the identity into the destination)
template <int CTRL>
__device__ __forceinline__ double dpp0_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
#define TUM_DPP_SCAN(OP, IDENT)                                   \
    v = OP(v, dpp0_f64<0x111>(v));                                \
    v = OP(v, dpp0_f64<0x112>(v));                                \
    v = OP(v, dpp0_f64<0x114>(v));                                \
    v = OP(v, dpp0_f64<0x118>(v));                                \
    v = OP(v, dpp_f64<0x142, 0xa>(IDENT, v));                     \
    v = OP(v, dpp_f64<0x143, 0xc>(IDENT, v));
__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
__device__ __forceinline__ double op_max(double a, double b) { return fmax(a, b); }
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double wave_prefix(double v, int /*lane*/)
{
    TUM_DPP_SCAN(op_add, 0.0)
    return v;
}
// inclusive suffix sum: reverse, prefix, reverse

__device__ __forceinline__ double wave_sum(double v)
{
    TUM_DPP_SCAN(op_add, 0.0)
    return rl(v, 63);
}
// value of lane l-K (row_shr) / l+K (row_shl) inside each row of 16 lanes, 0.0 where that lane is outside the row
template <int K> __device__ __forceinline__ double row_shr(double v) { return dpp0_f64<0x110 + K>(v); }
template <int K> __device__ __forceinline__ double row_shl(double v) { return dpp0_f64<0x100 + K>(v); }
// inclusive suffix sum over the wavefront: a suffix scan inside each row of 16 lanes (row_shl 1, 2, 4, 8), then the totals of the
// later rows (lanes 16, 32, 48 hold them) through scalar registers. (DPP has no backward counterpart of row_bcast 15 / 31, and
// reversing the lanes around the forward scan costs two trips through the LDS crossbar on the critical path.)
__device__ __forceinline__ double wave_suffix(double v, int lane)
{
    v += row_shl<1>(v); v += row_shl<2>(v); v += row_shl<4>(v); v += row_shl<8>(v);
    const double r1 = rl(v, 16), r2 = rl(v, 32), r3 = rl(v, 48);
    const int q = lane >> 4;
    double c = (q < 3) ? r3 : 0.0;
    c += (q < 2) ? r2 : 0.0;
    c += (q < 1) ? r1 : 0.0;
    return v + c;
}
// compile-time loop: f(integral_constant<int, K>) for K = LO..HI
template <int LO, int HI, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (LO <= HI) { f(std::integral_constant<int, LO>()); static_for<LO + 1, HI>(f); }
}
// max of NON-NEGATIVE values (identity 0)
// ---- cross-lane moves for vectors replicated over the four DPP rows of a wavefront (lane = (q, c) = (l >> 4, l & 15)).
// quad_sum: sum over the four lanes (0..3, c) that share a column index c, result in all four; gfx950 row swaps
// (v_permlane16_swap / v_permlane32_swap, VALU, no LDS), measured 105 cycles as a dependent step (scripts/probes/probe_permlane.cpp).
typedef unsigned int tum_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double quad_sum(double v)
{
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    tum_u32x2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    tum_u32x2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(s); hi = __double2hiint(s);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// lane_gather: value of the lane whose byte address (4 * lane id) is `addr4` (ds_bpermute: the LDS crossbar, no LDS memory;
// 82 cycles as a dependent step)
__device__ __forceinline__ double lane_gather(double v, int addr4)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v)
{
    TUM_DPP_SCAN(op_max, 0.0)
    return rl(v, 63);
}
// 1/x without the IEEE division sequence: v_rcp_f64 (4.4e-8 relative, measured) + ONE Newton step -> 2e-15
// relative (measured over 40 decades, scripts/probes/probe_trisolve2.cpp); 3 VALU instead of ~27. The IPM only
// uses it inside Newton-type iterations, which are self-correcting at that level.
__device__ __forceinline__ double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
// value of lane l (a compile-time constant) as a wave-uniform scalar
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
// four independent 4x4x4 products, one per 4-lane block of a DPP row: D_blk[i][j] = sum_k A_blk[i][k] B_blk[k][j] with
// A[i][k] on lane 16 k + 4 blk + i, B[k][j] on lane 16 k + 4 blk + j, D[i][j] on lane 16 i + 4 blk + j (one double per lane)
__device__ __forceinline__ double mfma4(double a, double b) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0); }
__device__ __forceinline__ double mfma4a(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// rotation by N lanes inside every row of 16 lanes: lane (q, c) takes the value of lane (q, (c - N) mod 16)   (DPP row_ror)
// value of lane K of the DPP quad (4 consecutive lanes) in all four
template <int K> __device__ __forceinline__ double quad_bcast(double v) { return dpp0_f64<0x55 * K>(v); }
template <int N> __device__ __forceinline__ double row_ror(double v) { return dpp0_f64<0x120 + N>(v); }    // (every lane has a source)
// LDS ordering point between the lanes of the ONE wavefront of a workgroup. The LDS executes a wave's DS
// instructions in issue order, so a later ds_read already observes an earlier ds_write of another lane: no
// s_barrier and no counter drain are needed, only a fence that stops the compiler from reordering across it.
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sin and cos together for moderate arguments (|x| < ~1e5: tyre-model angles, steering angle, yaw): two-constant
// Cody-Waite reduction by pi/2 with FMAs, then the classic minimax kernels on [-pi/4, pi/4] (coefficients of the
// fdlibm k_sin / k_cos kernels, < 1 ulp there). About a third of the instructions of the general-range library routine,
// which matters because the linearisation evaluates it 48 times per lane.
__device__ __forceinline__ void fast_sincos(double x, double *sn, double *cs)
{
    const double n = rint(x * 6.36619772367581382433e-01);              // x * 2/pi
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    const int q = (int)n & 3;
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                       2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                       -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const double s_ = (q & 1) ? c : s, c_ = (q & 1) ? s : c;
    *sn = (q & 2) ? -s_ : s_;
    *cs = ((q + 1) & 2) ? -c_ : c_;
}

// atan for any finite argument: three ranges (|x| <= tan(pi/8): x itself; <= tan(3pi/8): (|x|-1)/(|x|+1) + pi/4; beyond:
// -1/|x| + pi/2), reduced argument |t| <= 0.4143 where the fdlibm atan polynomial holds (< 1 ulp); the quotient is a
// reciprocal with one residual correction. Half the instructions of the library routine (85), 72 calls per lane.
__device__ __forceinline__ double fast_atan(double x)
{
    const double ax = fabs(x);
    const bool big = ax > 2.41421356237309492343e+00, mid = ax > 4.14213562373095034458e-01;
    const double num = big ? -1.0 : (mid ? ax - 1.0 : ax);
    const double den = big ? ax : (mid ? ax + 1.0 : 1.0);
    const double hi = big ? 1.57079632679489655800e+00 : (mid ? 7.85398163397448278999e-01 : 0.0);
    const double lo = big ? 6.12323399573676603587e-17 : (mid ? 3.06161699786838301793e-17 : 0.0);
    const double q = frcp(den);
    double t = num * q;
    t = fma(fma(-den, t, num), q, t);
    const double z = t * t, w = z * z;
    const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02),
                          6.66107313738753120669e-02), 9.09088713343650656196e-02), 1.42857142725034663711e-01), 3.33333333333329318027e-01);
    const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02),
                          -7.69187620504482999495e-02), -1.11111104054623557880e-01), -1.99999999998764832476e-01);
    const double r = hi - ((t * (s1 + s2) - lo) - t);
    return copysign(r, x);
}
// sqrt of a strictly positive normal number: v_rsq_f64 seed and two coupled Goldschmidt / Newton corrections (< 1 ulp
// measured); a quarter of the instructions of the IEEE sequence. Not for 0, denormals or infinities.
__device__ __forceinline__ double fast_sqrt_pos(double x)
{
    const double r = __builtin_amdgcn_rsq(x);
    double y = x * r, h = 0.5 * r;
    const double e = fma(-h, y, 0.5);
    y = fma(y, e, y); h = fma(h, e, h);
    return fma(fma(-y, y, x), h, y);
}

// ---------------------------------------------------------------- model
// Core of the single-track ODE: derivatives of (vlong, vlat, yawrate) and their partials w.r.t.
// (vl, vt, r, delta, a). pred_model_dynamic_stm_pacejka.py:118-175; derivative conventions follow
// CasADi (if_else / fmin / fmax: derivative of the taken branch).
__device__ __forceinline__ void pacejka(double B, double C, double D, double E, double al, double &Fy, double &dFy)
{
    const double x1 = B * al;
    const double at1 = fast_atan(x1);
    const double inner = x1 - E * (x1 - at1);
    const double th = fast_atan(inner);
    double sn, cs;
    fast_sincos(C * th, &sn, &cs);
    Fy = D * sn;
    dFy = D * cs * C * frcp(1.0 + inner * inner) * (1.0 - E + E * frcp(1.0 + x1 * x1)) * B;
}

__device__ __forceinline__ void stm_core(const Model &p, double vl, double vt, double r, double de, double a,
                                         double f[3], double J[3][5])
{
    const double vv = vl * vl + vt * vt;
    const double sp = fast_sqrt_pos(vv);
    const double w = 0.036 * sp;                       // v[km/h] / 100
    const double w2 = w * w;
    const double fr = p.fr0 + p.fr1 * w + p.fr4 * w2 * w2;
    const double dfr_dw = p.fr1 + 4.0 * p.fr4 * w2 * w;
    const double isp = 0.036 * frcp(sp);
    const double fr_vl = dfr_dw * isp * vl, fr_vt = dfr_dw * isp * vt;
    const double Fxf = -fr * p.Fz_f;
    const double Fxr = p.m * a - fr * p.Fz_r;
    double alf = 0, alf_vl = 0, alf_vt = 0, alf_r = 0, alf_de = 0, alr = 0, alr_vl = 0, alr_vt = 0, alr_r = 0;
    if (vl > 0.001) {
        const double ivl = frcp(vl);
        const double qf = (vt + p.lf * r) * ivl, cf2 = frcp(1.0 + qf * qf);
        alf = de - fast_atan(qf);
        alf_vl = qf * ivl * cf2; alf_vt = -ivl * cf2; alf_r = -p.lf * ivl * cf2; alf_de = 1.0;
        const double qr = (p.lr * r - vt) * ivl, cr2 = frcp(1.0 + qr * qr);
        alr = fast_atan(qr);
        alr_vl = -qr * ivl * cr2; alr_vt = -ivl * cr2; alr_r = p.lr * ivl * cr2;
    }
    double Fyf_lat, dFyf, Fyr_lat, dFyr;
    pacejka(p.Bf, p.Cf, p.Df, p.Ef, alf, Fyf_lat, dFyf);
    pacejka(p.Br, p.Cr, p.Dr, p.Er, alr, Fyr_lat, dFyr);
    // combined slip weighting, clipped at +-0.98
    double Gf = Fxf * p.invFmax_f, gf_on = 1.0;
    if (Gf > 0.98) { Gf = 0.98; gf_on = 0.0; } else if (Gf < -0.98) { Gf = -0.98; gf_on = 0.0; }
    double Gr = Fxr * p.invFmax_r, gr_on = 1.0;
    if (Gr > 0.98) { Gr = 0.98; gr_on = 0.0; } else if (Gr < -0.98) { Gr = -0.98; gr_on = 0.0; }
    const double cgf = fast_sqrt_pos(1.0 - Gf * Gf), cgr = fast_sqrt_pos(1.0 - Gr * Gr);   // cos(asin(G)), |G| <= 0.98
    const double dcgf = -Gf * frcp(cgf) * gf_on * p.invFmax_f;                // d cgf / d Fxf
    const double dcgr = -Gr * frcp(cgr) * gr_on * p.invFmax_r;                // d cgr / d Fxr
    const double Fxf_vl = -p.Fz_f * fr_vl, Fxf_vt = -p.Fz_f * fr_vt;
    const double Fxr_vl = -p.Fz_r * fr_vl, Fxr_vt = -p.Fz_r * fr_vt;
    const double Fyf = Fyf_lat * cgf, Fyr = Fyr_lat * cgr;
    const double Fyf_vl = dFyf * alf_vl * cgf + Fyf_lat * dcgf * Fxf_vl;
    const double Fyf_vt = dFyf * alf_vt * cgf + Fyf_lat * dcgf * Fxf_vt;
    const double Fyf_r = dFyf * alf_r * cgf;
    const double Fyf_de = dFyf * alf_de * cgf;
    const double Fyr_vl = dFyr * alr_vl * cgr + Fyr_lat * dcgr * Fxr_vl;
    const double Fyr_vt = dFyr * alr_vt * cgr + Fyr_lat * dcgr * Fxr_vt;
    const double Fyr_r = dFyr * alr_r * cgr;
    const double Fyr_a = Fyr_lat * dcgr * p.m;
    double sd, cd;
    fast_sincos(de, &sd, &cd);
    const double im = p.inv_m;
    f[0] = (Fxr - p.ka * vl * vl - Fyf * sd + Fxf * cd) * im + vt * r;
    J[0][0] = (Fxr_vl - 2.0 * p.ka * vl - Fyf_vl * sd + Fxf_vl * cd) * im;
    J[0][1] = (Fxr_vt - Fyf_vt * sd + Fxf_vt * cd) * im + r;
    J[0][2] = -Fyf_r * sd * im + vt;
    J[0][3] = (-Fyf_de * sd - Fyf * cd - Fxf * sd) * im;
    J[0][4] = 1.0;
    const double front = Fyf * cd + Fxf * sd;
    const double fr_vl_ = Fyf_vl * cd + Fxf_vl * sd, fr_vt_ = Fyf_vt * cd + Fxf_vt * sd;
    const double fr_r_ = Fyf_r * cd, fr_de_ = Fyf_de * cd - Fyf * sd + Fxf * cd;
    f[1] = (Fyr + front) * im - vl * r;
    J[1][0] = (Fyr_vl + fr_vl_) * im - r;
    J[1][1] = (Fyr_vt + fr_vt_) * im;
    J[1][2] = (Fyr_r + fr_r_) * im - vl;
    J[1][3] = fr_de_ * im;
    J[1][4] = Fyr_a * im;
    const double iz = p.inv_Iz;
    f[2] = (p.lf * front - p.lr * Fyr) * iz;
    J[2][0] = (p.lf * fr_vl_ - p.lr * Fyr_vl) * iz;
    J[2][1] = (p.lf * fr_vt_ - p.lr * Fyr_vt) * iz;
    J[2][2] = (p.lf * fr_r_ - p.lr * Fyr_r) * iz;
    J[2][3] = p.lf * fr_de_ * iz;
    J[2][4] = -p.lr * Fyr_a * iz;
}

// piecewise-linear gg table (casadi interpolant 'linear', NMPC_class.py:322-335)
__device__ __forceinline__ void interp_lin(int n, const double *xs, const double *ys, double x, double &y, double &dy)
{
    int i = 0;
    while (i < n - 2 && x >= xs[i + 1]) i++;
    const double sl = (ys[i + 1] - ys[i]) / (xs[i + 1] - xs[i]);
    y = ys[i] + sl * (x - xs[i]);
    dy = sl;
}

// h = (a/ax)^2 + (vl*r/ay)^2 and its gradient entries (d/dvl, d/dr, d/da)
// NMPC_STM_acados_settings.py:70-74,108-119 (combined_acc_limits == 2)
__device__ __forceinline__ void h_con(const Model &p, double vl, double r, double a, double &h, double &g3, double &g5, double &g7)
{
    double ax, dax, ay, day;
    interp_lin(p.n_ggv, p.ggv_v, p.ggv_ax, vl, ax, dax);
    interp_lin(p.n_ggv, p.ggv_v, p.ggv_ay, vl, ay, day);
    if (a < 0.0) { ax = p.ax_brake; dax = 0.0; }
    const double alat = vl * r, nlon = a / ax, nlat = alat / ay;
    h = nlon * nlon + nlat * nlat;
    g3 = 2.0 * nlat * (r / ay - alat / (ay * ay) * day) - 2.0 * nlon * a / (ax * ax) * dax;
    g5 = 2.0 * nlat * vl / ay;
    g7 = 2.0 * nlon / ax;
}

__device__ __forceinline__ double wrap_yaw(double yaw)     // NMPC_STM_acados_settings.py:41-42
{
    double y = fmod(yaw, 2.0 * M_PI);
    if (y < 0.0) y += 2.0 * M_PI;
    return y;
}

// One shooting interval with structured forward sensitivities (lane-local).
// Tracked: Sp[2] = d(px,py)/dpsi0 ; S[6][7] = d(px,py,psi,vl,vt,r)/d(vl0,vt0,r0,delta0,a0,jerk,steer_rate).
// delta and a are exact integrators of the inputs; d psi / d psi0 = 1; (px,py) columns are identity.
__device__ __forceinline__ void rk4_sens(const Model &p, const double x0[8], const double u[2], double dt, int nsub,
                                         double xn[8], double Sp[2], double S[6][7])
{
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = x0[i];
    Sp[0] = 0.0; Sp[1] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int c = 0; c < 7; c++) S[i][c] = 0.0;
    S[3][0] = 1.0; S[4][1] = 1.0; S[5][2] = 1.0;
    const double h = dt / nsub;
    for (int sub = 0; sub < nsub; sub++) {
        double xacc[6], kp[6], Spacc[2], Kpp[2], Sacc[6][7], Kp[6][7];
#pragma unroll
        for (int i = 0; i < 6; i++) { xacc[i] = 0.0; kp[i] = 0.0; }
        Spacc[0] = Spacc[1] = 0.0; Kpp[0] = Kpp[1] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int c = 0; c < 7; c++) { Sacc[i][c] = 0.0; Kp[i][c] = 0.0; }
#pragma unroll 1
        for (int st = 0; st < 4; st++) {     // not unrolled: keeps the live ranges of one stage from leaking into the next
            const double ci = (st == 0) ? 0.0 : (st == 3 ? 1.0 : 0.5);
            const double bi = (st == 0 || st == 3) ? (1.0 / 6.0) : (2.0 / 6.0);
            const double ch = ci * h;
            const double tau = sub * h + ch;                 // time since the start of the interval
            const double psi = x[2] + ch * kp[2];
            const double vl = x[3] + ch * kp[3], vt = x[4] + ch * kp[4], r = x[5] + ch * kp[5];
            const double de = x[6] + ch * u[1], a = x[7] + ch * u[0];
            double f[3], J[3][5];
            stm_core(p, vl, vt, r, de, a, f, J);
            double sn, cs;
            fast_sincos(psi, &sn, &cs);
            double k[6];
            k[0] = vl * cs - vt * sn; k[1] = vl * sn + vt * cs; k[2] = r;
            k[3] = f[0]; k[4] = f[1]; k[5] = f[2];
            const double J02 = -k[1], J12 = k[0];            // d(px_dot,py_dot)/dpsi
            // psi0 column: only rows px,py (S[psi][psi0] = 1, core rows 0)
            {
                const double K0 = J02, K1 = J12;
                Spacc[0] += bi * K0; Spacc[1] += bi * K1;
                Kpp[0] = K0; Kpp[1] = K1;
            }
#pragma unroll
            for (int c = 0; c < 7; c++) {
                double s[6];
#pragma unroll
                for (int i = 0; i < 6; i++) s[i] = S[i][c] + ch * Kp[i][c];
                const double sde = (c == 3 ? 1.0 : 0.0) + (c == 6 ? tau : 0.0);
                const double sa = (c == 4 ? 1.0 : 0.0) + (c == 5 ? tau : 0.0);
                double K[6];
                K[0] = J02 * s[2] + cs * s[3] - sn * s[4];
                K[1] = J12 * s[2] + sn * s[3] + cs * s[4];
                K[2] = s[5];
#pragma unroll
                for (int i = 0; i < 3; i++)
                    K[3 + i] = J[i][0] * s[3] + J[i][1] * s[4] + J[i][2] * s[5] + J[i][3] * sde + J[i][4] * sa;
#pragma unroll
                for (int i = 0; i < 6; i++) { Sacc[i][c] += bi * K[i]; Kp[i][c] = K[i]; }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) { xacc[i] += bi * k[i]; kp[i] = k[i]; }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] += h * xacc[i];
        x[6] += h * u[1]; x[7] += h * u[0];
        Sp[0] += h * Spacc[0]; Sp[1] += h * Spacc[1];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int c = 0; c < 7; c++) S[i][c] += h * Sacc[i][c];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) xn[i] = x[i];
}

// ---- the same interval with the work of ONE item spread over eight lanes (lin_cols_kernel, snmpc_lin_cols_kernel: the latency
// path for small batches): one sensitivity column per lane, the transcendental chains of the model split over
// each DPP quad.
struct TyreLane { double B, C, D, E, invFmax, Fz, lsgn; bool front; int qrole; };

// the lane's share of the model inside its DPP quad: lane 0 front tyre, 1 rear tyre, 2 sin / cos psi, 3 sin / cos delta (lanes 2, 3
// repeat a tyre chain whose result is not used)
__device__ __forceinline__ TyreLane tyre_lane(const Model &mp, int lane_in_group)
{
    TyreLane t;
    t.qrole = lane_in_group & 3; t.front = !(lane_in_group & 1);
    t.B = t.front ? mp.Bf : mp.Br; t.C = t.front ? mp.Cf : mp.Cr; t.D = t.front ? mp.Df : mp.Dr; t.E = t.front ? mp.Ef : mp.Er;
    t.invFmax = t.front ? mp.invFmax_f : mp.invFmax_r; t.Fz = t.front ? mp.Fz_f : mp.Fz_r; t.lsgn = t.front ? -mp.lf : mp.lr;
    return t;
}

// stm_core with the quad's division of labour; also returns sin / cos of psi (quad lane 2)
__device__ __forceinline__ void stm_core_quad(const Model &p, const TyreLane &t, double vl, double vt, double r, double de, double a,
                                              double psi, double f[3], double J[3][5], double &sn_psi, double &cs_psi)
{
    const double vv = vl * vl + vt * vt;
    const double sp = fast_sqrt_pos(vv);
    const double w = 0.036 * sp;                       // v[km/h] / 100
    const double w2 = w * w;
    const double fr = p.fr0 + p.fr1 * w + p.fr4 * w2 * w2;
    const double dfr_dw = p.fr1 + 4.0 * p.fr4 * w2 * w;
    const double isp = 0.036 * frcp(sp);
    const double fr_vl = dfr_dw * isp * vl, fr_vt = dfr_dw * isp * vt;
    const double Fxf = -fr * p.Fz_f;
    const double Fxr = p.m * a - fr * p.Fz_r;
    // ---- this lane's tyre
    double al = 0, al_vl = 0, al_vt = 0, al_r = 0, al_de = 0;
    if (vl > 0.001) {
        const double ivl = frcp(vl);
        const double nf = vt + p.lf * r, nr = p.lr * r - vt;
        const double q = (t.front ? nf : nr) * ivl, c2 = frcp(1.0 + q * q);
        const double at = fast_atan(q);
        al = t.front ? de - at : at;
        const double qs = t.front ? q : -q;
        al_vl = qs * ivl * c2; al_vt = -ivl * c2; al_r = t.lsgn * ivl * c2; al_de = 1.0;
    }
    const double x1 = t.B * al;
    const double at1 = fast_atan(x1);
    const double inner = x1 - t.E * (x1 - at1);
    const double th = fast_atan(inner);
    double sn, cs;
    fast_sincos(t.qrole < 2 ? t.C * th : (t.qrole == 2 ? psi : de), &sn, &cs);
    const double Fy_lat = t.D * sn;
    const double dFy = t.D * cs * t.C * frcp(1.0 + inner * inner) * (1.0 - t.E + t.E * frcp(1.0 + x1 * x1)) * t.B;
    // combined slip weighting, clipped at +-0.98
    double G = (t.front ? Fxf : Fxr) * t.invFmax, g_on = 1.0;
    if (G > 0.98) { G = 0.98; g_on = 0.0; } else if (G < -0.98) { G = -0.98; g_on = 0.0; }
    const double cg = fast_sqrt_pos(1.0 - G * G);                               // cos(asin(G)), |G| <= 0.98
    const double dcg = -G * frcp(cg) * g_on * t.invFmax;                        // d cg / d Fx
    const double Fx_vl = -t.Fz * fr_vl, Fx_vt = -t.Fz * fr_vt;
    const double tFy = Fy_lat * cg;
    const double tFy_vl = dFy * al_vl * cg + Fy_lat * dcg * Fx_vl;
    const double tFy_vt = dFy * al_vt * cg + Fy_lat * dcg * Fx_vt;
    const double tFy_r = dFy * al_r * cg;
    const double tX = t.front ? dFy * al_de * cg : Fy_lat * dcg * p.m;          // front: d Fyf / d delta; rear: d Fyr / d a
    // ---- exchange inside the quad
    const double Fyf = quad_bcast<0>(tFy), Fyf_vl = quad_bcast<0>(tFy_vl), Fyf_vt = quad_bcast<0>(tFy_vt), Fyf_r = quad_bcast<0>(tFy_r),
                 Fyf_de = quad_bcast<0>(tX);
    // (the rear tyre's Fy and d Fy / d r go over as their FACTORS: stm_core adds each of these two products to another term --
    //  Fyr + front, Fyr_r + fr_r_ -- which the compiler contracts into one FMA there; a product rounded on the tyre lane would
    //  differ from that in the last bit)
    const double Fyr_lat = quad_bcast<1>(Fy_lat), cgr = quad_bcast<1>(cg), Pr = quad_bcast<1>(dFy * al_r);
    const double Fyr = Fyr_lat * cgr, Fyr_r = Pr * cgr;
    const double Fyr_vl = quad_bcast<1>(tFy_vl), Fyr_vt = quad_bcast<1>(tFy_vt), Fyr_a = quad_bcast<1>(tX);
    sn_psi = quad_bcast<2>(sn); cs_psi = quad_bcast<2>(cs);
    const double sd = quad_bcast<3>(sn), cd = quad_bcast<3>(cs);
    const double Fxf_vl = -p.Fz_f * fr_vl, Fxf_vt = -p.Fz_f * fr_vt;
    const double Fxr_vl = -p.Fz_r * fr_vl, Fxr_vt = -p.Fz_r * fr_vt;
    const double im = p.inv_m;
    f[0] = (Fxr - p.ka * vl * vl - Fyf * sd + Fxf * cd) * im + vt * r;
    J[0][0] = (Fxr_vl - 2.0 * p.ka * vl - Fyf_vl * sd + Fxf_vl * cd) * im;
    J[0][1] = (Fxr_vt - Fyf_vt * sd + Fxf_vt * cd) * im + r;
    J[0][2] = -Fyf_r * sd * im + vt;
    J[0][3] = (-Fyf_de * sd - Fyf * cd - Fxf * sd) * im;
    J[0][4] = 1.0;
    const double front = Fyf * cd + Fxf * sd;
    const double fr_vl_ = Fyf_vl * cd + Fxf_vl * sd, fr_vt_ = Fyf_vt * cd + Fxf_vt * sd;
    const double fr_r_ = Fyf_r * cd, fr_de_ = Fyf_de * cd - Fyf * sd + Fxf * cd;
    f[1] = (Fyr + front) * im - vl * r;
    J[1][0] = (Fyr_vl + fr_vl_) * im - r;
    J[1][1] = (Fyr_vt + fr_vt_) * im;
    J[1][2] = (Fyr_r + fr_r_) * im - vl;
    J[1][3] = fr_de_ * im;
    J[1][4] = Fyr_a * im;
    const double iz = p.inv_Iz;
    f[2] = (p.lf * front - p.lr * Fyr) * iz;
    J[2][0] = (p.lf * fr_vl_ - p.lr * Fyr_vl) * iz;
    J[2][1] = (p.lf * fr_vt_ - p.lr * Fyr_vt) * iz;
    J[2][2] = (p.lf * fr_r_ - p.lr * Fyr_r) * iz;
    J[2][3] = p.lf * fr_de_ * iz;
    J[2][4] = -p.lr * Fyr_a * iz;
}

// rk4_sens with one sensitivity column per lane: col 0..6 = column of S (w.r.t. vl0, vt0, r0, delta0, a0, jerk, steering rate),
// col 7 = the psi0 column (rows px, py move; its psi entry is the constant 1). Sc: the lane's column, rows (px, py, psi, vl, vt, r).
__device__ __forceinline__ void rk4_sens_col(const Model &p, const TyreLane &t, int col, const double x0[8], const double u[2], double dt,
                                             int nsub, double xn[8], double Sc[6])
{
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = x0[i];
#pragma unroll
    for (int i = 0; i < 6; i++) Sc[i] = 0.0;
    Sc[2] = (col == 7) ? 1.0 : 0.0; Sc[3] = (col == 0) ? 1.0 : 0.0; Sc[4] = (col == 1) ? 1.0 : 0.0; Sc[5] = (col == 2) ? 1.0 : 0.0;
    const bool cpsi = col == 7;
    const double h = dt / nsub;
    for (int sub = 0; sub < nsub; sub++) {
        double xacc[6], kp[6], Sacc[6], Kp[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { xacc[i] = 0.0; kp[i] = 0.0; Sacc[i] = 0.0; Kp[i] = 0.0; }
#pragma unroll 1
        for (int st = 0; st < 4; st++) {
            const double ci = (st == 0) ? 0.0 : (st == 3 ? 1.0 : 0.5);
            const double bi = (st == 0 || st == 3) ? (1.0 / 6.0) : (2.0 / 6.0);
            const double ch = ci * h;
            const double tau = sub * h + ch;                 // time since the start of the interval
            const double psi = x[2] + ch * kp[2];
            const double vl = x[3] + ch * kp[3], vt = x[4] + ch * kp[4], r = x[5] + ch * kp[5];
            const double de = x[6] + ch * u[1], a = x[7] + ch * u[0];
            double f[3], J[3][5], sn, cs;
            stm_core_quad(p, t, vl, vt, r, de, a, psi, f, J, sn, cs);
            double k[6];
            k[0] = vl * cs - vt * sn; k[1] = vl * sn + vt * cs; k[2] = r;
            k[3] = f[0]; k[4] = f[1]; k[5] = f[2];
            const double J02 = -k[1], J12 = k[0];            // d(px_dot,py_dot)/dpsi
            {
                double s[6];
#pragma unroll
                for (int i = 0; i < 6; i++) s[i] = Sc[i] + ch * Kp[i];
                const double sde = (col == 3) ? 1.0 : ((col == 6) ? tau : 0.0);
                const double sa = (col == 4) ? 1.0 : ((col == 5) ? tau : 0.0);
                double K[6];
                const double K0 = J02 * s[2] + cs * s[3] - sn * s[4];
                const double K1 = J12 * s[2] + sn * s[3] + cs * s[4];
                K[0] = cpsi ? J02 : K0;                      // (the psi0 column: rows px, py only, S[psi][psi0] = 1)
                K[1] = cpsi ? J12 : K1;
                K[2] = s[5];
#pragma unroll
                for (int i = 0; i < 3; i++)
                    K[3 + i] = J[i][0] * s[3] + J[i][1] * s[4] + J[i][2] * s[5] + J[i][3] * sde + J[i][4] * sa;
#pragma unroll
                for (int i = 0; i < 6; i++) { Sacc[i] += bi * K[i]; Kp[i] = K[i]; }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) { xacc[i] += bi * k[i]; kp[i] = k[i]; }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] += h * xacc[i];
        x[6] += h * u[1]; x[7] += h * u[0];
#pragma unroll
        for (int i = 0; i < 6; i++) Sc[i] += h * Sacc[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) xn[i] = x[i];
}

// w <- A_k w using the compact record (Sp[2] | S[6][7] | b[8]); rows 6,7 of A are identity rows.
__device__ __forceinline__ void apply_A(const double *rec, double w[8])
{
    double n[6];
    n[0] = w[0] + rec[0] * w[2];
    n[1] = w[1] + rec[1] * w[2];
    n[2] = w[2];
    n[3] = 0.0; n[4] = 0.0; n[5] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double *Si = rec + 2 + i * 7;
#pragma unroll
        for (int c = 0; c < 5; c++) n[i] += Si[c] * w[3 + c];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = n[i];
}

// the same for two vectors at once (the two register banks of the condensing phase): every record entry is loaded once
__device__ __forceinline__ void apply_A2(const double *rec, double w[8], double v[8])
{
    const double sp0 = rec[0], sp1 = rec[1];
    double n[6], m[6];
    n[0] = w[0] + sp0 * w[2]; m[0] = v[0] + sp0 * v[2];
    n[1] = w[1] + sp1 * w[2]; m[1] = v[1] + sp1 * v[2];
    n[2] = w[2]; m[2] = v[2];
    n[3] = 0.0; n[4] = 0.0; n[5] = 0.0; m[3] = 0.0; m[4] = 0.0; m[5] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double *Si = rec + 2 + i * 7;
#pragma unroll
        for (int c = 0; c < 5; c++) { const double sv = Si[c]; n[i] += sv * w[3 + c]; m[i] += sv * v[3 + c]; }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { w[i] = n[i]; v[i] = m[i]; }
}

}  // namespace tum
