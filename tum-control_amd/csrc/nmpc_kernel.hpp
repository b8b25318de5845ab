// nmpc_kernel.hpp -- the fused SQP-RTI kernel (one wavefront = one OCP instance).
// DEVELOPMENT BUILD ONLY (-DTUM_DEV_KERNELS -> libtumnmpc_dev.so): round 1's kernel, kept as the second implementation the
// pipeline is held against in the tests. The shipped library is the pipeline alone: this 115 KB kernel sits at the register
// ceiling (129-158 spilled SGPRs; HISTORY.md, "An unexplained build failure") and is not part of the product any more.
#pragma once
#include <type_traits>

#include "common_kernels.hpp"
#include "snmpc_kernels.hpp"

namespace tum {


// Row-side state of the interior point method: every lane owns TWO rows (slots) x two sides (0 lower, 1 upper):
//   lanes 0..N-1   slot 0: steering-rate box of stage l (variable 2l+1)
//                  slot 1: steering-angle bound of stage l+1 (structured row: dt on the odd columns < 2(l+1))
//   lanes 40..59   slots 0,1: gg-circle constraints of stages 2j+1, 2j+2, j = lane-40 (packed rows of sCh)
// (240 row sides over 60 lanes: 4 per lane instead of 6 on 40 lanes.)

// Everything derived from the lane id that the IPM uses. Instantiated from an OPAQUE copy of the lane id inside
// the iteration loop: otherwise LICM hoists hundreds of lane predicates / LDS addresses out of the loop and the
// register allocator spills them (SGPR masks to VGPR lanes, addresses to scratch).
#define TUM_LANE_DEFS \
    const bool boxlane = lane < N; \
    const bool gglane = lane >= NMAX && lane < NMAX + 20; \
    const int gj = gglane ? lane - NMAX : 0; \
    const int stg0 = gglane ? 2 * gj + 1 : lane, stg1 = gglane ? 2 * gj + 2 : lane + 1;   /* stage of each slot */ \
    const bool on0 = gglane ? (stg0 <= N) : boxlane, on1 = gglane ? (stg1 <= N) : boxlane; \
    const int ty0 = gglane ? 2 : 0, ty1 = gglane ? 2 : 1;                                  /* 0 box, 1 steering, 2 gg */ \
    const int pc0 = gglane ? ((stg0 < N) ? 1 : 2) : ((lane == 0) ? 0 : 1), pc1 = (stg1 < N) ? 1 : 2; \
    const double psc0 = (gglane && stg0 >= N) ? 1.0 : dt, psc1 = (stg1 < N) ? dt : 1.0; \
    const int pix0 = (pc0 * 3 + ty0) * 4, pix1 = (pc1 * 3 + ty1) * 4; \
    auto pen = [&](int slot, int sd, int quad) -> double { \
        return (slot == 0 ? psc0 : psc1) * sPen[(slot == 0 ? pix0 : pix1) + 2 * quad + sd]; \
    }; \
    const bool v0on = lane < nv, v1on = (lane < 16) && (64 + lane < nv); \
    const bool odd = lane & 1; \
    const int lane1 = 64 + lc; \
    auto ctw = [&](double &o0, double &o1) {   /* C' w for this lane's columns */ \
        double a0 = (odd && v0on) ? sWb[lane >> 1] + dt * sSfx[(lane >> 1) + 1] : 0.0; \
        double a1 = (odd && v1on) ? sWb[32 + (lane >> 1)] + dt * sSfx[32 + (lane >> 1) + 1] : 0.0; \
        /* gg rows, fully unrolled: one base register, immediate offsets, column j belongs to row s iff j < 2s (a compile \
           time bound per s: no mask at all for s >= 32); rows beyond N are zero-filled after condensing and carry weight 0 */ \
        const double *pc = sCh + lane; \
        double e0 = 0.0, e1 = 0.0, e2 = 0.0; \
    _Pragma("unroll") \
        for (int s = 1; s <= NMAX; s++) { \
            const double c = pc[hoff(s)]; \
            const double t_ = ((lane < 2 * s) ? c : 0.0) * sWh[s - 1]; \
            if ((s & 3) == 0) a0 += t_; else if ((s & 3) == 1) e0 += t_; else if ((s & 3) == 2) e1 += t_; else e2 += t_; \
        } \
        a0 += (e0 + e1) + e2; \
        const double *pc1 = sCh + lane1; \
    _Pragma("unroll") \
        for (int s = 33; s <= NMAX; s++) { \
            const double c = pc1[hoff(s)]; \
            a1 += ((lane1 < 2 * s) ? c : 0.0) * sWh[s - 1]; \
        } \
        o0 = v0on ? a0 : 0.0; o1 = v1on ? a1 : 0.0; \
    }; \
    auto publish = [&](double w0_, double w1_, double *dstH) {   /* slot values: (box, steering) or (gg, gg) */ \
        const double sfx = wave_suffix(boxlane ? w1_ : 0.0, lane); \
        wsync(); \
        if (lane < NMAX) sWb[lane] = boxlane ? w0_ : 0.0; \
        if (lane < NMAX + 1) sSfx[lane + 1] = (lane < N) ? sfx : 0.0; \
        if (gglane) { dstH[2 * gj] = on0 ? w0_ : 0.0; dstH[2 * gj + 1] = on1 ? w1_ : 0.0; } \
        wsync(); \
    }; \
    int rb[NT]; \
    _Pragma("unroll") \
    for (int I = 0; I < NT; I++) rb[I] = lpk(16 * I + lc, 0); \
    const int myrow0 = lpk(lane, 0), myrow1 = lpk(lane1, 0);

// SN = true: the fused kernel of the coupled SNMPC OCP (snmpc_kernels.hpp): the stages 1..uph come condensed from the
// prologue kernel (KArgs::pro), the speed row of the cost is |v| and the gg limits are looked up at |v|.
template <bool PROF, bool SN = false>
__global__ void __launch_bounds__(64, 1) nmpc_rti_kernel(const KArgs ka)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= ka.batch) return;
    const int b = ka.order ? ka.order[blockIdx.x] : (int)blockIdx.x;     // instance solved by this wavefront
    const int N = ka.N, nv = 2 * N;
    const double dt = ka.dt;
    // The scalars of the interior point method are copied into vector registers here. As kernel arguments they would be
    // fetched through the kernel-argument base pointer where the IPM starts, and in the largest instantiation of this
    // kernel exactly those late scalar loads once came out wrong (HISTORY.md, "An unexplained build failure").
    double p_mu0 = ka.mu0, p_t0 = ka.t0, p_reg = ka.reg, p_ts = ka.tol_stat, p_ti = ka.tol_ineq, p_tc = ka.tol_comp;
    int p_itmax = ka.iter_max;
    asm volatile("" : "+v"(p_mu0), "+v"(p_t0), "+v"(p_reg), "+v"(p_ts), "+v"(p_ti), "+v"(p_tc), "+v"(p_itmax));
    const Model &mp = ka.mp;

    double *sAB = lds + O_AB, *sM = lds + O_M, *sStage = lds + O_STAGE, *sCh = lds + O_CH, *sX = lds + O_X;
    double *sG = lds + O_G, *sRes = lds + O_RES, *sGh = lds + O_GH, *sD = lds + O_D, *sGamH = lds + O_GAMH;
    double *sWh = lds + O_WH, *sWb = lds + O_WB, *sSfx = lds + O_SFX, *sDv = lds + O_DV;
    double *sU0 = lds + O_U0, *sU1 = lds + O_U1, *sXd = lds + O_XD, *sPen = lds + O_PEN;

    double *gX = ka.X + (size_t)b * (N + 1) * NX;
    double *gU = ka.U + (size_t)b * N * NU;
    const double *gx0 = ka.x0 + (size_t)b * NX;
    const double *gyref = ka.yref + (size_t)b * (N + 1) * 6;
    // (the fused kernel takes ONE W for the stages < N: stage 0's; per-stage weights are a feature of the pipeline)
    const double *gW = ka.W + (size_t)b * (N + 1) * 6;
    const double *gpen = ka.pen + (size_t)b * 36;
    const double *gbnd = ka.bnd + (size_t)b * 6 * (N + 1);
    const int uph = SN ? ka.uph : 0;
    constexpr int SN_PRO_G = 8 * 64, SN_PRO_STAGE = 9 * 64;        // (uph <= SN_UPHMAX_FUSED: pitch 64, sn_pro_pitch)
    const double *gpro = SN ? ka.pro + (size_t)b * uph * SN_PRO_STAGE : nullptr;

    long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
    // ------------------------------------------------------------ phase 0: loads
    {   // compile-time trip counts: all loads in flight together
        constexpr int NXI = ((NMAX + 1) * NX + 63) / 64;
        double tx[NXI];
#pragma unroll
        for (int j = 0; j < NXI; j++) { const int i = lane + 64 * j; tx[j] = (i < (N + 1) * NX) ? gX[i] : 0.0; }
        const double u0_ = (lane < nv) ? gU[lane] : 0.0, u1_ = (lane < 16 && 64 + lane < nv) ? gU[64 + lane] : 0.0;
#pragma unroll
        for (int j = 0; j < NXI; j++) { const int i = lane + 64 * j; if (i < (N + 1) * NX) sX[i] = tx[j]; }
        sU0[lane] = u0_;
        if (lane < 16) sU0[64 + lane] = u1_;
    }
    if (lane < 36) sPen[lane] = gpen[lane];
    wsync();

    // ------------------------------------------------------------ phase 1: linearise (lane = stage)
    {
        const int k = lane;
        if (k <= N) {
            double yr[6];
#pragma unroll
            for (int i = 0; i < 6; i++) yr[i] = gyref[k * 6 + i];
            double xk[8];
#pragma unroll
            for (int i = 0; i < 8; i++) xk[i] = sX[k * NX + i];
            // cost residual of the 4 state rows: y = [x0, x1, wrap(x2), x3]
            sRes[k * 4 + 0] = xk[0] - yr[0];
            sRes[k * 4 + 1] = xk[1] - yr[1];
            sRes[k * 4 + 2] = wrap_yaw(xk[2]) - yr[2];
            if (SN) {   // speed row |v| (SNMPC_acados_settings.py:153): its gradient (vl, vt)/|v| rides in the unused g slots
                const double vabs = sqrt(xk[3] * xk[3] + xk[4] * xk[4]), iv = (vabs > 0.0) ? 1.0 / vabs : 0.0;
                sRes[k * 4 + 3] = vabs - yr[3];
                sG[k * NX + 6] = xk[3] * iv; sG[k * NX + 7] = xk[4] * iv;
            } else sRes[k * 4 + 3] = xk[3] - yr[3];
            sXd[k] = xk[6];          // the condensing phase overwrites sX (it shares LDS with the packed gg rows)
            if (k >= 1) {
                double h, g3, g5, g7;
                if (SN) {
                    double g4;
                    h_con_vabs(mp, xk[3], xk[4], xk[5], xk[7], h, g3, g4, g5, g7);
                    sG[k * NX + 5] = g4;
                } else h_con(mp, xk[3], xk[5], xk[7], h, g3, g5, g7);
                sGh[k * 4 + 0] = g3; sGh[k * 4 + 1] = g5; sGh[k * 4 + 2] = g7; sGh[k * 4 + 3] = h;
            }
            if (k < N && k >= uph) {
                double uk[2] = {sU0[2 * k], sU0[2 * k + 1]};
                double xn[8], Sp[2], S[6][7];
                rk4_sens(mp, xk, uk, dt, ka.nsub, xn, Sp, S);
                double *rec = sAB + k * ABS;
                rec[0] = Sp[0]; rec[1] = Sp[1];
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int c = 0; c < 7; c++) rec[2 + i * 7 + c] = S[i][c];
#pragma unroll
                for (int i = 0; i < 8; i++) rec[44 + i] = xn[i] - sX[(k + 1) * NX + i];
                if (ka.flags & 1) {   // full A (8x8), B (8x2), b (8) of this linearisation, row-major
                    double *q = ka.qpin + ((size_t)b * N + k) * 88;
                    for (int i = 0; i < 64; i++) q[i] = 0.0;
                    q[0 * 8 + 0] = 1.0; q[1 * 8 + 1] = 1.0; q[2 * 8 + 2] = 1.0; q[6 * 8 + 6] = 1.0; q[7 * 8 + 7] = 1.0;
                    q[0 * 8 + 2] = Sp[0]; q[1 * 8 + 2] = Sp[1];
                    for (int i = 0; i < 6; i++) {
                        for (int c = 0; c < 5; c++) q[i * 8 + 3 + c] = S[i][c];
                        q[64 + i * 2 + 0] = S[i][5]; q[64 + i * 2 + 1] = S[i][6];
                    }
                    q[64 + 6 * 2 + 0] = 0.0; q[64 + 6 * 2 + 1] = dt; q[64 + 7 * 2 + 0] = dt; q[64 + 7 * 2 + 1] = 0.0;
                    for (int i = 0; i < 8; i++) q[80 + i] = rec[44 + i];
                }
            }
        }
    }
    wsync();

    TUM_TICK(0);
    // ------------------------------------------------------------ phase 2: condense
    // bank 0: column `lane` (stage lane>>1, input lane&1); bank 1: column 64+lane for lane < 16,
    // lane 16 of bank 1 carries g_k (the response to dx_0 and the defects b_k).
    const int j0 = lane >> 1, r0 = lane & 1, j1 = 32 + (lane >> 1);
    const bool isg = (lane == 16);
    const int lq = lane >> 4, lc = lane & 15;
    double q0 = 0.0, q1 = 0.0;
    d4 Ht[NTT];
    {
    double Wd[6], We[4];
#pragma unroll
    for (int i = 0; i < 6; i++) Wd[i] = gW[i];
#pragma unroll
    for (int i = 0; i < 4; i++) We[i] = gW[N * 6 + i];
#pragma unroll
    for (int i = 0; i < NTT; i++) Ht[i] = d4{0.0, 0.0, 0.0, 0.0};
    {
        double w0[8], w1[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { w0[i] = 0.0; w1[i] = 0.0; }
        if (isg) {
#pragma unroll
            for (int i = 0; i < 8; i++) { w1[i] = gx0[i] - sX[i]; if (!SN || i < 5) sG[i] = w1[i]; }
        }
        // One condensing stage. The number of 16-column tiles the stage touches (Ts) is a compile-time
        // constant per segment of 8 stages, so every MFMA targets a fixed accumulator (no conditional tiles).
        auto stage_body = [&](const int k, auto tsc) {
            constexpr int Ts = decltype(tsc)::value;
            const double *rec = sAB + k * ABS;
            // Both banks advance together (shared record loads, no exec-mask regions): columns that have not started yet
            // hold zeros, so w <- A_k w leaves them at zero; the column that starts at this stage adds B_k with a 0/1
            // multiplier and the g column adds the defect b_k the same way.
            if (SN && k < uph) {
                // stage s = k+1 <= uph: G_nom,s and g_nom,s are PCE means of the sample recursions (prologue kernel)
                const double *pg = gpro + (size_t)k * SN_PRO_STAGE;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const double gv = pg[i * 64 + lane], gg = pg[i * 64 + 2 * uph];
                    w0[i] = (lane < 2 * uph) ? gv : 0.0;
                    w1[i] = isg ? gg : 0.0;
                }
            } else {
            apply_A2(rec, w0, w1);
            {
                const double sel0 = (j0 == k) ? 1.0 : 0.0, sel1 = (lane < 16 && j1 == k) ? 1.0 : 0.0, selg = isg ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double bc = rec[2 + i * 7 + 5 + r0];
                    w0[i] += sel0 * bc; w1[i] += sel1 * bc;
                }
                const double b6 = r0 ? dt : 0.0, b7 = r0 ? 0.0 : dt;
                w0[6] += sel0 * b6; w0[7] += sel0 * b7; w1[6] += sel1 * b6; w1[7] += sel1 * b7;
#pragma unroll
                for (int i = 0; i < 8; i++) w1[i] += selg * rec[44 + i];
            }
            }
            const int s = k + 1;                         // stage whose G_s the lanes now hold
            const double sc = (s < N) ? dt : 1.0;
            const double g3 = sGh[s * 4 + 0], g5 = sGh[s * 4 + 1], g7 = sGh[s * 4 + 2];
            const double g4 = SN ? sG[s * NX + 5] : 0.0, cvl = SN ? sG[s * NX + 6] : 1.0, cvt = SN ? sG[s * NX + 7] : 0.0;
            double hr0 = g3 * w0[3] + g5 * w0[5] + g7 * w0[7], hr1 = g3 * w1[3] + g5 * w1[5] + g7 * w1[7];
            double hd = sGh[s * 4 + 3];
            if (SN) {
                hr0 += g4 * w0[4]; hr1 += g4 * w1[4];
                if (s < uph) {   // chance-constraint row E + kappa sqrt(Var) over the samples (prologue kernel)
                    const double *pr = gpro + (size_t)k * SN_PRO_STAGE + SN_PRO_G;
                    const double rv = pr[lane], rg = pr[2 * uph];
                    hr0 = (lane < 2 * uph) ? rv : 0.0; hr1 = isg ? rg : 0.0; hd = 0.0;
                }
            }
            // the speed row of the cost: vl (nominal OCP) or |v| (SNMPC: gradient (vl, vt)/|v|)
            const double c30 = SN ? cvl * w0[3] + cvt * w0[4] : w0[3], c31 = SN ? cvl * w1[3] + cvt * w1[4] : w1[3];
            if (isg) {
#pragma unroll
                for (int i = 0; i < 8; i++) if (!SN || i < 5) sG[s * NX + i] = w1[i];
                sD[2 * (s - 1)] = sXd[s] + w1[6];
                sD[2 * (s - 1) + 1] = hd + hr1;
            }
            // gg-constraint row of stage s and staging of the 4 cost rows
            if (lane < 2 * s) sCh[hoff(s) + lane] = hr0;
            if (lane < 16 && 64 + lane < 2 * s) sCh[hoff(s) + 64 + lane] = hr1;
#pragma unroll
            for (int r = 0; r < 3; r++) sStage[r * NVP + lane] = w0[r];
            sStage[3 * NVP + lane] = c30;
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 3; r++) sStage[r * NVP + 64 + lane] = w1[r];
                sStage[3 * NVP + 64 + lane] = c31;
            }
            wsync();
            // gradient: q += sum_r sc*W_r*(res_r + g_s[r]) * G_s[r,:]
            double wr[4];
#pragma unroll
            for (int r = 0; r < 4; r++) wr[r] = sc * ((s < N) ? Wd[r] : We[r]);
            {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double gs = (SN && r == 3) ? cvl * sG[s * NX + 3] + cvt * sG[s * NX + 4] : sG[s * NX + r];
                    const double e = wr[r] * (sRes[s * 4 + r] + gs);
                    a0 += e * ((r == 3) ? c30 : w0[r]); a1 += e * ((r == 3) ? c31 : w1[r]);
                }
                q0 += a0;
                q1 += (lane < 16) ? a1 : 0.0;
            }
            // Gauss-Newton Hessian SYRK (rank-4 update per stage) on the matrix cores
            const double wl = (lq == 0) ? wr[0] : (lq == 1) ? wr[1] : (lq == 2) ? wr[2] : wr[3];
            double aop[Ts], bop[Ts];
#pragma unroll
            for (int T = 0; T < Ts; T++) {
                bop[T] = sStage[lq * NVP + 16 * T + lc];
                aop[T] = bop[T] * wl;
            }
#pragma unroll
            for (int K = 0; K < Ts; K++)
#pragma unroll
                for (int I = K; I < Ts; I++) Ht[tidx(K, I)] = mfma(aop[K], bop[I], Ht[tidx(K, I)]);
            wsync();
        };
        // stage s = k+1 touches columns < 2s, i.e. ceil(s/8) tiles
        for (int k = 0; k < N && k < 8; k++) stage_body(k, std::integral_constant<int, 1>());
        for (int k = 8; k < N && k < 16; k++) stage_body(k, std::integral_constant<int, 2>());
        for (int k = 16; k < N && k < 24; k++) stage_body(k, std::integral_constant<int, 3>());
        for (int k = 24; k < N && k < 32; k++) stage_body(k, std::integral_constant<int, 4>());
        for (int k = 32; k < N; k++) stage_body(k, std::integral_constant<int, 5>());
        // rows of the packed gg block beyond the horizon: zero (they are multiplied by zero weights later; they must be finite)
        for (int s = N + 1; s <= NMAX; s++)
            for (int c = lane; c < 2 * s; c += 64) sCh[hoff(s) + c] = 0.0;
    }
    // input cost (R) and padding on the diagonal, gradient of the input cost
#pragma unroll
    for (int K = 0; K < NT; K++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int row = lq + 4 * jj;
            if (row == lc) {
                const int idx = 16 * K + row;
                Ht[tidx(K, K)][jj] += (idx < nv) ? dt * Wd[4 + (idx & 1)] : 1.0;
            }
        }
    {
        if (lane < nv) q0 += dt * Wd[4 + r0] * (sU0[lane] - gyref[j0 * 6 + 4 + r0]);
        if (lane < 16 && 64 + lane < nv) q1 += dt * Wd[4 + r0] * (sU0[64 + lane] - gyref[j1 * 6 + 4 + r0]);
    }
    }   // Wd, We

    if (PROF && (ka.flags & 2) && b < 4) {   // debug dump of the condensed QP
        double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
#pragma unroll
        for (int K = 0; K < NT; K++)
#pragma unroll
            for (int I = K; I < NT; I++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int row = 16 * K + lq + 4 * jj, col = 16 * I + lc;
                    dbg[row * NVP + col] = Ht[tidx(K, I)][jj];
                    dbg[col * NVP + row] = Ht[tidx(K, I)][jj];
                }
        dbg[6400 + lane] = q0;
        if (lane < 16) dbg[6400 + 64 + lane] = q1;
        for (int s = 1; s <= N; s++)
            for (int c = lane; c < NVP; c += 64) {
                dbg[6480 + (2 * (s - 1)) * NVP + c] = ((c & 1) && c < 2 * s) ? dt : 0.0;
                dbg[6480 + (2 * (s - 1) + 1) * NVP + c] = (c < 2 * s) ? sCh[hoff(s) + c] : 0.0;
            }
        for (int i = lane; i < 2 * N; i += 64) dbg[12880 + i] = sD[i];
        for (int i = lane; i < (N + 1) * NX; i += 64) dbg[12960 + i] = sG[i];
    }

    // park the linearisation records in the HBM/L2 workspace: their LDS space becomes the KKT matrix
    {
        double *ws = ka.ws + (size_t)b * WS_DOUBLES;
        for (int i = lane; i < N * ABS; i += 64) ws[i] = sAB[i];
    }
    TUM_TICK(1);
    // ------------------------------------------------------------ phase 3: interior point
    // Row state lives in registers: rowst[field][slot*2 + side], two row slots and both sides per lane
    // (box rows on lanes 0..39, the gg rows on lanes 40..59, see TUM_LANE_DEFS).
    // fields: 0 s, 1 t, 2 lam, 3 mu, 4 rs (slack stationarity residual), 5 rt (primal residual).
    double v0 = 0.0, v1 = 0.0, rv0, rv1, qn;
    double rowst[6][4];      // IPM row state of this lane: [s, t, lam, mu, rs, rt][slot*2+side]
    const double npairs = 12.0 * N;
    const double inv_npairs = 1.0 / npairs;
    const int nchunk = (N + 3) >> 2;                 // chunks of 4 gg rows
    int it = 0, qp_status = 1;
    double res_stat = 0.0, res_ineq = 0.0, res_comp = 0.0;
    {
        TUM_LANE_DEFS
    {
        const int NB = N + 1;
        double dval[2], lo[2], hi[2];
        {
            // slot 0: box of stage `lane` (value = current steering rate) or gg row of stage stg0
            const int i0 = on0 ? stg0 : (gglane ? 1 : 0), i1 = on1 ? stg1 : 1;
            dval[0] = gglane ? sD[2 * (i0 - 1) + 1] : sU0[2 * i0 + 1];
            lo[0] = gbnd[(gglane ? 4 : 0) * NB + i0]; hi[0] = gbnd[(gglane ? 5 : 1) * NB + i0];
            dval[1] = gglane ? sD[2 * (i1 - 1) + 1] : sD[2 * (i1 - 1)];
            lo[1] = gbnd[(gglane ? 4 : 2) * NB + i1]; hi[1] = gbnd[(gglane ? 5 : 3) * NB + i1];
        }
        // slack-equation-feasible start (same rule as the oracle): s*z = mu0, mu_s from z + Z s - lam - mu_s = 0
        // (floored), t = max(r0 + s, t0), lam = mu0 / t
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                const int k = rr * 2 + sd;
                const bool on = rr ? on1 : on0;
                const double eps = sd ? -1.0 : 1.0, bnd = sd ? hi[rr] : lo[rr];
                const double r0v = eps * (dval[rr] - bnd);
                const double z = pen(rr, sd, 0), Z = pen(rr, sd, 1);
                const double s0 = p_mu0 / (z > 1e-6 ? z : 1e-6);
                double t = r0v + s0;
                if (t < p_t0) t = p_t0;
                const double lam = p_mu0 / t;
                double ms = z + Z * s0 - lam;
                const double msf = 1e-2 * p_mu0 / s0;
                if (ms < msf) ms = msf;
                // slots without a row keep a neutral state (never read into any reduction)
                ROWF(0, k) = on ? s0 : 1.0; ROWF(1, k) = on ? t : 1.0; ROWF(2, k) = on ? lam : 1.0; ROWF(3, k) = on ? ms : 1.0;
                ROWF(4, k) = on ? z + Z * s0 - lam - ms : 0.0;
                ROWF(5, k) = on ? t - r0v - s0 : 0.0;
            }
        wsync();          // the aliased condensing scratch (sD, sU0) has been consumed
    }

    // initial stationarity residual r_v = q - C'(lam_l - lam_u)   (v = 0)
    wsync();
    publish(ROWF(2, 0) - ROWF(2, 1), ROWF(2, 2) - ROWF(2, 3), sWh);
    {
        double c0, c1;
        ctw(c0, c1);
        rv0 = v0on ? q0 - c0 : 0.0; rv1 = v1on ? q1 - c1 : 0.0;
    }
    qn = wave_max(fmax(fabs(q0), (lane < 16) ? fabs(q1) : 0.0));
    if (qn < 1.0) qn = 1.0;
    }
    const int lane_outer = lane;
    for (;; it++) {
        int lane_v = lane_outer;
        asm volatile("" : "+v"(lane_v));          // opaque per iteration (see TUM_LANE_DEFS)
        const int lane = lane_v;
        const int lq = lane >> 4, lc = lane & 15;
        TUM_LANE_DEFS
        // ---- row phase A: residual norms, gamma
        double gap;
        double rD[4], rG[4];       // D = 1/(Z s + mu), G = 1/(t + lam s D) of every row side: fixed until the end of the iteration
        {
            double ls = fmax(fabs(rv0), fabs(rv1)), li = 0.0, lcmp = 0.0, lg = 0.0;
            double gsum[2];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                gsum[rr] = 0.0;
                const bool on = rr ? on1 : on0;
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    const int k = rr * 2 + sd;
                    const double s_ = ROWF(0, k), t_ = ROWF(1, k), l_ = ROWF(2, k), m_ = ROWF(3, k);
                    ls = fmax(ls, on ? fabs(ROWF(4, k)) : 0.0);
                    li = fmax(li, on ? fabs(ROWF(5, k)) : 0.0);
                    const double c1 = t_ * l_, c2 = s_ * m_;
                    lcmp = fmax(lcmp, on ? fmax(c1, c2) : 0.0);
                    lg += on ? c1 + c2 : 0.0;
                    // gamma = 1 / (t/l + 1/(Z + mu/s)) with two reciprocals instead of four: 1/(Z + mu/s) = s / (Z s + mu)
                    rD[k] = frcp(pen(rr, sd, 1) * s_ + m_);
                    rG[k] = frcp(t_ + l_ * s_ * rD[k]);
                    gsum[rr] += l_ * rG[k];
                }
            }
            // convergence: "no lane above its tolerance" is one ballot; the three norms themselves are only reported, so
            // their wave-wide maxima are taken once after the loop (res_* hold this lane's values until then)
            res_stat = ls; res_ineq = li; res_comp = lcmp;
            gap = wave_sum(lg) * inv_npairs;
            const bool lane_nan = !(ls == ls) || !(li == li) || !(lcmp == lcmp);
            if (__any(lane_nan) || !(gap == gap)) { qp_status = 3; break; }
            const bool lane_open = (ls > p_ts * qn) || (li > p_ti) || (lcmp > p_tc);
            if (!__any(lane_open)) { qp_status = 0; break; }
            if (it >= p_itmax) { qp_status = 1; break; }
            TUM_TICK(2);
            publish(gsum[0], gsum[1], sGamH);
        }
        // ---- M = H + C' Gamma C, one 16x16 tile at a time (single accumulator live)
        {
            // MFMA operands of the gg rows, fetched once: chunk c (rows 4c+1..4c+4) x tile column T, needed for c >= 2T
            // (row s has 2s entries). chv = the row entries (B operand), the A operand is chv * gamma(row).
            double gch[10];
#pragma unroll
            for (int c = 0; c < 10; c++) gch[c] = sGamH[4 * c + lq];        // rows beyond the horizon were published as 0
            double chv[10][NT];
#pragma unroll
            for (int T = 0; T < NT; T++)
#pragma unroll
                for (int c = 2 * T; c < 10; c++) {
                    const int s = 4 * c + lq + 1;                     // <= 40 always; rows beyond the horizon are zero
                    const double v = sCh[hoff(s) + 16 * T + lc];     // in range for every lane, masked below
                    // the row-length mask only bites for the first two chunks of a tile column (16T + 15 < 2(4c + 1) beyond)
                    chv[c][T] = (c >= 2 * T + 2) ? v : ((16 * T + lc < 2 * s) ? v : 0.0);
                }
            const double dt2 = dt * dt;
            // steering-angle rows (structured): element (row, col), both odd, gets dt^2 * suffix(max(row, col)); in a tile
            // right of the diagonal that is a function of the column only
            double sfxo[NT];
#pragma unroll
            for (int I = 0; I < NT; I++) {
                const int col = 16 * I + lc;
                const double sf = sSfx[(col >> 1) + 1];
                sfxo[I] = ((lq & 1) && (lc & 1) && col < nv) ? dt2 * sf : 0.0;
            }
#pragma unroll
            for (int K = 0; K < NT; K++)
#pragma unroll
                for (int I = K; I < NT; I++) {
                    d4 acc = Ht[tidx(K, I)];
                    if (K == I) {
                        // diagonal tile: steering rows, box rows and regularisation
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) {
                            const int row = 16 * K + lq + 4 * jj, col = 16 * I + lc;
                            const int mx = (row > col) ? row : col;
                            const double sf = sSfx[(mx >> 1) + 1];
                            double add = ((row & 1) && (col & 1) && mx < nv) ? dt2 * sf : 0.0;
                            const double wb = sWb[(row >> 1) < NMAX ? (row >> 1) : 0];
                            if (row == col) add += p_reg + (((row & 1) && row < nv) ? wb : 0.0);
                            acc[jj] += add;
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) acc[jj] += sfxo[I];
                    }
                    // gg rows: SYRK over chunks of 4 rows (stages 4c+1 .. 4c+4); tile column I needs c >= 2I
#pragma unroll
                    for (int c = 2 * I; c < 10; c++) acc = mfma(chv[c][K] * gch[c], chv[c][I], acc);
                    // store as packed lower triangle: M[col_g][row_g] for col_g >= row_g
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int rg = 16 * K + lq + 4 * jj, cg = 16 * I + lc;
                        if (K < I) sM[rb[I] + rg] = acc[jj];
                        else sM[(cg >= rg) ? rb[I] + rg : (O_DUMMY - O_M)] = acc[jj];
                    }
                }
        }
        wsync();
        if (PROF && (ka.flags & 2) && b < 4 && it == 0) {
            double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
            for (int i = lane; i < LPK; i += 64) dbg[13300 + i] = sM[i];
        }

        TUM_TICK(3);
        // ---- blocked L D L' factorisation. Unit-lower L overwrites the strict lower triangle of M, the
        //      pivots go on its diagonal. Block column J (16 wide): (1) left-looking update with the block
        //      columns already factorised (MFMA, 4 per earlier block), result kept in registers as D-layout
        //      tiles; (2) four 4-column micro-panels: the 4x4 diagonal block is factorised redundantly by
        //      every lane from LDS broadcasts (all scalars stay in VGPRs: no readlane / SGPR traffic), every
        //      row below solves its 4 entries against it, and the rest of the panel gets a rank-4 MFMA update.
        double dmin = 1.0;                               // smallest pivot seen (the matrix must be positive definite)
#pragma unroll
        for (int J = 0; J < NT; J++) {
            // Register tiles of the ROW panel A_JI (rows of block J, columns of block I; by symmetry the same numbers as
            // the column panel): element jj of lane (lq, lc) is A[16J + lq + 4jj][16I + lc]. In this orientation the four
            // columns of micro-panel m are element m of EVERY lane, so publishing them is one unmasked store per tile with
            // a trivial address (rb[I] + column), and the diagonal tile is simply kept symmetric.
            d4 T[NT];
#pragma unroll
            for (int I = J; I < NT; I++) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int r = lq + 4 * jj;
                    if (I > J) T[I][jj] = sM[rb[I] + 16 * J + r];
                    else { const int hi = (r > lc) ? r : lc, lo = (r > lc) ? lc : r; T[I][jj] = sM[lpk(16 * J + hi, 16 * J + lo)]; }
                }
            }
#pragma unroll
            for (int K = 0; K < J; K++)
#pragma unroll
                for (int kc = 0; kc < 4; kc++) {
                    const int kk = 16 * K + 4 * kc + lq;
                    const double aJ = -sM[rb[J] + kk] * sM[lpk(kk, kk)];
#pragma unroll
                    for (int I = J; I < NT; I++) T[I] = mfma(aJ, sM[rb[I] + kk], T[I]);
                }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int c0 = 16 * J + 4 * m;
                // (a) publish columns c0..c0+3 of the panel tiles (rows >= column) for the lane = row readers
#pragma unroll
                for (int I = J; I < NT; I++)
                    sM[(I > J || lc >= 4 * m + lq) ? rb[I] + c0 + lq : (O_DUMMY - O_M)] = T[I][m];
                wsync();
                // (b) 4x4 diagonal block, L D L' on uniform values
                const int r1 = lpk(c0 + 1, c0), r2 = lpk(c0 + 2, c0), r3 = lpk(c0 + 3, c0);
                const double a00 = sM[lpk(c0, c0)];
                const double a10 = sM[r1], a11 = sM[r1 + 1];
                const double a20 = sM[r2], a21 = sM[r2 + 1], a22 = sM[r2 + 2];
                const double a30 = sM[r3], a31 = sM[r3 + 1], a32 = sM[r3 + 2], a33 = sM[r3 + 3];
                // (c) this lane's rows below the block (issued together with the broadcasts)
                const int row = c0 + 4 + lane;
                const bool rin = row < NVP;
                const int rbase = lpk(rin ? row : NVP - 1, c0);
                double e0 = sM[rbase], e1 = sM[rbase + 1], e2 = sM[rbase + 2], e3 = sM[rbase + 3];
                const bool two = (J == 0) && (m < 3);             // rows c0+68.. exist only for c0 < 12
                const int row1 = c0 + 68 + lane;
                const bool rin1 = two && (row1 < NVP);
                const int rbase1 = lpk(rin1 ? row1 : NVP - 1, c0);
                double f0 = 0, f1 = 0, f2 = 0, f3 = 0;
                if (two) { f0 = sM[rbase1]; f1 = sM[rbase1 + 1]; f2 = sM[rbase1 + 2]; f3 = sM[rbase1 + 3]; }
                const double d0 = a00, i0 = frcp(d0);
                const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
                const double d1 = a11 - l10 * a10, i1 = frcp(d1);
                const double y21 = a21 - l20 * a10, y31 = a31 - l30 * a10;
                const double l21 = y21 * i1, l31 = y31 * i1;
                const double d2 = a22 - l20 * a20 - l21 * y21, i2 = frcp(d2);
                const double y32 = a32 - l30 * a20 - l31 * y21;
                const double l32 = y32 * i2;
                const double d3 = a33 - l30 * a30 - l31 * y31 - l32 * y32, i3 = frcp(d3);
                dmin = fmin(dmin, fmin(fmin(d0, d1), fmin(d2, d3)));      // (fmin drops NaNs: those surface in the residual test)
                // forward substitution of every row below against L4 (y = L * d), then L = y / d
                e1 -= l10 * e0; e2 -= l20 * e0 + l21 * e1; e3 -= l30 * e0 + l31 * e1 + l32 * e2;
                if (rin) { sM[rbase] = e0 * i0; sM[rbase + 1] = e1 * i1; sM[rbase + 2] = e2 * i2; sM[rbase + 3] = e3 * i3; }
                if (two) {
                    f1 -= l10 * f0; f2 -= l20 * f0 + l21 * f1; f3 -= l30 * f0 + l31 * f1 + l32 * f2;
                    if (rin1) { sM[rbase1] = f0 * i0; sM[rbase1 + 1] = f1 * i1; sM[rbase1 + 2] = f2 * i2; sM[rbase1 + 3] = f3 * i3; }
                }
                if (lane == 0) {
                    sM[r1] = l10; sM[r2] = l20; sM[r2 + 1] = l21; sM[r3] = l30; sM[r3 + 1] = l31; sM[r3 + 2] = l32;
                    sM[lpk(c0, c0)] = d0; sM[lpk(c0 + 1, c0 + 1)] = d1; sM[lpk(c0 + 2, c0 + 2)] = d2; sM[lpk(c0 + 3, c0 + 3)] = d3;
                }
                wsync();
                // (d) rank-4 update of the panel's remaining columns on the matrix cores
                if (m < 3) {
                    const double dsel = (lq == 0) ? d0 : (lq == 1) ? d1 : (lq == 2) ? d2 : d3;
                    const int kcol = c0 + lq;
                    const int rowJ = 16 * J + lc;
                    const double bl = sM[lpk(rowJ, kcol)];        // (finite whatever it is; selected below)
                    const double bval = (rowJ > kcol) ? bl : ((rowJ == kcol) ? 1.0 : 0.0);
#pragma unroll
                    for (int I = J; I < NT; I++) {
                        double aval;
                        if (I == J) aval = bval;
                        else aval = sM[rb[I] + kcol];                 // rows of later blocks are always below
                        T[I] = mfma(-bval * dsel, aval, T[I]);
                    }
                }
            }
        }
        if (!(dmin > 1e-300)) { qp_status = 3; break; }
        if (PROF && (ka.flags & 2) && b < 4 && it == 0) {
            double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
            for (int i = lane; i < LPK; i += 64) dbg[16540 + i] = sM[i];
        }

        // ---- inverses of the unit-lower 16x16 diagonal blocks, in place (strict lower triangle of every diagonal block of
        //      M now holds inv(L_JJ)): the substitutions below then treat 16 unknowns at a time. Lane = column c of its block:
        //      X[r] = delta(r,c) - sum_{k<r} L[r][k] X[k]; entries above the diagonal stay 0, so no lane masks are needed.
        {
            auto inv_diag = [&](const int g, const bool own) {
                const int gb = g & ~15, cl = g & 15;
                double X[16];
#pragma unroll
                for (int k = 0; k < 16; k++) X[k] = (k == cl) ? 1.0 : 0.0;
                int rowb = lpk(gb, gb);
                int rows[16];
#pragma unroll
                for (int r = 1; r < 16; r++) {
                    rowb += gb + r;                               // lpk(gb + r, gb)
                    rows[r] = rowb;
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll
                    for (int k = 0; k < r; k++) {
                        if (k & 1) a1 += sM[rowb + k] * X[k]; else a0 += sM[rowb + k] * X[k];
                    }
                    X[r] -= a0 + a1;
                }
                wsync();                                          // every lane has read its L entries
#pragma unroll
                for (int r = 1; r < 16; r++) sM[(own && r > cl) ? rows[r] + cl : (O_DUMMY - O_M)] = X[r];
                wsync();
            };
            inv_diag(lane, true);
            inv_diag(lane1, lane < 16);
        }

        TUM_TICK(4);
        // ---- predictor / corrector
        double cross1[4], cross2[4];                 // dT*dL and dS*dMu of the affine step
        double dv0 = 0.0, dv1 = 0.0, alpha = 1.0, sigma = 0.0;

#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            // the lane-derived predicates / addresses are derived afresh for the solve passes: kept from the top of the
            // iteration they would sit in (spilled) SGPRs across the whole factorisation
            int lane_p = lane_outer;
            asm volatile("" : "+v"(lane_p));
            const int lane = lane_p;
            const int lq = lane >> 4, lc = lane & 15;
            TUM_LANE_DEFS
            const double tau = (pass == 1) ? fmax(sigma * gap, 0.1 * p_tc) : 0.0;
            // row phase B1: rhs weights  w = gam_l*rho_l - gam_u*rho_u
            {
                double w[2];
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    w[rr] = 0.0;
#pragma unroll
                    for (int sd = 0; sd < 2; sd++) {
                        const int k = rr * 2 + sd;
                        const double s_ = ROWF(0, k), t_ = ROWF(1, k), l_ = ROWF(2, k), m_ = ROWF(3, k);
                        // gam * rho with gam = l G, G = 1/(t + l s D), D = 1/(Z s + mu):
                        //   gam * rho = G * (rc1 - l * (rt + (rs s + rc2) D))       (two reciprocals instead of four)
                        const double D = rD[k], G = rG[k];
                        double rc1 = t_ * l_, rc2 = s_ * m_;
                        if (pass == 1) { rc1 += cross1[k] - tau; rc2 += cross2[k] - tau; }
                        const double gr = G * (rc1 - l_ * (ROWF(5, k) + (ROWF(4, k) * s_ + rc2) * D));
                        w[rr] += sd ? -gr : gr;
                    }
                }
                publish(w[0], w[1], sWh);
            }
            double b0, b1;
            ctw(b0, b1);
            b0 = v0on ? -rv0 - b0 : 0.0; b1 = v1on ? -rv1 - b1 : 0.0;
            if (PROF && (ka.flags & 2) && b < 4 && it == 0 && pass == 0) {
                double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
                dbg[19780 + lane] = b0;
                if (lane < 16) dbg[19780 + 64 + lane] = b1;
            }
            TUM_TICK(5);
            // Triangular solves on 16x16 tiles. Lane (q, c) = (lane >> 4, lane & 15); a vector block is kept REPLICATED over the
            // four DPP rows (lane (q, c) holds v[c]); a tile X is read from the packed factor as X[c][q + 4 jj], jj = 0..3 (row c,
            // four of its columns), so every lane adds 4 products per tile, the four partial sums of a row meet in a quad_sum
            // (row swaps, VALU) and the only other cross-lane step is handing the 16 new unknowns to the lanes that need them
            // as column values (v[q + 4 jj]: four lane gathers per block). No LDS broadcasts, no DPP chains: per block row two
            // quad_sums and two rounds of gathers on the critical path, 60 tile reads + 60 FMAs per solve.
            {
                int ga[4];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) ga[jj] = ((lane & 48) | ((lq + 4 * jj) & 15)) << 2;
                const bool ondiag = (lc >= lq) && (((lc - lq) & 3) == 0);      // this lane holds the unit diagonal entry of its row
                double bj[NT], vs[NT][4];
#pragma unroll
                for (int J = 0; J < 4; J++) bj[J] = lane_gather(b0, (16 * J + lc) << 2);
                bj[4] = lane_gather(b1, lc << 2);
                // ---- forward: L y = b
#pragma unroll
                for (int J = 0; J < NT; J++) {
                    double t = bj[J];
                    if (J > 0) {
                        double acc = 0.0, acc1 = 0.0;
#pragma unroll
                        for (int K = 0; K < J; K++)
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const double lv = sM[rb[J] + 16 * K + lq + 4 * jj];
                                if (jj & 1) acc1 += lv * vs[K][jj]; else acc += lv * vs[K][jj];
                            }
                        t -= quad_sum(acc + acc1);
                    }
                    double a2 = ondiag ? t : 0.0;                  // inv(L_JJ) has a unit diagonal
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const double lv = sM[rb[J] + 16 * J + lq + 4 * jj];          // entry (c, q + 4 jj) of the block: inside M for any jj
                        const double tv = lane_gather(t, ga[jj]);
                        a2 += ((lq + 4 * jj < lc) ? lv : 0.0) * tv;
                    }
                    const double y = quad_sum(a2);
                    bj[J] = y * frcp(sM[rb[J] + 16 * J + lc]);       // z = D^-1 y rides along (pivot d on the diagonal of M)
                    if (J < NT - 1) {
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) vs[J][jj] = lane_gather(y, ga[jj]);
                    }
                }
                // ---- backward: L' x = z
#pragma unroll
                for (int J = NT - 1; J >= 0; J--) {
                    double t = bj[J];
                    if (J < NT - 1) {
                        double acc = 0.0, acc1 = 0.0;
#pragma unroll
                        for (int I = J + 1; I < NT; I++)
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const double lv = sM[lpk(16 * I + lq + 4 * jj, 0) + 16 * J + lc];   // L[16I + q + 4jj][16J + c]
                                if (jj & 1) acc1 += lv * vs[I][jj]; else acc += lv * vs[I][jj];
                            }
                        t -= quad_sum(acc + acc1);
                    }
                    double a2 = ondiag ? t : 0.0;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const double lv = sM[lpk(16 * J + lq + 4 * jj, 0) + 16 * J + lc];           // inv(L_JJ)[q + 4jj][c]
                        const double tv = lane_gather(t, ga[jj]);
                        a2 += ((lq + 4 * jj > lc) ? lv : 0.0) * tv;
                    }
                    const double x = quad_sum(a2);
                    bj[J] = x;
                    if (J > 0) {
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) vs[J][jj] = lane_gather(x, ga[jj]);
                    }
                }
                b0 = (lq == 0) ? bj[0] : (lq == 1) ? bj[1] : (lq == 2) ? bj[2] : bj[3];
                b1 = (lane < 16) ? bj[4] : 0.0;
            }
            dv0 = b0; dv1 = b1;
            TUM_TICK(6);
            wsync();
            sDv[lane] = dv0;
            if (lane < 16) sDv[64 + lane] = dv1;
            wsync();
            if (PROF && (ka.flags & 2) && b < 4 && it == 0 && pass == 0) {
                double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
                dbg[19860 + lane] = dv0;
                if (lane < 16) dbg[19860 + 64 + lane] = dv1;
            }
            // row phase B2: C*dv for this lane's rows, step in (s,t,lam,mu), step length
            double cdv[2];
            {
                const double xo = boxlane ? sDv[2 * lane + 1] : 0.0;
                const double pfx = dt * wave_prefix(xo, lane);
                // gg rows: row s has 2s entries (up to 80). Rows 17..40 are split in two halves of s entries, the second
                // half goes to lanes 40..63, so no lane walks more than 40 entries; the row sums then travel through LDS
                // (the weight buffer is free between the right-hand side and the next publish) to the lanes that own
                // the gg rows.
                const int hs = (lane < NMAX) ? lane + 1 : lane - 23;              // row handled (lanes 40..63: rows 17..40)
                const int cstart = (lane < NMAX) ? 0 : hs;
                const int ncol = (hs <= N) ? ((hs <= 16) ? 2 * hs : hs) : 0;       // 0: no such row at this horizon
                const double *ch = sCh + hoff(hs) + cstart;
                const double *dvp = sDv + cstart;
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int i = 0; i < NMAX; i += 2) {
                    const double x0_ = ch[i], x1_ = ch[i + 1];
                    a0 += ((i < ncol) ? x0_ : 0.0) * dvp[i];
                    a1 += ((i + 1 < ncol) ? x1_ : 0.0) * dvp[i + 1];
                }
                const double part = a0 + a1;
                const double other = __shfl(part, (lane + 24) & 63, 64);           // second half of rows 17..40
                wsync();
                if (lane < NMAX) sWh[lane] = (lane >= 16) ? part + other : part;    // C dv of gg row lane+1
                wsync();
                cdv[0] = gglane ? sWh[2 * gj] : xo;
                cdv[1] = gglane ? sWh[2 * gj + 1] : pfx;
            }
            double amax = 1.0, lmu = 0.0;
            double dcur[4][4];
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    const int k = rr * 2 + sd;
                    const bool on = rr ? on1 : on0;
                    const double eps = sd ? -1.0 : 1.0;
                    const double s_ = ROWF(0, k), t_ = ROWF(1, k), l_ = ROWF(2, k), m_ = ROWF(3, k);
                    const double is_ = frcp(s_), il_ = frcp(l_), it_ = frcp(t_), im_ = frcp(m_);
                    const double iDs = s_ * rD[k];
                    const double gam = l_ * rG[k];
                    double rc1 = t_ * l_, rc2 = s_ * m_;
                    if (pass == 1) { rc1 += cross1[k] - tau; rc2 += cross2[k] - tau; }
                    const double rsk = ROWF(4, k);
                    const double rho = -ROWF(5, k) + rc1 * il_ - (rsk + rc2 * is_) * iDs;
                    const double dl = -gam * (eps * cdv[rr] + rho);
                    const double dsl = (dl - rsk - rc2 * is_) * iDs;
                    const double dm = (-rc2 - m_ * dsl) * is_;
                    const double dtt = (-rc1 - t_ * dl) * il_;
                    dcur[0][k] = on ? dsl : 0.0; dcur[1][k] = on ? dtt : 0.0; dcur[2][k] = on ? dl : 0.0; dcur[3][k] = on ? dm : 0.0;
                    // largest alpha keeping x + alpha*dx >= 0:  alpha <= 1 / max(-dx/x)
                    double q = fmax(fmax(-dsl * is_, -dtt * it_), fmax(-dl * il_, -dm * im_));
                    q = on ? q : 0.0;
                    amax = fmax(amax, q);                 // amax temporarily holds max(1, max ratio)
                    if (pass == 0) { cross1[k] = on ? dtt * dl : 0.0; cross2[k] = on ? dsl * dm : 0.0; }
                }
            amax = frcp(wave_max(amax));                  // = min(1, 1/max ratio)
            if (pass == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double pr = (ROWF(1, k) + amax * dcur[1][k]) * (ROWF(2, k) + amax * dcur[2][k])
                                    + (ROWF(0, k) + amax * dcur[0][k]) * (ROWF(3, k) + amax * dcur[3][k]);
                    lmu += ((k < 2) ? on0 : on1) ? pr : 0.0;
                }
                const double mu_aff = wave_sum(lmu) * inv_npairs;
                const double ratio = mu_aff * frcp(gap);
                sigma = ratio * ratio * ratio;
                // safeguard (same rule as the oracle): an affine step blocked almost immediately makes the second-order
                // terms a wild extrapolation (a soft-constraint pair at the kink of its L1 penalty can then cycle for all
                // 50 iterations); drop them for this iteration
                if (amax < 0.1) {
#pragma unroll
                    for (int k = 0; k < 4; k++) { cross1[k] = 0.0; cross2[k] = 0.0; }
                }
            } else {
                alpha = (amax >= 1.0) ? 1.0 : 0.995 * amax;
                if (alpha >= 1e-12) {
                    const double om_ = 1.0 - alpha;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
#pragma unroll
                        for (int f = 0; f < 4; f++) ROWF(f, k) += alpha * dcur[f][k];
                        ROWF(4, k) *= om_; ROWF(5, k) *= om_;
                    }
                }
            }
            TUM_TICK(7);
        }
        if (alpha < 1e-12) { qp_status = 2; break; }
        v0 += alpha * dv0; v1 += alpha * dv1;
        const double om = 1.0 - alpha;
        rv0 *= om; rv1 *= om;
        wsync();
    }
    const int status = acados_status(qp_status);
    res_stat = wave_max(res_stat); res_ineq = wave_max(res_ineq); res_comp = wave_max(res_comp);
    TUM_LANE_DEFS

    TUM_TICK(8);
    // ------------------------------------------------------------ phase 4: expand, full step, cost
    // slack part of the cost and the slack outputs first: the row state is about to be overwritten by X, U
    double cl = 0.0;
    {
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                const double sv = ROWF(0, rr * 2 + sd);
                const double c = pen(rr, sd, 0) * sv + 0.5 * pen(rr, sd, 1) * sv * sv;
                cl += (rr ? on1 : on0) ? c : 0.0;
            }
        if (ka.slack) {
            double *sl = ka.slack + (size_t)b * 6 * N;
            // order: [lower: box_k (N) | (bx_k, h_k) k=1..N] then the same for upper
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                if (boxlane) {
                    sl[sd * 3 * N + lane] = ROWF(0, 0 + sd);
                    sl[sd * 3 * N + N + 2 * lane] = ROWF(0, 2 + sd);
                }
                if (gglane && on0) sl[sd * 3 * N + N + 2 * (stg0 - 1) + 1] = ROWF(0, 0 + sd);
                if (gglane && on1) sl[sd * 3 * N + N + 2 * (stg1 - 1) + 1] = ROWF(0, 2 + sd);
            }
        }
    }
    wsync();
    sDv[lane] = v0;
    if (lane < 16) sDv[64 + lane] = v1;
    if (SN) {   // the sample copies take their step in the epilogue kernel
        ka.dv[(size_t)b * NVP + lane] = v0;
        if (lane < 16) ka.dv[(size_t)b * NVP + 64 + lane] = v1;
    }
    // bring back the iterate and the linearisation records
    // (compile-time trip counts: every global load of these copies is in flight before the first one is consumed)
    double gx0r, Wc[10], yrc[6];
    {
        constexpr int NXI = ((NMAX + 1) * NX + 63) / 64, NWI = (NMAX * ABS + 63) / 64;
        const double *ws = ka.ws + (size_t)b * WS_DOUBLES;
        double tx[NXI], tw[NWI];
        gx0r = gx0[(lane < 8) ? lane : 0];                   // (used by the expansion and the cost below)
#pragma unroll
        for (int i = 0; i < 10; i++) Wc[i] = (i < 6) ? gW[i] : gW[N * 6 + i - 6];
#pragma unroll
        for (int i = 0; i < 6; i++) yrc[i] = gyref[((lane <= N) ? lane : 0) * 6 + i];
#pragma unroll
        for (int j = 0; j < NXI; j++) { const int i = lane + 64 * j; tx[j] = (i < (N + 1) * NX) ? gX[i] : 0.0; }
#pragma unroll
        for (int j = 0; j < NWI; j++) { const int i = lane + 64 * j; tw[j] = (i < N * ABS) ? ws[i] : 0.0; }
#pragma unroll
        for (int j = 0; j < NXI; j++) { const int i = lane + 64 * j; if (i < (N + 1) * NX) sX[i] = tx[j]; }
#pragma unroll
        for (int j = 0; j < NWI; j++) { const int i = lane + 64 * j; if (i < N * ABS) sAB[i] = tw[j]; }
    }
    for (int i = lane; i < NVP; i += 64) sU1[i] = (i < nv) ? gU[i] : 0.0;
    wsync();
    if (status == 0) {
        // dx_0 = x0 - X_0 ; dx_{k+1} = A_k dx_k + B_k du_k + b_k, lane i < 8 carries row i of dx:
        //   dx'_i = diag_i dx_i + cpsi_i dx_2 + sum_{c<5} S_i[c] dx_{3+c} + S_i[5] du0 + S_i[6] du1 + b_i
        // (rows 6,7: pure integrators of the inputs). The six coupled entries are broadcast with readlane.
        const int ri = (lane < 8) ? lane : 0;                 // lanes >= 8 shadow row 0 and never store
        const bool core = ri < 6;
        const double diag = (ri < 3 || ri >= 6) ? 1.0 : 0.0;
        double dxi = gx0r - sX[ri];
        wsync();
        if (lane < 8) sX[lane] += dxi;
        if (SN) {
            // stages 1..uph: dx_s = G_nom,s dU + g_nom,s straight from the prologue's matrices (2s columns each)
            for (int s = 1; s <= uph; s++) {
                const double *pg = gpro + (size_t)(s - 1) * SN_PRO_STAGE + ri * 64;
                double acc = pg[2 * uph];
                for (int j = 0; j < 2 * s; j++) acc += pg[j] * sDv[j];
                dxi = acc;
                if (lane < 8) sX[s * NX + lane] += dxi;
            }
        }
        for (int k = uph; k < N; k++) {
            const double *rec = sAB + k * ABS;
            const double *Si = rec + 2 + (core ? ri : 0) * 7;
            const double du0 = sDv[2 * k], du1 = sDv[2 * k + 1];
            const double cpsi = (ri < 2) ? rec[ri] : 0.0;
            double c0 = Si[0], c1 = Si[1], c2 = Si[2], c3 = Si[3], c4 = Si[4], c5 = Si[5], c6 = Si[6];
            if (!core) { c0 = c1 = c2 = c3 = c4 = 0.0; c5 = (ri == 7) ? dt : 0.0; c6 = (ri == 6) ? dt : 0.0; }
            const double bi = rec[44 + ri];
            const double x2 = rl(dxi, 2), x3 = rl(dxi, 3), x4 = rl(dxi, 4), x5 = rl(dxi, 5), x6 = rl(dxi, 6), x7 = rl(dxi, 7);
            double acc0 = diag * dxi + cpsi * x2 + bi;
            double acc1 = c5 * du0 + c6 * du1;
            acc0 += c0 * x3; acc1 += c1 * x4;
            acc0 += c2 * x5; acc1 += c3 * x6;
            acc0 += c4 * x7;
            dxi = acc0 + acc1;
            if (lane < 8) sX[(k + 1) * NX + lane] += dxi;
        }
        sU1[lane] += v0;
        if (lane < 16) sU1[64 + lane] += v1;
    }
    wsync();
    // cost at the (new) iterate: stage terms scaled by dt, terminal unscaled, slack penalties pre-scaled
    // (weights and references are re-read here rather than kept in registers across the whole IPM)
    if (lane <= N) {
        double Wd[6], We[4], yr[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { Wd[i] = Wc[i]; yr[i] = yrc[i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) We[i] = Wc[6 + i];
        const int k = lane;
        const double sc = (k < N) ? dt : 1.0;
        double acc = 0.0, e;
        e = sX[k * NX + 0] - yr[0]; acc += ((k < N) ? Wd[0] : We[0]) * e * e;
        e = sX[k * NX + 1] - yr[1]; acc += ((k < N) ? Wd[1] : We[1]) * e * e;
        e = wrap_yaw(sX[k * NX + 2]) - yr[2]; acc += ((k < N) ? Wd[2] : We[2]) * e * e;
        e = (SN ? sqrt(sX[k * NX + 3] * sX[k * NX + 3] + sX[k * NX + 4] * sX[k * NX + 4]) : sX[k * NX + 3]) - yr[3];
        acc += ((k < N) ? Wd[3] : We[3]) * e * e;
        if (k < N) {
            e = sU1[2 * k] - yr[4]; acc += Wd[4] * e * e;
            e = sU1[2 * k + 1] - yr[5]; acc += Wd[5] * e * e;
        }
        cl += 0.5 * sc * acc;
    }
    const double cost = wave_sum(cl);

    // ------------------------------------------------------------ stores
    for (int i = lane; i < (N + 1) * NX; i += 64) gX[i] = sX[i];
    for (int i = lane; i < nv; i += 64) gU[i] = sU1[i];
    TUM_TICK(9);
    if (PROF && lane == 0)
        for (int i = 0; i < 12; i++) ka.prof[(size_t)b * 12 + i] = pacc[i];
    if (lane == 0) {
        ka.cost[b] = cost;
        ka.status[b] = status;
        ka.qp_iter[b] = it;
        ka.qp_status[b] = qp_status;
        ka.res[b * 3 + 0] = res_stat; ka.res[b * 3 + 1] = res_ineq; ka.res[b * 3 + 2] = res_comp;
    }
}

}  // namespace tum
