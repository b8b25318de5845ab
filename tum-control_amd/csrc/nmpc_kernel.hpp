// nmpc_kernel.hpp -- the fused SQP-RTI kernel (one wavefront = one OCP instance).
#pragma once
#include "nmpc_device.hpp"

namespace tum {

// Row-side state of the interior point method held by lane l < N:
//   row 0: steering-rate box of stage l      (variable 2l+1)
//   row 1: steering-angle bound of stage l+1 (general row 2l)
//   row 2: gg-circle constraint of stage l+1 (general row 2l+1)
// side 0 = lower, 1 = upper.
struct RowState {
    double s[3][2], t[3][2], lam[3][2], mu[3][2];
    double z[3][2], Z[3][2];
    double rs[3][2], rt[3][2];
};

// optional in-kernel phase timers (flags & 4): cycles per phase accumulated into ka.prof[b][16]
#define TUM_TICK(slot) do { if (ka.flags & 4) { const long long t_ = __builtin_readcyclecounter(); pacc[slot] += t_ - tprev; tprev = t_; } } while (0)

__device__ __forceinline__ int acados_status(int qp_status) { return (qp_status == 0 || qp_status == 1) ? 0 : 4; }

__global__ void __launch_bounds__(64, 1) nmpc_rti_kernel(const KArgs ka)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= ka.batch) return;
    const int N = ka.N, nv = 2 * N;
    const double dt = ka.dt;
    const Model &mp = ka.mp;

    double *sAB = lds + O_AB, *sM = lds + O_M, *sStage = lds + O_STAGE, *sC = lds + O_C, *sX = lds + O_X;
    double *sG = lds + O_G, *sRes = lds + O_RES, *sGh = lds + O_GH, *sD = lds + O_D, *sGam = lds + O_GAM;
    double *sWr = lds + O_WR, *sWb = lds + O_WB, *sDv = lds + O_DV, *sInvD = lds + O_INVD, *sU = lds + O_U;

    double *gX = ka.X + (size_t)b * (N + 1) * NX;
    double *gU = ka.U + (size_t)b * N * NU;
    const double *gx0 = ka.x0 + (size_t)b * NX;
    const double *gyref = ka.yref + (size_t)b * (N + 1) * 6;
    const double *gW = ka.W + (size_t)b * 10;
    const double *gpen = ka.pen + (size_t)b * 36;
    const double *gbnd = ka.bnd + (size_t)b * 6 * (N + 1);

    long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
    // ------------------------------------------------------------ phase 0: loads
    for (int i = lane; i < (N + 1) * NX; i += 64) sX[i] = gX[i];
    for (int i = lane; i < NVP; i += 64) sU[i] = (i < nv) ? gU[i] : 0.0;
    double Wd[6], We[4];
#pragma unroll
    for (int i = 0; i < 6; i++) Wd[i] = gW[i];
#pragma unroll
    for (int i = 0; i < 4; i++) We[i] = gW[6 + i];
    wsync();

    // ------------------------------------------------------------ phase 1: linearise (lane = stage)
    double yr[6];
    {
        const int k = lane;
        if (k <= N) {
#pragma unroll
            for (int i = 0; i < 6; i++) yr[i] = gyref[k * 6 + i];
            double xk[8];
#pragma unroll
            for (int i = 0; i < 8; i++) xk[i] = sX[k * NX + i];
            // cost residual of the 4 state rows: y = [x0, x1, wrap(x2), x3]
            sRes[k * 4 + 0] = xk[0] - yr[0];
            sRes[k * 4 + 1] = xk[1] - yr[1];
            sRes[k * 4 + 2] = wrap_yaw(xk[2]) - yr[2];
            sRes[k * 4 + 3] = xk[3] - yr[3];
            if (k >= 1) {
                double h, g3, g5, g7;
                h_con(mp, xk[3], xk[5], xk[7], h, g3, g5, g7);
                sGh[k * 4 + 0] = g3; sGh[k * 4 + 1] = g5; sGh[k * 4 + 2] = g7; sGh[k * 4 + 3] = h;
            }
            if (k < N) {
                double uk[2] = {sU[2 * k], sU[2 * k + 1]};
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
    double w0[8], w1[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { w0[i] = 0.0; w1[i] = 0.0; }
    const int j0 = lane >> 1, r0 = lane & 1, j1 = 32 + (lane >> 1);
    const bool isg = (lane == 16);
    if (isg) {
#pragma unroll
        for (int i = 0; i < 8; i++) { w1[i] = gx0[i] - sX[i]; sG[i] = w1[i]; }
    }
    double q0 = 0.0, q1 = 0.0;
    d4 Ht[NTT];
#pragma unroll
    for (int i = 0; i < NTT; i++) Ht[i] = d4{0.0, 0.0, 0.0, 0.0};
    const int lq = lane >> 4, lc = lane & 15;

    for (int k = 0; k < N; k++) {
        const double *rec = sAB + k * ABS;
        // bank 0
        if (j0 < k) apply_A(rec, w0);
        else if (j0 == k) {
#pragma unroll
            for (int i = 0; i < 6; i++) w0[i] = rec[2 + i * 7 + 5 + r0];
            w0[6] = r0 ? dt : 0.0; w0[7] = r0 ? 0.0 : dt;
        }
        // bank 1
        if (isg) {
            apply_A(rec, w1);
#pragma unroll
            for (int i = 0; i < 8; i++) w1[i] += rec[44 + i];
        } else if (lane < 16) {
            if (j1 < k) apply_A(rec, w1);
            else if (j1 == k) {
#pragma unroll
                for (int i = 0; i < 6; i++) w1[i] = rec[2 + i * 7 + 5 + r0];
                w1[6] = r0 ? dt : 0.0; w1[7] = r0 ? 0.0 : dt;
            }
        }
        const int s = k + 1;                         // stage whose G_s the lanes now hold
        const double sc = (s < N) ? dt : 1.0;
        const double g3 = sGh[s * 4 + 0], g5 = sGh[s * 4 + 1], g7 = sGh[s * 4 + 2];
        if (isg) {
#pragma unroll
            for (int i = 0; i < 8; i++) sG[s * NX + i] = w1[i];
            sD[2 * (s - 1)] = sX[s * NX + 6] + w1[6];
            sD[2 * (s - 1) + 1] = sGh[s * 4 + 3] + g3 * w1[3] + g5 * w1[5] + g7 * w1[7];
        }
        // constraint rows of stage s and staging of the 4 cost rows
        if (lane < 2 * s) {
            sC[coff(s, 0) + lane] = w0[6];
            sC[coff(s, 1) + lane] = g3 * w0[3] + g5 * w0[5] + g7 * w0[7];
        }
        if (lane < 16 && 64 + lane < 2 * s) {
            sC[coff(s, 0) + 64 + lane] = w1[6];
            sC[coff(s, 1) + 64 + lane] = g3 * w1[3] + g5 * w1[5] + g7 * w1[7];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            sStage[r * NVP + lane] = w0[r];
            if (lane < 16) sStage[r * NVP + 64 + lane] = w1[r];
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
                const double e = wr[r] * (sRes[s * 4 + r] + sG[s * NX + r]);
                a0 += e * w0[r]; a1 += e * w1[r];
            }
            q0 += a0;
            if (lane < 16) q1 += a1;
        }
        // Gauss-Newton Hessian SYRK (rank-4 update per stage) on the matrix cores
        const int Ts = (2 * s + 15) >> 4;
        const double wl = (lq == 0) ? wr[0] : (lq == 1) ? wr[1] : (lq == 2) ? wr[2] : wr[3];
        double aop[NT], bop[NT];
#pragma unroll
        for (int T = 0; T < NT; T++) {
            bop[T] = (T < Ts) ? sStage[lq * NVP + 16 * T + lc] : 0.0;
            aop[T] = bop[T] * wl;
        }
#pragma unroll
        for (int K = 0; K < NT; K++)
#pragma unroll
            for (int I = K; I < NT; I++)
                if (I < Ts) Ht[tidx(K, I)] = mfma(aop[K], bop[I], Ht[tidx(K, I)]);
        wsync();
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
        if (lane < nv) q0 += dt * Wd[4 + r0] * (sU[lane] - gyref[j0 * 6 + 4 + r0]);
        if (lane < 16 && 64 + lane < nv) q1 += dt * Wd[4 + r0] * (sU[64 + lane] - gyref[j1 * 6 + 4 + r0]);
    }

    if ((ka.flags & 2) && b < 4) {   // debug dump of the condensed QP
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
            for (int wch = 0; wch < 2; wch++) {
                const int row = 2 * (s - 1) + wch;
                for (int c = lane; c < NVP; c += 64) dbg[6480 + row * NVP + c] = (c < 2 * s) ? sC[coff(s, wch) + c] : 0.0;
            }
        for (int i = lane; i < 2 * N; i += 64) dbg[12880 + i] = sD[i];
        for (int i = lane; i < (N + 1) * NX; i += 64) dbg[12960 + i] = sG[i];
    }

    TUM_TICK(1);
    // ------------------------------------------------------------ phase 3: interior point
    RowState R;
    double dval[3], lo[3], hi[3];
    const bool rowlane = lane < N;
    {
        const int l = rowlane ? lane : 0;
        const int NB = N + 1;
        // constant terms and bounds
        dval[0] = sU[2 * l + 1]; lo[0] = gbnd[0 * NB + l]; hi[0] = gbnd[1 * NB + l];
        dval[1] = sD[2 * l];     lo[1] = gbnd[2 * NB + l + 1]; hi[1] = gbnd[3 * NB + l + 1];
        dval[2] = sD[2 * l + 1]; lo[2] = gbnd[4 * NB + l + 1]; hi[2] = gbnd[5 * NB + l + 1];
        // penalties: class 0 = stage 0, 1 = stages 1..N-1, 2 = stage N; slot = row; [zl, zu, Zl, Zu]
        const int cls0 = (l == 0) ? 0 : 1, cls12 = (l + 1 < N) ? 1 : 2;
        const double sc0 = dt, sc12 = (l + 1 < N) ? dt : 1.0;
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            const int cls = rr == 0 ? cls0 : cls12;
            const double sc = rr == 0 ? sc0 : sc12;
            const double *pp = gpen + (cls * 3 + rr) * 4;
            R.z[rr][0] = sc * pp[0]; R.z[rr][1] = sc * pp[1];
            R.Z[rr][0] = sc * pp[2]; R.Z[rr][1] = sc * pp[3];
        }
    }
    const double thr = sqrt(ka.mu0);
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
        for (int sd = 0; sd < 2; sd++) {
            const double eps = sd ? -1.0 : 1.0, bnd = sd ? hi[rr] : lo[rr];
            const double r0v = eps * (dval[rr] - bnd);
            R.s[rr][sd] = thr;
            double t = r0v + thr;
            if (t < thr) t = thr;
            R.t[rr][sd] = t;
            R.lam[rr][sd] = ka.mu0 / t;
            R.mu[rr][sd] = ka.mu0 / thr;
            R.rs[rr][sd] = R.z[rr][sd] + R.Z[rr][sd] * thr - R.lam[rr][sd] - R.mu[rr][sd];
            R.rt[rr][sd] = t - r0v - thr;
        }

    // v-space vectors in "vector layout": element c in lane c (bank 0), element 64+c in lane c<16 (bank 1)
    double v0 = 0.0, v1 = 0.0, rv0, rv1;

    // y = C' * w  with w_box in sWb[k], w_gen in sWr[row]; result in vector layout
    auto ctw = [&](double &o0, double &o1) {
        double a0 = 0.0, a1 = 0.0;
        if (lane & 1) a0 = ((lane >> 1) < N) ? sWb[lane >> 1] : 0.0;
        if ((lane & 1) && lane < 16) a1 = ((32 + (lane >> 1)) < N) ? sWb[32 + (lane >> 1)] : 0.0;
        for (int s = 1; s <= N; s++) {
            const double wd = sWr[2 * (s - 1)], wh = sWr[2 * (s - 1) + 1];
            if (lane < 2 * s) a0 += sC[coff(s, 0) + lane] * wd + sC[coff(s, 1) + lane] * wh;
            if (lane < 16 && 64 + lane < 2 * s) a1 += sC[coff(s, 0) + 64 + lane] * wd + sC[coff(s, 1) + 64 + lane] * wh;
        }
        o0 = a0; o1 = a1;
    };

    // initial stationarity residual r_v = q - C'(lam_l - lam_u)   (v = 0)
    if (rowlane) {
        sWb[lane] = R.lam[0][0] - R.lam[0][1];
        sWr[2 * lane] = R.lam[1][0] - R.lam[1][1];
        sWr[2 * lane + 1] = R.lam[2][0] - R.lam[2][1];
    }
    wsync();
    {
        double c0, c1;
        ctw(c0, c1);
        rv0 = q0 - c0; rv1 = q1 - c1;
        if (lane >= nv) rv0 = 0.0;
        if (!(lane < 16 && 64 + lane < nv)) rv1 = 0.0;
    }
    double qn = wave_max(fmax(fabs(q0), (lane < 16) ? fabs(q1) : 0.0));
    if (qn < 1.0) qn = 1.0;
    const double npairs = 12.0 * N;
    int it = 0, qp_status = 1;
    double res_stat = 0.0, res_ineq = 0.0, res_comp = 0.0;

    for (;; it++) {
        // ---- residual norms
        double ls = fmax(fabs(rv0), fabs(rv1)), li = 0.0, lcmp = 0.0, lg = 0.0;
        if (rowlane) {
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    ls = fmax(ls, fabs(R.rs[rr][sd]));
                    li = fmax(li, fabs(R.rt[rr][sd]));
                    const double c1 = R.t[rr][sd] * R.lam[rr][sd], c2 = R.s[rr][sd] * R.mu[rr][sd];
                    lcmp = fmax(lcmp, fmax(c1, c2));
                    lg += c1 + c2;
                }
        }
        res_stat = wave_max(ls); res_ineq = wave_max(li); res_comp = wave_max(lcmp);
        const double gap = wave_sum(lg) / npairs;
        if (!(res_stat == res_stat) || !(gap == gap)) { qp_status = 3; break; }
        if (res_stat <= ka.tol_stat * qn && res_ineq <= ka.tol_ineq && res_comp <= ka.tol_comp) { qp_status = 0; break; }
        if (it >= ka.iter_max) { qp_status = 1; break; }

        TUM_TICK(2);
        // ---- gamma, M = H + C' Gamma C
        double gam[3][2], Ds[3][2];
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                Ds[rr][sd] = R.Z[rr][sd] + R.mu[rr][sd] / R.s[rr][sd];
                gam[rr][sd] = 1.0 / (R.t[rr][sd] / R.lam[rr][sd] + 1.0 / Ds[rr][sd]);
            }
        if (rowlane) {
            sWb[lane] = gam[0][0] + gam[0][1];
            sGam[2 * lane] = gam[1][0] + gam[1][1];
            sGam[2 * lane + 1] = gam[2][0] + gam[2][1];
        } else if (lane < NMAX) {
            sWb[lane] = 0.0; sGam[2 * lane] = 0.0; sGam[2 * lane + 1] = 0.0;
        }
        wsync();
        {
            d4 Mt[NTT];
#pragma unroll
            for (int i = 0; i < NTT; i++) Mt[i] = Ht[i];
#pragma unroll
            for (int K = 0; K < NT; K++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int row = lq + 4 * jj;
                    if (row == lc) {
                        const int idx = 16 * K + row;
                        double add = ka.reg;
                        if ((idx & 1) && idx < nv) add += sWb[idx >> 1];
                        Mt[tidx(K, K)][jj] += add;
                    }
                }
            const int nchunk = (2 * N + 3) >> 2;
            for (int c = 0; c < nchunk; c++) {
                const int row = 4 * c + lq;                // general row handled by this lane group
                const int s = (row >> 1) + 1, wch = row & 1;
                const int smax = 2 * c + 2;                // last stage in the chunk
                const int Tc = (2 * smax + 15) >> 4;
                const double gr = (s <= N) ? sGam[row] : 0.0;
                double aop[NT], bop[NT];
#pragma unroll
                for (int T = 0; T < NT; T++) {
                    const int col = 16 * T + lc;
                    bop[T] = (T < Tc && s <= N && col < 2 * s) ? sC[coff(s, wch) + col] : 0.0;
                    aop[T] = bop[T] * gr;
                }
#pragma unroll
                for (int K = 0; K < NT; K++)
#pragma unroll
                    for (int I = K; I < NT; I++)
                        if (I < Tc) Mt[tidx(K, I)] = mfma(aop[K], bop[I], Mt[tidx(K, I)]);
            }
            // store as packed lower triangle: M[col_g][row_g] for col_g >= row_g
#pragma unroll
            for (int K = 0; K < NT; K++)
#pragma unroll
                for (int I = K; I < NT; I++)
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int rg = 16 * K + lq + 4 * jj, cg = 16 * I + lc;
                        if (cg >= rg) sM[lpk(cg, rg)] = Mt[tidx(K, I)][jj];
                    }
        }
        wsync();
        if ((ka.flags & 2) && b < 4 && it == 0) {
            double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
            for (int i = lane; i < LPK; i += 64) dbg[13300 + i] = sM[i];
        }

        TUM_TICK(3);
        // ---- blocked Cholesky, left-looking: M = L L'
        bool chol_ok = true;
#pragma unroll
        for (int J = 0; J < NT; J++) {
            // (1) update block column J with the block columns already factorised
            if (J > 0) {
#pragma unroll
                for (int I = J; I < NT; I++) {
                    d4 T;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int rg = 16 * I + lq + 4 * jj, cg = 16 * J + lc;
                        T[jj] = (rg >= cg) ? sM[lpk(rg, cg)] : 0.0;
                    }
#pragma unroll
                    for (int K = 0; K < J; K++)
#pragma unroll
                        for (int kc = 0; kc < 4; kc++) {
                            const double a = -sM[lpk(16 * I + lc, 16 * K + 4 * kc + lq)];
                            const double bb = sM[lpk(16 * J + lc, 16 * K + 4 * kc + lq)];
                            T = mfma(a, bb, T);
                        }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int rg = 16 * I + lq + 4 * jj, cg = 16 * J + lc;
                        if (rg >= cg) sM[lpk(rg, cg)] = T[jj];
                    }
                }
                wsync();
            }
            // (2) factorise the 16-wide panel: lane = row (bank 0: rows 0..63, bank 1: rows 64..79)
            double P0[16], P1[16];
            const int i0 = lane, i1 = 64 + lane;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const int jc = 16 * J + jj;
                P0[jj] = (i0 >= jc) ? sM[lpk(i0, jc)] : 0.0;
                P1[jj] = (lane < 16 && i1 >= jc) ? sM[lpk(i1, jc)] : 0.0;
            }
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const int jc = 16 * J + jj;
                const int pl = jc & 63;
                const double piv = (J < 4) ? rl(P0[jj], pl) : rl(P1[jj], pl);
                if (!(piv > 1e-300)) chol_ok = false;
                const double dg = sqrt(piv), inv = 1.0 / dg;
                if (lane == 0) sInvD[jc] = inv;
                if (i0 > jc) P0[jj] *= inv; else if (i0 == jc) P0[jj] = dg;
                if (i1 > jc) P1[jj] *= inv; else if (i1 == jc) P1[jj] = dg;
#pragma unroll
                for (int cc = jj + 1; cc < 16; cc++) {
                    const int pc = (16 * J + cc) & 63;
                    const double lcj = (J < 4) ? rl(P0[jj], pc) : rl(P1[jj], pc);
                    P0[cc] -= P0[jj] * lcj;
                    P1[cc] -= P1[jj] * lcj;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const int jc = 16 * J + jj;
                if (i0 >= jc) sM[lpk(i0, jc)] = P0[jj];
                if (lane < 16 && i1 >= jc) sM[lpk(i1, jc)] = P1[jj];
            }
            wsync();
        }
        if (!chol_ok) { qp_status = 3; break; }
        if ((ka.flags & 2) && b < 4 && it == 0) {
            double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
            for (int i = lane; i < LPK; i += 64) dbg[16540 + i] = sM[i];
        }

        TUM_TICK(4);
        // ---- predictor / corrector
        double dS[3][2], dT[3][2], dL[3][2], dMu[3][2];
        double dv0 = 0.0, dv1 = 0.0, alpha = 1.0, sigma = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            double rc1[3][2], rc2[3][2], rho[3][2];
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    rc1[rr][sd] = R.t[rr][sd] * R.lam[rr][sd];
                    rc2[rr][sd] = R.s[rr][sd] * R.mu[rr][sd];
                    if (pass == 1) {
                        // centring target floored just below tol_comp: keeps gamma = lam/t (and the
                        // conditioning of M) bounded once complementarity has converged
                        const double tau = fmax(sigma * gap, 0.1 * ka.tol_comp);
                        rc1[rr][sd] += dT[rr][sd] * dL[rr][sd] - tau;
                        rc2[rr][sd] += dS[rr][sd] * dMu[rr][sd] - tau;
                    }
                    rho[rr][sd] = -R.rt[rr][sd] + rc1[rr][sd] / R.lam[rr][sd]
                                  - (R.rs[rr][sd] + rc2[rr][sd] / R.s[rr][sd]) / Ds[rr][sd];
                }
            wsync();
            if (rowlane) {
                sWb[lane] = gam[0][0] * rho[0][0] - gam[0][1] * rho[0][1];
                sWr[2 * lane] = gam[1][0] * rho[1][0] - gam[1][1] * rho[1][1];
                sWr[2 * lane + 1] = gam[2][0] * rho[2][0] - gam[2][1] * rho[2][1];
            }
            wsync();
            double b0, b1;
            ctw(b0, b1);
            b0 = -rv0 - b0; b1 = -rv1 - b1;
            if (lane >= nv) b0 = 0.0;
            if (!(lane < 16 && 64 + lane < nv)) b1 = 0.0;
            if ((ka.flags & 2) && b < 4 && it == 0 && pass == 0) {
                double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
                dbg[19780 + lane] = b0;
                if (lane < 16) dbg[19780 + 64 + lane] = b1;
            }
            TUM_TICK(5);
            // forward substitution L y = b (column oriented)
            for (int j = 0; j < 64; j++) {
                const double yj = rl(b0, j) * sInvD[j];
                if (lane == j) b0 = yj;
                if (lane > j) b0 -= sM[lpk(lane, j)] * yj;
                if (lane < 16) b1 -= sM[lpk(64 + lane, j)] * yj;
            }
            for (int j = 64; j < NVP; j++) {
                const double yj = rl(b1, j - 64) * sInvD[j];
                if (lane == j - 64) b1 = yj;
                if (lane < 16 && lane > j - 64) b1 -= sM[lpk(64 + lane, j)] * yj;
            }
            // backward substitution L' x = y
            for (int j = NVP - 1; j >= 64; j--) {
                const double xj = rl(b1, j - 64) * sInvD[j];
                if (lane == j - 64) b1 = xj;
                if (lane < 16 && lane < j - 64) b1 -= sM[lpk(j, 64 + lane)] * xj;
                b0 -= sM[lpk(j, lane)] * xj;
            }
            for (int j = 63; j >= 0; j--) {
                const double xj = rl(b0, j) * sInvD[j];
                if (lane == j) b0 = xj;
                if (lane < j) b0 -= sM[lpk(j, lane)] * xj;
            }
            dv0 = b0; dv1 = b1;
            TUM_TICK(6);
            wsync();
            sDv[lane] = dv0;
            if (lane < 16) sDv[64 + lane] = dv1;
            wsync();
            if ((ka.flags & 2) && b < 4 && it == 0 && pass == 0) {
                double *dbg = ka.dbg + (size_t)b * ka.dbg_stride;
                dbg[19860 + lane] = dv0;
                if (lane < 16) dbg[19860 + 64 + lane] = dv1;
            }
            // C * dv for this lane's rows
            double cdv[3] = {0.0, 0.0, 0.0};
            if (rowlane) {
                cdv[0] = sDv[2 * lane + 1];
                const int s = lane + 1;
                const double *c0 = sC + coff(s, 0), *c1 = sC + coff(s, 1);
                double a0 = 0.0, a1 = 0.0;
                for (int c = 0; c < 2 * s; c++) {
                    const double x = sDv[c];
                    a0 += c0[c] * x; a1 += c1[c] * x;
                }
                cdv[1] = a0; cdv[2] = a1;
            }
            double amax = 1.0, lmu = 0.0;
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    const double eps = sd ? -1.0 : 1.0;
                    const double dl = -gam[rr][sd] * (eps * cdv[rr] + rho[rr][sd]);
                    const double dsl = (dl - R.rs[rr][sd] - rc2[rr][sd] / R.s[rr][sd]) / Ds[rr][sd];
                    const double dm = (-rc2[rr][sd] - R.mu[rr][sd] * dsl) / R.s[rr][sd];
                    const double dtt = (-rc1[rr][sd] - R.t[rr][sd] * dl) / R.lam[rr][sd];
                    dL[rr][sd] = dl; dS[rr][sd] = dsl; dMu[rr][sd] = dm; dT[rr][sd] = dtt;
                    if (rowlane) {
                        if (dtt < 0.0) amax = fmin(amax, -R.t[rr][sd] / dtt);
                        if (dsl < 0.0) amax = fmin(amax, -R.s[rr][sd] / dsl);
                        if (dl < 0.0) amax = fmin(amax, -R.lam[rr][sd] / dl);
                        if (dm < 0.0) amax = fmin(amax, -R.mu[rr][sd] / dm);
                    }
                }
            amax = wave_min(amax);
            if (pass == 0) {
                if (rowlane) {
#pragma unroll
                    for (int rr = 0; rr < 3; rr++)
#pragma unroll
                        for (int sd = 0; sd < 2; sd++)
                            lmu += (R.t[rr][sd] + amax * dT[rr][sd]) * (R.lam[rr][sd] + amax * dL[rr][sd])
                                 + (R.s[rr][sd] + amax * dS[rr][sd]) * (R.mu[rr][sd] + amax * dMu[rr][sd]);
                }
                const double mu_aff = wave_sum(lmu) / npairs;
                const double ratio = mu_aff / gap;
                sigma = ratio * ratio * ratio;
            } else {
                alpha = (amax >= 1.0) ? 1.0 : 0.995 * amax;
            }
        }
        TUM_TICK(7);
        if (alpha < 1e-12) { qp_status = 2; break; }
        v0 += alpha * dv0; v1 += alpha * dv1;
        const double om = 1.0 - alpha;
        rv0 *= om; rv1 *= om;
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                R.t[rr][sd] += alpha * dT[rr][sd]; R.s[rr][sd] += alpha * dS[rr][sd];
                R.lam[rr][sd] += alpha * dL[rr][sd]; R.mu[rr][sd] += alpha * dMu[rr][sd];
                R.rs[rr][sd] *= om; R.rt[rr][sd] *= om;
            }
    }
    const int status = acados_status(qp_status);

    TUM_TICK(8);
    // ------------------------------------------------------------ phase 4: expand, full step, cost
    wsync();
    sDv[lane] = v0;
    if (lane < 16) sDv[64 + lane] = v1;
    wsync();
    if (status == 0) {
        // dx_0 = g_0 ; dx_{k+1} = A_k dx_k + B_k du_k + b_k   (every lane runs the same recurrence)
        double dx[8];
#pragma unroll
        for (int i = 0; i < 8; i++) dx[i] = sG[i];
        auto pick = [&](const double (&a)[8]) {
            double m = a[0];
#pragma unroll
            for (int i = 1; i < 8; i++) if (lane == i) m = a[i];
            return m;
        };
        if (lane < 8) sX[lane] += pick(dx);
        for (int k = 0; k < N; k++) {
            const double *rec = sAB + k * ABS;
            const double du0 = sDv[2 * k], du1 = sDv[2 * k + 1];
            apply_A(rec, dx);
#pragma unroll
            for (int i = 0; i < 6; i++) dx[i] += rec[2 + i * 7 + 5] * du0 + rec[2 + i * 7 + 6] * du1 + rec[44 + i];
            dx[6] += dt * du1 + rec[50];
            dx[7] += dt * du0 + rec[51];
            if (lane < 8) sX[(k + 1) * NX + lane] += pick(dx);
        }
        sU[lane] += v0;
        if (lane < 16) sU[64 + lane] += v1;
    }
    wsync();
    // cost at the (new) iterate: stage terms scaled by dt, terminal unscaled, slack penalties pre-scaled
    double cl = 0.0;
    if (lane <= N) {
        const int k = lane;
        const double sc = (k < N) ? dt : 1.0;
        double acc = 0.0, e;
        e = sX[k * NX + 0] - yr[0]; acc += ((k < N) ? Wd[0] : We[0]) * e * e;
        e = sX[k * NX + 1] - yr[1]; acc += ((k < N) ? Wd[1] : We[1]) * e * e;
        e = wrap_yaw(sX[k * NX + 2]) - yr[2]; acc += ((k < N) ? Wd[2] : We[2]) * e * e;
        e = sX[k * NX + 3] - yr[3]; acc += ((k < N) ? Wd[3] : We[3]) * e * e;
        if (k < N) {
            e = sU[2 * k] - yr[4]; acc += Wd[4] * e * e;
            e = sU[2 * k + 1] - yr[5]; acc += Wd[5] * e * e;
        }
        cl = 0.5 * sc * acc;
    }
    if (rowlane) {
#pragma unroll
        for (int rr = 0; rr < 3; rr++)
#pragma unroll
            for (int sd = 0; sd < 2; sd++)
                cl += R.z[rr][sd] * R.s[rr][sd] + 0.5 * R.Z[rr][sd] * R.s[rr][sd] * R.s[rr][sd];
    }
    const double cost = wave_sum(cl);

    // ------------------------------------------------------------ stores
    for (int i = lane; i < (N + 1) * NX; i += 64) gX[i] = sX[i];
    for (int i = lane; i < nv; i += 64) gU[i] = sU[i];
    if (rowlane && ka.slack) {
        double *sl = ka.slack + (size_t)b * 6 * N;
        // order: [lower: box_k (N) | (bx_k, h_k) k=1..N] then the same for upper
#pragma unroll
        for (int sd = 0; sd < 2; sd++) {
            sl[sd * 3 * N + lane] = R.s[0][sd];
            sl[sd * 3 * N + N + 2 * lane] = R.s[1][sd];
            sl[sd * 3 * N + N + 2 * lane + 1] = R.s[2][sd];
        }
    }
    TUM_TICK(9);
    if ((ka.flags & 4) && lane == 0)
        for (int i = 0; i < 12; i++) ka.prof[(size_t)b * 12 + i] = pacc[i];
    if (lane == 0) {
        ka.cost[b] = cost;
        ka.status[b] = status;
        ka.qp_iter[b] = it;
        ka.qp_status[b] = qp_status;
        ka.res[b * 3 + 0] = res_stat; ka.res[b * 3 + 1] = res_ineq; ka.res[b * 3 + 2] = res_comp;
    }
}

// cold start on the device: X_k = x0, U = 0 (acados create / reset + set(i,'x',x0); NMPC_class.py:250-254)
__global__ void cold_start_kernel(double *X, double *U, const double *x0, int N, int batch)
{
    const int b = blockIdx.x;
    if (b >= batch) return;
    for (int i = threadIdx.x; i < (N + 1) * NX; i += blockDim.x) X[(size_t)b * (N + 1) * NX + i] = x0[(size_t)b * NX + (i & 7)];
    for (int i = threadIdx.x; i < N * NU; i += blockDim.x) U[(size_t)b * N * NU + i] = 0.0;
}

}  // namespace tum
