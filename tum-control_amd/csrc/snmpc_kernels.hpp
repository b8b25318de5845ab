// snmpc_kernels.hpp -- the coupled SNMPC OCP around the fused SQP-RTI kernel.
//
// Reference: Stochastic_NMPC/pred_model_dynamic_disc.py:121-220 (DISCRETE dynamics of the stacked state: nominal copy +
// n_s sample copies under ONE input), Stochastic_NMPC/SNMPC_acados_settings.py:57-194 (cost on the nominal copy, chance
// constraint E + kappa sqrt(Var) over the samples), Stochastic_NMPC/SNMPC_class.py:103-104,179-198 (stop flags of the
// uncertainty propagation horizon `uph`, solve).
//
// Structure that this build exploits instead of condensing an 88-state system:
//   * the stage matrix is block diagonal over the sample copies; the nominal copy of stage s <= uph is the PCE mean of
//     the sample copies (row 0 of A_pce), so  G_nom,s = sum_i a_i G^(i)_s  and the full-condensed QP keeps the shape
//     of the nominal one (2N variables, the same 3 soft rows per stage);
//   * from stage uph on the samples are frozen and never read again: the nominal recursion of the fused kernel takes
//     over from G_nom,uph;
//   * at stage s the sample matrices G^(i)_s only have 2s <= 2 uph non-zero columns.
// So the solve is three launches on one stream:
//   snmpc_prologue_kernel   sample stages: linearise ns*uph RK4 steps (lane = (stage, sample)), PCE weights of the chance
//                           rows, column recursions of all samples in registers (lane = column), hands G_nom,s, g_nom,s
//                           and the chance-constraint rows of the stages 1..uph to the fused kernel through `pro`
//   nmpc_rti_kernel<.,true> the fused kernel: cost rows / gg rows / Hessian of stages <= uph from `pro`, nominal recursion
//                           from stage uph, interior point, expansion of the nominal copy
//   snmpc_epilogue_kernel   full step of the sample copies (lane = sample)
#pragma once
#include "nmpc_device.hpp"

namespace tum {

constexpr int SN_NSMAX = 16;                 // max samples
constexpr int SN_LMAX = 16;                  // max PCE terms
constexpr int SN_UPHMAX = 31;                // sample columns 0..2 uph-1 and the g column 2 uph share one wavefront
constexpr int SN_PRO_G = 8 * 64;             // doubles per stage of G_nom in `pro`: [row][lane]
constexpr int SN_PRO_STAGE = SN_PRO_G + 64;  // + one chance-constraint row [lane]
constexpr int SN_ITEMS = SN_UPHMAX * SN_NSMAX;

struct SnArgs {
    int N, batch, ns, L, uph;
    double dt, kappa;
    Model mp;
    const double *X;          // [b][(N+1)*8]   nominal copy of the iterate
    const double *U;          // [b][N*2]
    double *XS;               // [b][N+1][ns][8] sample copies of the iterate
    const double *xs0;        // [b][ns][8]      initial condition of the samples (lbx_0 = ubx_0)
    const double *Apce;       // [L][ns]         PCE matrix (SNMPC_class.py:124), shared by the batch
    double *ws2;              // [b][uph*ns][ABS] sample linearisation records
    double *pro;              // [b][uph][SN_PRO_STAGE]
    const double *dv;         // [b][NVP]        QP solution of the fused kernel (epilogue)
    const int *status;        // [b]
};

// gg circle with the limits looked up at |v| (SNMPC_acados_settings.py:60-67,100-113): value and d/d(vl, vt, r, a)
__device__ __forceinline__ void h_con_vabs(const Model &p, double vl, double vt, double r, double a,
                                           double &h, double &g3, double &g4, double &g5, double &g7)
{
    const double vabs = sqrt(vl * vl + vt * vt);
    double ax, dax, ay, day;
    interp_lin(p.n_ggv, p.ggv_v, p.ggv_ax, vabs, ax, dax);
    interp_lin(p.n_ggv, p.ggv_v, p.ggv_ay, vabs, ay, day);
    if (a < 0.0) { ax = p.ax_brake; dax = 0.0; }
    const double alat = vl * r, nlon = a / ax, nlat = alat / ay;
    h = nlon * nlon + nlat * nlat;
    const double dvv = -2.0 * nlat * alat / (ay * ay) * day - 2.0 * nlon * a / (ax * ax) * dax;
    const double iv = (vabs > 0.0) ? 1.0 / vabs : 0.0;
    g3 = 2.0 * nlat * r / ay + dvv * vl * iv;
    g4 = dvv * vt * iv;
    g5 = 2.0 * nlat * vl / ay;
    g7 = 2.0 * nlon / ax;
}

// NSM: compile-time bound of the number of samples (register file of the column recursions: NSM x 8 doubles per lane)
template <int NSM>
__global__ void __launch_bounds__(64) snmpc_prologue_kernel(const SnArgs sa)
{
    __shared__ double sH[SN_ITEMS], sGh[SN_ITEMS * 4], sCoef[SN_ITEMS], sHval[SN_UPHMAX + 1];
    __shared__ double sRec[NSM * ABS];      // the records of one stage, all samples
    const int lane = threadIdx.x, b = blockIdx.x;
    if (b >= sa.batch) return;
    const int N = sa.N, ns = sa.ns, L = sa.L, uph = sa.uph;
    const double dt = sa.dt;
    const double *gX = sa.X + (size_t)b * (N + 1) * NX;
    const double *gU = sa.U + (size_t)b * N * NU;
    const double *gXS = sa.XS + (size_t)b * (N + 1) * ns * NX;
    double *ws2 = sa.ws2 + (size_t)b * uph * ns * ABS;
    double *pro = sa.pro + (size_t)b * uph * SN_PRO_STAGE;
    const int nitem = uph * ns;

    // ---- P1: one RK4 step with sensitivities per (stage, sample); chance-constraint terms of the sample
    for (int item = lane; item < nitem; item += 64) {
        const int k = item / ns, i = item - k * ns;
        const double *xp = gXS + ((size_t)k * ns + i) * NX;
        double xk[8], uk[2] = {gU[2 * k], gU[2 * k + 1]};
#pragma unroll
        for (int r = 0; r < 8; r++) xk[r] = xp[r];
        double xn[8], Sp[2], S[6][7];
        rk4_sens(sa.mp, xk, uk, dt, 1, xn, Sp, S);
        double *rec = ws2 + (size_t)item * ABS;
        rec[0] = Sp[0]; rec[1] = Sp[1];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 7; c++) rec[2 + r * 7 + c] = S[r][c];
        const double *xq = gXS + ((size_t)(k + 1) * ns + i) * NX;
#pragma unroll
        for (int r = 0; r < 8; r++) rec[44 + r] = xn[r] - xq[r];
        double h = 0.0, g3 = 0.0, g4 = 0.0, g5 = 0.0, g7 = 0.0;
        if (k >= 1) h_con_vabs(sa.mp, xk[3], xk[4], xk[5], xk[7], h, g3, g4, g5, g7);
        sH[item] = h;
        sGh[item * 4 + 0] = g3; sGh[item * 4 + 1] = g4; sGh[item * 4 + 2] = g5; sGh[item * 4 + 3] = g7;
    }
    __threadfence_block();      // the records are read back by other lanes below
    __syncthreads();

    // ---- P2: PCE coefficients of the sample values, weights d(E + kappa sqrt(Var)) / d h_i  (stages 1..uph-1)
    for (int item = lane; item < nitem; item += 64) {
        const int k = item / ns, i = item - k * ns;
        double w = 0.0;
        if (k >= 1) {
            double c0 = 0.0, var = 0.0, acc = 0.0;
            for (int l = 0; l < L; l++) {
                double cl = 0.0;
                for (int j = 0; j < ns; j++) cl += sa.Apce[l * ns + j] * sH[k * ns + j];
                if (l == 0) c0 = cl;
                else { var += cl * cl; acc += cl * sa.Apce[l * ns + i]; }
            }
            const double sd = sqrt(var);
            w = sa.Apce[i] + ((sd > 0.0) ? sa.kappa * acc / sd : 0.0);
            if (i == 0) sHval[k] = c0 + sa.kappa * sd;
        }
        sCoef[item] = w;
    }
    __syncthreads();

    // ---- P3: column recursions of all samples, lane = column of G (2 uph of them), lane 2 uph = the constant column g
    const int gl = 2 * uph;
    const bool isg = (lane == gl);
    const int jst = lane >> 1, r0 = lane & 1;
    double w[NSM][8];
    // the records of a stage are contiguous in ws2: staged through LDS one stage ahead (uniform-address vector loads of
    // 530 doubles per stage straight from L2 cost more than the arithmetic of this phase)
    constexpr int NCH = (NSM * ABS + 63) / 64;
    const int nrec = ns * ABS;
    double pre[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; pre[c] = (uph > 0 && idx < nrec) ? ws2[idx] : 0.0; }
#pragma unroll
    for (int i = 0; i < NSM; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) w[i][r] = 0.0;
        if (i < ns && isg) {
#pragma unroll
            for (int r = 0; r < 8; r++) w[i][r] = sa.xs0[((size_t)b * ns + i) * NX + r] - gXS[(size_t)i * NX + r];
        }
    }
    for (int k = 0; k < uph; k++) {
        const int s = k + 1;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; if (idx < nrec) sRec[idx] = pre[c]; }
        __syncthreads();
        if (k + 1 < uph) {
#pragma unroll
            for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; pre[c] = (idx < nrec) ? ws2[(size_t)(k + 1) * nrec + idx] : 0.0; }
        }
        double gn[8], row = 0.0;
#pragma unroll
        for (int r = 0; r < 8; r++) gn[r] = 0.0;
        const double sel = (lane < gl && jst == k) ? 1.0 : 0.0, selg = isg ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < NSM; i++) {
            if (i < ns) {
                const double *rec = sRec + i * ABS;
                apply_A(rec, w[i]);
#pragma unroll
                for (int r = 0; r < 6; r++) w[i][r] += sel * rec[2 + r * 7 + 5 + r0];
                w[i][6] += sel * (r0 ? dt : 0.0);
                w[i][7] += sel * (r0 ? 0.0 : dt);
#pragma unroll
                for (int r = 0; r < 8; r++) w[i][r] += selg * rec[44 + r];
                const double a = sa.Apce[i];
#pragma unroll
                for (int r = 0; r < 8; r++) gn[r] += a * w[i][r];
                if (isg) {   // the nominal copy's own defect: sum_i a_i X^(i)_s - X_nom,s
                    const double *xq = gXS + ((size_t)s * ns + i) * NX;
#pragma unroll
                    for (int r = 0; r < 8; r++) gn[r] += a * xq[r];
                }
                if (s < uph) {
                    const int it = s * ns + i;
                    row += sCoef[it] * (sGh[it * 4 + 0] * w[i][3] + sGh[it * 4 + 1] * w[i][4] + sGh[it * 4 + 2] * w[i][5] +
                                        sGh[it * 4 + 3] * w[i][7]);
                }
            }
        }
        if (isg) {
#pragma unroll
            for (int r = 0; r < 8; r++) gn[r] -= gX[s * NX + r];
            if (s < uph) row += sHval[s];
        }
        double *pg = pro + (size_t)k * SN_PRO_STAGE;
#pragma unroll
        for (int r = 0; r < 8; r++) pg[r * 64 + lane] = gn[r];
        pg[SN_PRO_G + lane] = row;
    }
}

// full step of the sample copies: dx^(i)_0 = xs0 - X^(i)_0, dx^(i)_{k+1} = A dx^(i)_k + B du_k + b for k < uph; frozen
// afterwards (X^(i)_k = X^(i)_uph for k > uph, pred_model_dynamic_disc.py:203). One lane per sample.
__global__ void __launch_bounds__(64) snmpc_epilogue_kernel(const SnArgs sa)
{
    const int lane = threadIdx.x, b = blockIdx.x;
    if (b >= sa.batch || sa.status[b] != 0) return;
    const int N = sa.N, ns = sa.ns, uph = sa.uph, i = lane;
    if (i >= ns) return;
    double *gXS = sa.XS + (size_t)b * (N + 1) * ns * NX;
    const double *ws2 = sa.ws2 + (size_t)b * uph * ns * ABS;
    const double *dv = sa.dv + (size_t)b * NVP;
    double dx[8], xnew[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double x = gXS[(size_t)i * NX + r];
        dx[r] = sa.xs0[((size_t)b * ns + i) * NX + r] - x;
        xnew[r] = x + dx[r];
        gXS[(size_t)i * NX + r] = xnew[r];
    }
    for (int k = 0; k < uph; k++) {
        const double *rec = ws2 + (size_t)(k * ns + i) * ABS;
        const double du0 = dv[2 * k], du1 = dv[2 * k + 1];
        apply_A(rec, dx);
#pragma unroll
        for (int r = 0; r < 6; r++) dx[r] += rec[2 + r * 7 + 5] * du0 + rec[2 + r * 7 + 6] * du1;
        dx[6] += sa.dt * du1; dx[7] += sa.dt * du0;
#pragma unroll
        for (int r = 0; r < 8; r++) dx[r] += rec[44 + r];
        double *xq = gXS + ((size_t)(k + 1) * ns + i) * NX;
#pragma unroll
        for (int r = 0; r < 8; r++) { xnew[r] = xq[r] + dx[r]; xq[r] = xnew[r]; }
    }
    for (int k = uph + 1; k <= N; k++) {
        double *xq = gXS + ((size_t)k * ns + i) * NX;
#pragma unroll
        for (int r = 0; r < 8; r++) xq[r] = xnew[r];
    }
}

// cold start of the sample copies: X^(i)_k = xs0^(i) for all k (SNMPC_class.py:126-127)
__global__ void snmpc_cold_start_kernel(double *XS, const double *xs0, int N, int ns, int batch)
{
    const int b = blockIdx.x;
    if (b >= batch) return;
    const int per = ns * NX;
    for (int i = threadIdx.x; i < (N + 1) * per; i += blockDim.x) XS[(size_t)b * (N + 1) * per + i] = xs0[(size_t)b * per + i % per];
}

}  // namespace tum
