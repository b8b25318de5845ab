// aux_kernels.hpp -- the small kernels either side of the SQP-RTI solve.
#pragma once
#include "nmpc_device.hpp"

namespace tum {

// K6a  sigma-point fan-out: instance p*S1 + 0 = pose p (nominal), p*S1 + s = pose p + offs[s-1]
//      (Stochastic_NMPC/stochastic_mpc_utils.py:78-91 compute_x0dist, as a batch axis)
__global__ void sigma_fanout_kernel(double *x0, const double *pose, const double *offs, int P, int S1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over P*S1*8
    if (i >= P * S1 * NX) return;
    const int c = i & 7, inst = i >> 3, p = inst / S1, s = inst - p * S1;
    x0[i] = pose[p * NX + c] + (s > 0 ? offs[(s - 1) * NX + c] : 0.0);
}

// K6b  PCE moments over scenario groups: c = A v (A: L x S least-squares PCE matrix), E = c_0,
//      Var = sum_{k>=1} c_k^2  (Stochastic_NMPC/SNMPC_acados_settings.py:116-133).
//      V: per-instance records `rec` doubles apart, the m quantities at their start (read straight from the iterate; row
//      p*S1 is the nominal instance and is skipped); one thread per (group, component); A is small (L1 / scalar path).
__global__ void pce_moments_kernel(const double *V, size_t rec, const double *A, int P, int S1, int m, int L, double *mean, double *var)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over P*m
    if (i >= P * m) return;
    const int p = i / m, c = i - p * m, S = S1 - 1;
    const double *v = V + ((size_t)p * S1 + 1) * rec + c;
    double e = 0.0, va = 0.0;
    for (int k = 0; k < L; k++) {
        double ck = 0.0;
        for (int s = 0; s < S; s++) ck += A[k * S + s] * v[(size_t)s * rec];
        if (k == 0) e = ck; else va += ck * ck;
    }
    mean[i] = e; var[i] = va;
}

// K7  R2NMPC constraint tightening (Reduced_Robustified_NMPC_class.py:286-366): per instance
//      Sigma_{k+1} = A_k Sigma_k A_k' + B W B'   (Robust_NMPC_pred_model_utils.py:221-223),
//      for k = 1..uph-1: steering back-off sqrt(Sigma_k[6][6]), gg back-off sqrt(grad_h(x_k)' Sigma_k grad_h(x_k)),
//      lbx_k = delta_min + b_d, ubx_k = delta_max - b_d, uh_k = uh_nom - b_h; stages uph..N-1 reuse the last pair.
//      One wavefront per instance, lane (i,j) owns entry (i,j) of the 8x8 covariance; the 8x8 blocks go
//      through LDS (the "small-block LDS path" of BASELINE config 5).
//      A_k comes from the full blocks the fused kernel keeps for get_from_qp_in (qpin: [b][N][88]) or, when the pipeline
//      solved (rec != null: [b][N+1][rec_stride]), straight from the compact stage records: identity on (px, py, psi, delta,
//      a), Sp[2] in the psi column of (px, py), S[6][5] in the (vl, vt, r, delta, a) columns of the first six rows.
__global__ void __launch_bounds__(256) r2_backoff_kernel(const double *qpin, const double *X, double *bnd, const Model mp,
                                                         const double *Sigma0, const double *BWB, int N, int uph, int batch,
                                                         double dmin, double dmax, double uh_nom, double *backoff_out,
                                                         const int *status, const double *rec = nullptr, int rec_stride = 0)
{
    __shared__ double sA[4][64], sS[4][64], sT[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + w;
    // an instance whose solve failed keeps its previous bounds (`if status == 0:`, Reduced_Robustified_NMPC_class.py:276)
    const bool on = b < batch && (!status || status[b] == 0);
    const int i = lane >> 3, j = lane & 7;
    double sig = Sigma0[lane];
    const double bwb = BWB[lane];
    double bd = 0.0, bh = 0.0;
    const int NB = N + 1;
    for (int k = 0; k < uph && k < N; k++) {
        if (k > 0) {
            // back-offs from Sigma_k at the new iterate x_k
            sS[w][lane] = sig;
            __syncthreads();
            if (on) {
                const double *x = X + ((size_t)b * (N + 1) + k) * NX;
                double h, g3, g5, g7;
                h_con(mp, x[3], x[5], x[7], h, g3, g5, g7);
                const double q = g3 * (g3 * sS[w][3 * 8 + 3] + g5 * sS[w][3 * 8 + 5] + g7 * sS[w][3 * 8 + 7])
                               + g5 * (g3 * sS[w][5 * 8 + 3] + g5 * sS[w][5 * 8 + 5] + g7 * sS[w][5 * 8 + 7])
                               + g7 * (g3 * sS[w][7 * 8 + 3] + g5 * sS[w][7 * 8 + 5] + g7 * sS[w][7 * 8 + 7]);
                bh = sqrt(q);
                bd = sqrt(sS[w][6 * 8 + 6]);
                if (lane == 0) {
                    double *bb = bnd + (size_t)b * 6 * NB;
                    bb[2 * NB + k] = dmin + bd; bb[3 * NB + k] = dmax - bd; bb[5 * NB + k] = uh_nom - bh;
                    if (backoff_out) { backoff_out[((size_t)b * N + k) * 2] = bd; backoff_out[((size_t)b * N + k) * 2 + 1] = bh; }
                }
            }
            __syncthreads();
        }
        // Sigma <- A Sigma A' + B W B'
        if (rec) {
            const double *r = rec + ((size_t)b * (N + 1) + k) * rec_stride;
            double a = (i == j && (i < 3 || i >= 6)) ? 1.0 : 0.0;
            if (on && j == 2 && i < 2) a = r[i];
            if (on && j >= 3 && i < 6) a = r[2 + i * 7 + (j - 3)];
            sA[w][lane] = on ? a : 0.0;
        } else sA[w][lane] = on ? qpin[((size_t)b * N + k) * 88 + lane] : 0.0;     // A row-major 8x8
        sS[w][lane] = sig;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < 8; l++) t += sA[w][i * 8 + l] * sS[w][l * 8 + j];
        sT[w][lane] = t;
        __syncthreads();
        double s2 = bwb;
#pragma unroll
        for (int l = 0; l < 8; l++) s2 += sT[w][i * 8 + l] * sA[w][j * 8 + l];
        sig = s2;
        __syncthreads();
    }
    if (on && lane == 0) {
        double *bb = bnd + (size_t)b * 6 * NB;
        for (int k = uph; k < N; k++) {
            bb[2 * NB + k] = dmin + bd; bb[3 * NB + k] = dmax - bd; bb[5 * NB + k] = uh_nom - bh;
            if (backoff_out) { backoff_out[((size_t)b * N + k) * 2] = bd; backoff_out[((size_t)b * N + k) * 2 + 1] = bh; }
        }
    }
}

// result summary of every instance as 5 doubles [u0_jerk, u0_steering_rate, cost, status, qp_iter]: one buffer, one
// collective for the rooted gather of the multi-GPU job
__global__ void pack_summary_kernel(const double *U, const double *cost, const int *status, const int *qp_iter, int N, int b0, int nb, double *dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int b = b0 + i;
    dst[5 * i + 0] = U[(size_t)b * N * NU]; dst[5 * i + 1] = U[(size_t)b * N * NU + 1];
    dst[5 * i + 2] = cost[b]; dst[5 * i + 3] = (double)status[b]; dst[5 * i + 4] = (double)qp_iter[b];
}

// the full-W array of a capsule from its diagonal one (first cost_set 'W' with an off-diagonal entry): n = batch x (N + 1) stage slots
__global__ void wf_from_diag_kernel(const double *W, double *Wf, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int e = 0; e < 36; e++) Wf[i * 36 + e] = (e / 6 == e % 6) ? W[i * 6 + e / 6] : 0.0;
}

// The same plus the whole iterate into (host-mapped) slabs: for small batches one launch instead of the summary kernel and two
// copy commands (tum_ocp_results_async with_iterate; 3 KB per instance across PCIe as plain stores).
// ts (nullable): [1] takes the device's wall clock (constant-rate counter) when this kernel starts -- the end of the step's device
// time, whose start stage_in_kernel leaves in [0]: a step is timed without events on the stream (each one a 5 us gap)
__global__ void pack_results_kernel(const double *X, const double *U, const double *cost, const int *status, const int *qp_iter, int N, int nb,
                                    double *dsum, double *hX, double *hU, unsigned long long *ts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (ts && i == 0) ts[1] = wall_clock64();
    if (i < nb) {
        dsum[5 * i + 0] = U[(size_t)i * N * NU]; dsum[5 * i + 1] = U[(size_t)i * N * NU + 1];
        dsum[5 * i + 2] = cost[i]; dsum[5 * i + 3] = (double)status[i]; dsum[5 * i + 4] = (double)qp_iter[i];
    }
    const size_t nX = (size_t)nb * (N + 1) * NX, nU = (size_t)nb * N * NU;
    for (size_t k = i; k < nX; k += nth) hX[k] = X[k];
    for (size_t k = i; k < nU; k += nth) hU[k] = U[k];
}

// x0 | yref of a step from the capsule's (host-mapped) staging area into their device arrays: one launch instead of two copy
// commands (tum_ocp_step_async, small batches)
__global__ void stage_in_kernel(const double *hin, double *x0, int nx0, double *yref, int nyr, int have_x0, int have_yref, unsigned long long *ts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (ts && i == 0) ts[0] = wall_clock64();
    if (have_x0) for (int k = i; k < nx0; k += nth) x0[k] = hin[k];
    if (have_yref) for (int k = i; k < nyr; k += nth) yref[k] = hin[nx0 + k];
}

// The pinned shadow of a small capsule's per-step setters (tum_ocp_set "yref" per stage, constraints_set lbx_0 / ubx_0) into the device
// arrays, in front of the next solve: x0 when it was set, and the yref records of the stages in `mask` (bit k = stage k, `nst` stages per
// instance) -- what no setter touched keeps its device value (a device-to-device upload may have written it). ts[0]: the device clock at
// the start of a synchronous solve.
__global__ void stage_in_masked_kernel(const double *hin, double *x0, int nx0, double *yref, int nyr, int nst, int have_x0, unsigned long long mask,
                                       unsigned long long *ts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (ts && i == 0) ts[0] = wall_clock64();
    if (have_x0) for (int k = i; k < nx0; k += nth) x0[k] = hin[k];
    if (mask) for (int k = i; k < nyr; k += nth) if ((mask >> ((k / 6) % nst)) & 1ull) yref[k] = hin[nx0 + k];
}

}  // namespace tum