// common_kernels.hpp -- what every kernel variant shares: phase timers, the acados status code, the row-state accessor of the
// interior point kernels, the longest-first schedule and the cold start.
#pragma once
#include "nmpc_device.hpp"

namespace tum {

// optional in-kernel phase timers (flags & 4): cycles per phase accumulated into ka.prof[b][12]
#define TUM_TICK(slot) do { asm volatile("; TUM_MARK " #slot); if (PROF) { const long long t_ = __builtin_readcyclecounter(); pacc[slot] += t_ - tprev; tprev = t_; } } while (0)

__device__ __forceinline__ int acados_status(int qp_status) { return (qp_status == 0 || qp_status == 1) ? 0 : 4; }

// row-side state of the interior point kernels: rowst[field][slot * 2 + side], fields s, t, lam, mu, rs, rt
#define ROWF(f, k) rowst[f][k]

// Longest-first schedule for the NEXT solve: workgroup i gets the instance with the i-th largest iteration count of THIS
// solve (counting sort, one workgroup). A batch is only a few rounds of resident wavefronts (4096 instances = 4 rounds of
// 1024), and the time of an instance is proportional to its iteration count (4..15), so in natural order the last round
// leaves most of the GPU idle while a few long instances finish: measured 2.62 ms natural order, 2.17 ms longest-first,
// 2.34 ms shortest-first for the same 4096 instances. Iteration counts of consecutive solves of an MPC are strongly
// correlated (and identical for a repeated batch), which makes the last solve a good predictor.
__global__ void __launch_bounds__(1024) lpt_order_kernel(const int *qp_iter, int *order, int batch)
{
    __shared__ int hist[64], offs[64];
    const int t = threadIdx.x;
    if (t < 64) hist[t] = 0;
    __syncthreads();
    for (int i = t; i < batch; i += 1024) { int k = qp_iter[i]; k = k < 0 ? 0 : (k > 63 ? 63 : k); atomicAdd(&hist[k], 1); }
    __syncthreads();
    if (t == 0) { int acc = 0; for (int k = 63; k >= 0; k--) { offs[k] = acc; acc += hist[k]; } }
    __syncthreads();
    for (int i = t; i < batch; i += 1024) { int k = qp_iter[i]; k = k < 0 ? 0 : (k > 63 ? 63 : k); order[atomicAdd(&offs[k], 1)] = i; }
}

// cold start on the device: X_k = x0, U = 0 (acados create / reset + set(i,'x',x0); NMPC_class.py:250-254)
__global__ void cold_start_kernel(double *X, double *U, const double *x0, int N, int batch, double *qp_lam)
{
    const int b = blockIdx.x;
    if (b >= batch) return;
    if (qp_lam && threadIdx.x == 0) qp_lam[(size_t)b * (6 * N + 2) + 6 * N] = 0.0;          // (the interior point method of the next solve starts cold as well)
    for (int i = threadIdx.x; i < (N + 1) * NX; i += blockDim.x) X[(size_t)b * (N + 1) * NX + i] = x0[(size_t)b * NX + (i & 7)];
    for (int i = threadIdx.x; i < N * NU; i += blockDim.x) U[(size_t)b * N * NU + i] = 0.0;
}

}  // namespace tum
