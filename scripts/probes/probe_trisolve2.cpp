// Probe 2: restructured micro-block chain (separate result vector, minimal dependent ops, explicit prefetch) + rcp accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ double rl(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int USE_LDS, int WRITE_Y>
__global__ void k_new(double* out, long long* cyc, const double* Lg) {
    extern __shared__ double sM[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 3240; i += 64) sM[i] = Lg[i];
    __syncthreads();
    double b0 = 1.0 + lane * 0.01, yv = 0.0;
    const int myrow0 = lane * (lane + 1) / 2;
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < 16; rep++) {
        // prefetch block 0
        double l10, l20, l21, l30, l31, l32, m0, m1, m2, m3;
        {
            const int c0 = 0;
            const int r1 = (c0 + 1) * (c0 + 2) / 2 + c0, r2 = (c0 + 2) * (c0 + 3) / 2 + c0, r3 = (c0 + 3) * (c0 + 4) / 2 + c0;
            l10 = sM[r1]; l20 = sM[r2]; l21 = sM[r2 + 1]; l30 = sM[r3]; l31 = sM[r3 + 1]; l32 = sM[r3 + 2];
            const int o0 = myrow0 + ((lane > c0 + 3) ? c0 : 0);
            m0 = sM[o0]; m1 = sM[o0 + 1]; m2 = sM[o0 + 2]; m3 = sM[o0 + 3];
        }
#pragma unroll 4
        for (int c0 = 0; c0 < 64; c0 += 4) {
            // prefetch next block
            const int cn = (c0 + 4 < 64) ? c0 + 4 : 0;
            const int r1 = (cn + 1) * (cn + 2) / 2 + cn, r2 = (cn + 2) * (cn + 3) / 2 + cn, r3 = (cn + 3) * (cn + 4) / 2 + cn;
            double nl10, nl20, nl21, nl30, nl31, nl32, nm0, nm1, nm2, nm3;
            if (USE_LDS) {
                nl10 = sM[r1]; nl20 = sM[r2]; nl21 = sM[r2 + 1]; nl30 = sM[r3]; nl31 = sM[r3 + 1]; nl32 = sM[r3 + 2];
                const int o0 = myrow0 + ((lane > cn + 3) ? cn : 0);
                nm0 = sM[o0]; nm1 = sM[o0 + 1]; nm2 = sM[o0 + 2]; nm3 = sM[o0 + 3];
            } else { nl10 = 1e-3; nl20 = 2e-3; nl21 = 3e-3; nl30 = 1e-3; nl31 = 2e-3; nl32 = 1e-3; nm0 = 1e-3 * lane; nm1 = 2e-3; nm2 = 1e-3; nm3 = 3e-3; }
            // current block
            const double g0 = rl(b0, c0), g1 = rl(b0, c0 + 1), g2 = rl(b0, c0 + 2), g3 = rl(b0, c0 + 3);
            const double y0 = g0;
            double p = b0 - m0 * y0;                 // off the critical path
            const double y1 = g1 - l10 * y0;
            const double h2 = g2 - l20 * y0, h3 = g3 - l30 * y0;
            p -= m1 * y1;
            const double y2 = h2 - l21 * y1;
            const double h3b = h3 - l31 * y1;
            p -= m2 * y2;
            const double y3 = h3b - l32 * y2;
            b0 = p - m3 * y3;
            const int k4 = lane - c0;
            if (WRITE_Y) yv = (k4 == 0) ? y0 : (k4 == 1) ? y1 : (k4 == 2) ? y2 : (k4 == 3) ? y3 : yv;
            l10 = nl10; l20 = nl20; l21 = nl21; l30 = nl30; l31 = nl31; l32 = nl32; m0 = nm0; m1 = nm1; m2 = nm2; m3 = nm3;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = b0 + yv;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rcp(const double* x, double* r0, double* r1, double* r2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v); r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r); r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r); r2[i] = r;
}
int main() {
    std::vector<double> L(3240); for (int i = 0; i < 3240; i++) L[i] = 1e-3 * ((i * 37) % 11);
    double *dL, *dO; long long* dC; hipMalloc(&dL, 3240 * 8); hipMalloc(&dO, 4096 * 64 * 8); hipMalloc(&dC, 4096 * 8);
    hipMemcpy(dL, L.data(), 3240 * 8, hipMemcpyHostToDevice);
    std::vector<long long> c(4096);
    for (int nb : {1, 1024}) {
        k_new<1,1><<<nb, 64, 26000>>>(dO, dC, dL); hipDeviceSynchronize(); hipMemcpy(c.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("blocks %d  restructured LDS+chain+y : %.1f cycles per micro-block\n", nb, c[0] / (16.0 * 16));
        k_new<0,1><<<nb, 64, 26000>>>(dO, dC, dL); hipDeviceSynchronize(); hipMemcpy(c.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("blocks %d  restructured chain+y only: %.1f cycles per micro-block\n", nb, c[0] / (16.0 * 16));
        k_new<0,0><<<nb, 64, 26000>>>(dO, dC, dL); hipDeviceSynchronize(); hipMemcpy(c.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("blocks %d  restructured chain only  : %.1f cycles per micro-block\n", nb, c[0] / (16.0 * 16));
        k_new<1,0><<<nb, 64, 26000>>>(dO, dC, dL); hipDeviceSynchronize(); hipMemcpy(c.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("blocks %d  restructured LDS+chain   : %.1f cycles per micro-block\n", nb, c[0] / (16.0 * 16));
    }
    const int n = 1 << 16; std::vector<double> x(n), a(n), b(n), cc(n);
    for (int i = 0; i < n; i++) x[i] = std::exp((i % 4001) * 0.02 - 40.0) * (1.0 + (i * 7919 % 1000) * 1e-3);
    double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k_rcp<<<n / 256, 256>>>(dx, d0, d1, d2); hipDeviceSynchronize();
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(cc.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) { double t = 1.0 / x[i]; e0 = fmax(e0, fabs(a[i] - t) / t); e1 = fmax(e1, fabs(b[i] - t) / t); e2 = fmax(e2, fabs(cc[i] - t) / t); }
    printf("v_rcp_f64 max rel err: raw %.3e, 1 Newton step %.3e, 2 steps %.3e\n", e0, e1, e2);
    return 0;
}
