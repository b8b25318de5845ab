// Probe: cost per 4-wide micro-block of the forward substitution chain (readlane broadcast -> uniform 4x4 solve -> row FMAs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double rl(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double* out, long long* cyc, const double* Lg) {
    extern __shared__ double sM[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 3240; i += 64) sM[i] = Lg[i];
    __syncthreads();
    double b0 = 1.0 + lane * 0.01, b1 = 0.5;
    const int myrow0 = lane * (lane + 1) / 2, myrow1 = (64 + (lane & 15)) * (65 + (lane & 15)) / 2;
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < 16; rep++) {
#pragma unroll 4
        for (int c0 = 0; c0 < 64; c0 += 4) {
            double l10, l20, l21, l30, l31, l32, m0, m1, m2, m3, n0, n1, n2, n3;
            const bool below = lane > c0 + 3;
            if (MODE == 0) {
                const int r1 = (c0 + 1) * (c0 + 2) / 2 + c0, r2 = (c0 + 2) * (c0 + 3) / 2 + c0, r3 = (c0 + 3) * (c0 + 4) / 2 + c0;
                l10 = sM[r1]; l20 = sM[r2]; l21 = sM[r2 + 1]; l30 = sM[r3]; l31 = sM[r3 + 1]; l32 = sM[r3 + 2];
                const int o0 = myrow0 + (below ? c0 : 0);
                m0 = sM[o0]; m1 = sM[o0 + 1]; m2 = sM[o0 + 2]; m3 = sM[o0 + 3];
                const int o1 = myrow1 + c0;
                n0 = sM[o1]; n1 = sM[o1 + 1]; n2 = sM[o1 + 2]; n3 = sM[o1 + 3];
            } else {
                l10 = 1e-3; l20 = 2e-3; l21 = 3e-3; l30 = 1e-3; l31 = 2e-3; l32 = 1e-3;
                m0 = 1e-3 * lane; m1 = 2e-3; m2 = 1e-3; m3 = 3e-3; n0 = n1 = n2 = n3 = 1e-3;
            }
            const double y0 = rl(b0, c0);
            const double y1 = rl(b0, c0 + 1) - l10 * y0;
            const double y2 = rl(b0, c0 + 2) - l20 * y0 - l21 * y1;
            const double y3 = rl(b0, c0 + 3) - l30 * y0 - l31 * y1 - l32 * y2;
            if (MODE == 2) {   // no readlane feedback: chain broken
                b0 -= m0 * 1e-3 + m1 * 2e-3;
            } else {
                if (below) b0 -= m0 * y0 + m1 * y1 + m2 * y2 + m3 * y3;
                if (lane == c0 + 1) b0 = y1;
                if (lane == c0 + 2) b0 = y2;
                if (lane == c0 + 3) b0 = y3;
                if (lane < 16) b1 -= n0 * y0 + n1 * y1 + n2 * y2 + n3 * y3;
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = b0 + b1;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    std::vector<double> L(3240); for (int i = 0; i < 3240; i++) L[i] = 1e-3 * ((i * 37) % 11);
    double *dL, *dO; long long* dC; hipMalloc(&dL, 3240 * 8); hipMalloc(&dO, 4096 * 64 * 8); hipMalloc(&dC, 4096 * 8);
    hipMemcpy(dL, L.data(), 3240 * 8, hipMemcpyHostToDevice);
    std::vector<long long> c(4096);
    for (int nb : {1, 1024}) {
        k<0><<<nb, 64, 26000>>>(dO, dC, dL); hipDeviceSynchronize(); hipMemcpy(c.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("blocks %d  LDS+chain : %.1f cycles per micro-block\n", nb, c[0] / (16.0 * 16));
        k<1><<<nb, 64, 26000>>>(dO, dC, dL); hipDeviceSynchronize(); hipMemcpy(c.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("blocks %d  chain only: %.1f cycles per micro-block\n", nb, c[0] / (16.0 * 16));
        k<2><<<nb, 64, 26000>>>(dO, dC, dL); hipDeviceSynchronize(); hipMemcpy(c.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("blocks %d  no feedback: %.1f cycles per micro-block\n", nb, c[0] / (16.0 * 16));
    }
    return 0;
}
