// accuracy of fast_sincos against the library routine over the argument ranges the model uses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../tum-control_amd/csrc/nmpc_device.hpp"
__global__ void k2(const double *x, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    o[2 * i] = fabs(tum::fast_atan(x[i]) - atan(x[i]));
    const double p = fabs(x[i]) + 1e-6;
    o[2 * i + 1] = fabs(tum::fast_sqrt_pos(p) - sqrt(p)) / sqrt(p);
}
__global__ void k(const double *x, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double s, c, s2, c2; tum::fast_sincos(x[i], &s, &c); sincos(x[i], &s2, &c2);
    o[2 * i] = fabs(s - s2); o[2 * i + 1] = fabs(c - c2);
}
int main() {
    const int n = 1 << 20; double *hx = new double[n], *ho = new double[2 * n];
    double worst[3] = {0, 0, 0};
    const double ranges[3] = {3.0, 100.0, 1e4};
    for (int r = 0; r < 3; r++) {
        for (int i = 0; i < n; i++) hx[i] = ranges[r] * (2.0 * (double)rand() / RAND_MAX - 1.0);
        double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 2 * n * 8);
        hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
        hipMemcpy(ho, dout, 2 * n * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 2 * n; i++) if (ho[i] > worst[r]) worst[r] = ho[i];
        printf("|x| < %g: max abs error vs sincos %.3e\n", ranges[r], worst[r]);
        hipLaunchKernelGGL(k2, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
        hipMemcpy(ho, dout, 2 * n * 8, hipMemcpyDeviceToHost);
        double wa = 0, wsq = 0;
        for (int i = 0; i < n; i++) { if (ho[2 * i] > wa) wa = ho[2 * i]; if (ho[2 * i + 1] > wsq) wsq = ho[2 * i + 1]; }
        printf("|x| < %g: atan max abs error %.3e, sqrt max rel error %.3e\n", ranges[r], wa, wsq);
        hipFree(dx); hipFree(dout);
    }
    return 0;
}
