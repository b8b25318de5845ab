// probe_mfma_4x4x4.cpp -- lane layout and issue time of v_mfma_f64_4x4x4_4b_f64 on gfx950.
// For every A lane a: A = one-hot(a), B[lane] = lane + 1, C = 0  ->  D[o] = sum of the B values (lane + 1) that meet A lane a in output lane o.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void layout(double *out)
{
    const int l = threadIdx.x;
    for (int a = 0; a < 64; a++) {
        const double A = (l == a) ? 1.0 : 0.0, B = l + 1.0;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, 0.0, 0, 0, 0);
        out[a * 64 + l] = d;
    }
}
__global__ void timing(double *out, long long *cyc)
{
    const int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 256; i++) d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
    long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 64; i++) {
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
    }
    long long t2 = __builtin_readcyclecounter();
    out[l] = d0 + d1 + d2 + d3;
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main()
{
    double *d; long long *c; hipMalloc(&d, 64 * 64 * 8); hipMalloc(&c, 16);
    layout<<<1, 64>>>(d);
    static double h[64 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // for every output lane o and A lane a: which B lane (value - 1) contributed
    printf("A lane a -> (output lane o : B lane) pairs\n");
    for (int a = 0; a < 64; a++) {
        printf("a=%2d:", a);
        for (int o = 0; o < 64; o++) if (h[a * 64 + o] != 0.0) printf(" %d:%d", o, (int)h[a * 64 + o] - 1);
        printf("\n");
    }
    timing<<<1, 64>>>(d, c);
    long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    printf("dependent chain: %.1f cycles per mfma_f64_4x4x4; four independent accumulators: %.1f cycles per mfma\n", hc[0] / 256.0, hc[1] / 256.0);
    return 0;
}
