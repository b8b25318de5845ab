// probe_dpp64_bcast.cpp -- v_fmac_f64_dpp ... row_newbcast:n on gfx950 (the only DPP control 64-bit vector arithmetic takes): which lane does a lane
// read, and what does the instruction cost a lone wavefront next to a plain v_fma_f64.
//   hipcc --offload-arch=gfx950 -O3 -o exp_libs/probe_dpp64_bcast scripts/probes/probe_dpp64_bcast.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __device__ double bc(double x)
{
    double acc = 0.0, one = 1.0;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(one), "n"(N));
    return acc;
}
__global__ void layout(double *out)
{
    const int l = threadIdx.x;
    const double x = l;
    out[0 * 64 + l] = bc<0>(x); out[1 * 64 + l] = bc<1>(x); out[2 * 64 + l] = bc<5>(x); out[3 * 64 + l] = bc<15>(x);
}
__global__ void timing(double *out, long long *cyc)
{
    const int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l, f[8] = {0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8};
    asm volatile("" : "+v"(a), "+v"(b));
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 128; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[j]) : "v"(a), "v"(b));
    }
    long long t1 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 128; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(f[j]) : "v"(a), "v"(b));
    }
    long long t2 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 128; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(f[0]) : "v"(a), "v"(b));
    }
    long long t3 = __builtin_readcyclecounter();
    out[l] = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7];
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main()
{
    double *d; long long *c; (void)hipMalloc(&d, 4 * 64 * 8); (void)hipMalloc(&c, 32);
    layout<<<1, 64>>>(d);
    static double h[4 * 64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int ns[4] = {0, 1, 5, 15};
    for (int r = 0; r < 4; r++) { printf("row_newbcast:%-2d lane -> source lane:", ns[r]); for (int l = 0; l < 64; l++) printf(" %d", (int)h[r * 64 + l]); printf("\n"); }
    timing<<<1, 64>>>(d, c); timing<<<1, 64>>>(d, c);
    long long hc[3]; (void)hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost);
    printf("lone wavefront, 8 instructions per block (+ ~24 cycles of loop): v_fma_f64 independent %.1f, v_fmac_f64_dpp independent %.1f, v_fmac_f64_dpp dependent %.1f cycles per block\n",
           hc[0] / 128.0, hc[1] / 128.0, hc[2] / 128.0);
    return 0;
}
