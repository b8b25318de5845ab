// probe: gfx950 v_permlane16_swap / v_permlane32_swap semantics and a 4-row (lanes l, l^16, l^32, l^48) sum of doubles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ inline double quad_sum(double v)
{
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    u32x2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    u32x2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    double s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(s); hi = __double2hiint(s);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ inline long long now(double dep)
{
    int tmp;
    asm volatile("v_readfirstlane_b32 %0, %1\n s_nop 4" : "=s"(tmp) : "v"(__double2hiint(dep)) : "memory");   // wait for dep
    return __builtin_readcyclecounter();
}
__global__ void k(double *out, long long *cyc)
{
    const int l = threadIdx.x;
    double v = 1.0 + l;
    long long t0 = now(v);
    double s = quad_sum(v + (double)(t0 >> 62));
#pragma unroll
    for (int i = 0; i < 15; i++) s = quad_sum(s * 0.25);
    long long t1 = now(s);
    out[l] = quad_sum(v);
    out[64 + l] = s;
    // gather within a row: lane (q,c) takes the value of lane (q, (q + 4*jj) & 15)
    int src = (l & 48) + (((l >> 4) + 4 * 1) & 15);
    double g = v;
    long long t2 = now(g);
    g += (double)(t2 >> 62);                 // the chain starts after the time stamp
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int glo = __builtin_amdgcn_ds_bpermute(src * 4, __double2loint(g));
        int ghi = __builtin_amdgcn_ds_bpermute(src * 4, __double2hiint(g));
        g = __hiloint2double(ghi, glo) + 1.0;
    }
    long long t3 = now(g);
    out[128 + l] = g;
    // calibration: 1024 dependent v_fma_f64 (7.5 cycles each for a lone wavefront, profiles/r01_probe_mfma_f64.txt)
    double f = v;
    long long t4 = now(f);
    f += (double)(t4 >> 62);
#pragma unroll 16
    for (int i = 0; i < 1024; i++) f = __builtin_fma(f, 0.999, 0.001);
    long long t5 = now(f);
    out[128 + l] += f * 1e-300;
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; cyc[2] = t5 - t4; }
}
int main()
{
    double *d; long long *c; hipMalloc(&d, 192 * 8); hipMalloc(&c, 32);
    k<<<1, 64>>>(d, c);
    double h[192]; long long hc[3];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; l++) { double want = 4.0 + (l & 15) * 4 + (0 + 16 + 32 + 48); if (h[l] != want) ok = false; }
    printf("quad_sum correct: %d (lane 5: %g)\n", ok, h[5]);
    printf("16 dependent quad_sums: %lld cycles (%.1f each); 16 dependent double bpermute gathers: %lld cycles (%.1f each)\n",
           hc[0], hc[0] / 16.0, hc[1], hc[1] / 16.0);
    printf("1024 dependent fma: %lld ticks (%.3f each) -> one tick = %.1f fma-cycles at 7.5 cycles per fma\n", hc[2], hc[2] / 1024.0, 7.5 * 1024.0 / hc[2]);
    return 0;
}
