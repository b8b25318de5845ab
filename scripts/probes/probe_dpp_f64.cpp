// Probe (round 6): semantics of the FP64 DPP forms the interior point kernel's lane-distributed micro-panels use on gfx950:
//   v_fmac_f64_dpp d, -s0, s1 row_newbcast:n   (negated broadcast operand)     v_fmac_f64_dpp d, -d, s1 (accumulator as its own broadcast source)
//   v_rcp_f64_dpp d, s0 row_newbcast:n
// build: hipcc --offload-arch=gfx950 -O2 -o probe_dpp_f64 probe_dpp_f64.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *in, double *out)
{
    const int l = threadIdx.x;
    double a = in[l], b = in[64 + l], c = in[128 + l];
    double acc = c;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b));
    out[l] = acc;                       // c - a[row*16+5] * b
    double self = c;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, -%0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(self) : "v"(b));
    out[64 + l] = self;                 // c - c[row*16+2] * b
    double r;
    asm volatile("s_nop 4\n\tv_rcp_f64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a));
    out[128 + l] = r;                   // ~ 1 / a[row*16+3]
    // back-to-back dependent: VALU write then DPP read WITHOUT wait states (what the hazard is about)
    double t = a * 2.0, u = c;
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(u) : "v"(t), "v"(b));
    out[192 + l] = u;                   // c - 2 a[row*16+1] * b  (if the hazard bites: stale t)
}
int main()
{
    double h[192], o[256], *di, *dout;
    for (int i = 0; i < 192; i++) h[i] = 1.0 + 0.37 * i + 0.001 * i * i;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    double e[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; l++) {
        const int row = l / 16;
        const double *a = h, *b = h + 64, *c = h + 128;
        e[0] = fmax(e[0], fabs(o[l] - fma(-a[row * 16 + 5], b[l], c[l])));
        e[1] = fmax(e[1], fabs(o[64 + l] - fma(-c[row * 16 + 2], b[l], c[l])));
        e[2] = fmax(e[2], fabs(o[128 + l] * a[row * 16 + 3] - 1.0));
        e[3] = fmax(e[3], fabs(o[192 + l] - fma(-2.0 * a[row * 16 + 1], b[l], c[l])));
    }
    printf("fmac_dpp -src0 bcast: max err %.3e | accumulator as broadcast source: %.3e | rcp_dpp: max |r a - 1| %.3e | no wait states after a VALU write: %.3e\n", e[0], e[1], e[2], e[3]);
    return 0;
}
