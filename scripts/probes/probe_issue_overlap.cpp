// probe_issue_overlap.cpp -- what a LONE wavefront (one per SIMD, ipm_kernel's occupancy) pays per instruction on gfx950, and what
// overlaps with an FP64 matrix instruction in flight: the question behind "15.6 k cycles of an interior point iteration carry no
// FP64" (DESIGN 4, K3). Every case is a loop of REP blocks of asm volatile instructions (the compiler keeps their order), timed with
// the shader clock; all operands independent of each other unless the case says "dependent".
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_issue_overlap scripts/probes/probe_issue_overlap.cpp && /tmp/probe_issue_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef double d4 __attribute__((ext_vector_type(4)));

#define FMA64(acc) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MUL64(acc) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc) : "v"(a))
#define ADD32(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(one))
#define CND32(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(one))
#define DPP32(r) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(one))
#define MFMA16(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA4(acc) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define DSR64(r) asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr))
#define DSW64(r) asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(r))
#define ACCW(ar, r) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ar) : "v"(r))
#define RCP64(acc) asm volatile("v_rcp_f64 %0, %1" : "=v"(acc) : "v"(a))
#define WAITL asm volatile("s_waitcnt lgkmcnt(0)")

constexpr int REP = 128;

#define REPEAT4(X) X X X X
#define REPEAT8(X) REPEAT4(X) REPEAT4(X)

// a case = MODE; returns cycles for REP blocks
template <int MODE>
__device__ long long run_case(double &sink, int &isink, double *lds)
{
    const int l = threadIdx.x & 63;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    double f0 = 0.1, f1 = 0.2, f2 = 0.3, f3 = 0.4, f4 = 0.5, f5 = 0.6, f6 = 0.7, f7 = 0.8;
    int r0 = l, r1 = l + 1, r2 = l + 2, r3 = l + 3, r4 = l + 4, r5 = l + 5, r6 = l + 6, r7 = l + 7, one = 1;
    d4 m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
    double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    double x0, x1, x2, x3;
    int ar0, ar1, ar2, ar3;
    unsigned addr = (unsigned)(size_t)lds + 8u * l;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(one), "+v"(addr));
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < REP; i++) {
        if constexpr (MODE == 0) { FMA64(f0); FMA64(f1); FMA64(f2); FMA64(f3); FMA64(f4); FMA64(f5); FMA64(f6); FMA64(f7); }                    // 8 independent fma64
        if constexpr (MODE == 1) { ADD32(r0); ADD32(r1); ADD32(r2); ADD32(r3); ADD32(r4); ADD32(r5); ADD32(r6); ADD32(r7); }                    // 8 add32
        if constexpr (MODE == 2) { FMA64(f0); ADD32(r0); FMA64(f1); ADD32(r1); FMA64(f2); ADD32(r2); FMA64(f3); ADD32(r3);
                                   FMA64(f4); ADD32(r4); FMA64(f5); ADD32(r5); FMA64(f6); ADD32(r6); FMA64(f7); ADD32(r7); }                    // 8 + 8 interleaved
        if constexpr (MODE == 3) { FMA64(f0); FMA64(f0); FMA64(f0); FMA64(f0); FMA64(f0); FMA64(f0); FMA64(f0); FMA64(f0); }                    // 8 dependent fma64
        if constexpr (MODE == 4) { MFMA16(m0); MFMA16(m1); }                                                                                    // 2 independent mfma 16x16x4
        if constexpr (MODE == 5) { MFMA16(m0); REPEAT8(ADD32(r0);) MFMA16(m1); REPEAT8(ADD32(r1);) }                                            // 2 x (mfma + 8 add32)
        if constexpr (MODE == 6) { MFMA16(m0); REPEAT8(ADD32(r0);) REPEAT4(ADD32(r2);) MFMA16(m1); REPEAT8(ADD32(r1);) REPEAT4(ADD32(r3);) }    // 2 x (mfma + 12 add32)
        if constexpr (MODE == 7) { MFMA16(m0); REPEAT4(FMA64(f0);) MFMA16(m1); REPEAT4(FMA64(f1);) }                                            // 2 x (mfma + 4 fma64 dependent among themselves)
        if constexpr (MODE == 8) { MFMA16(m0); FMA64(f0); FMA64(f1); FMA64(f2); FMA64(f3); MFMA16(m1); FMA64(f4); FMA64(f5); FMA64(f6); FMA64(f7); }   // 2 x (mfma + 4 independent fma64)
        if constexpr (MODE == 9) { MFMA4(q0); MFMA4(q1); MFMA4(q2); MFMA4(q3); }                                                                // 4 independent mfma 4x4x4
        if constexpr (MODE == 10) { MFMA4(q0); REPEAT4(ADD32(r0);) MFMA4(q1); REPEAT4(ADD32(r1);) MFMA4(q2); REPEAT4(ADD32(r2);) MFMA4(q3); REPEAT4(ADD32(r3);) }   // 4 x (mfma4 + 4 add32)
        if constexpr (MODE == 11) { MFMA16(m0); REPEAT8(CND32(r0);) MFMA16(m1); REPEAT8(CND32(r1);) }                                           // 2 x (mfma + 8 cndmask)
        if constexpr (MODE == 12) { MFMA16(m0); REPEAT8(DPP32(r0);) MFMA16(m1); REPEAT8(DPP32(r1);) }                                           // 2 x (mfma + 8 dpp mov)
        if constexpr (MODE == 13) { MFMA16(m0); DSR64(x0); DSR64(x1); DSR64(x2); DSR64(x3); MFMA16(m1); DSR64(x0); DSR64(x1); DSR64(x2); DSR64(x3); WAITL; }   // 2 x (mfma + 4 ds_read)
        if constexpr (MODE == 14) { DSR64(x0); DSR64(x1); DSR64(x2); DSR64(x3); DSR64(x0); DSR64(x1); DSR64(x2); DSR64(x3); WAITL; }            // 8 ds_read
        if constexpr (MODE == 15) { FMA64(f0); DSR64(x0); FMA64(f1); DSR64(x1); FMA64(f2); DSR64(x2); FMA64(f3); DSR64(x3);
                                    FMA64(f4); DSR64(x0); FMA64(f5); DSR64(x1); FMA64(f6); DSR64(x2); FMA64(f7); DSR64(x3); WAITL; }            // 8 fma64 + 8 ds_read interleaved
        if constexpr (MODE == 16) { MFMA16(m0); ACCW(ar0, r0); ACCW(ar1, r1); ACCW(ar2, r2); ACCW(ar3, r3); ACCW(ar0, r4); ACCW(ar1, r5); ACCW(ar2, r6); ACCW(ar3, r7);
                                    MFMA16(m1); ACCW(ar0, r0); ACCW(ar1, r1); ACCW(ar2, r2); ACCW(ar3, r3); ACCW(ar0, r4); ACCW(ar1, r5); ACCW(ar2, r6); ACCW(ar3, r7); }   // 2 x (mfma + 8 accvgpr_write)
        if constexpr (MODE == 17) { RCP64(x0); RCP64(x1); RCP64(x2); RCP64(x3); RCP64(x0); RCP64(x1); RCP64(x2); RCP64(x3); }                   // 8 rcp64
        if constexpr (MODE == 18) { MUL64(f0); MUL64(f1); MUL64(f2); MUL64(f3); MUL64(f4); MUL64(f5); MUL64(f6); MUL64(f7); }                    // 8 mul64
        if constexpr (MODE == 19) { MFMA16(m0); REPEAT8(ADD32(r0);) REPEAT8(ADD32(r2);) MFMA16(m1); REPEAT8(ADD32(r1);) REPEAT8(ADD32(r3);) }    // 2 x (mfma + 16 add32)
        if constexpr (MODE == 20) { MFMA16(m0); DSW64(f0); DSW64(f1); DSW64(f2); DSW64(f3); MFMA16(m1); DSW64(f4); DSW64(f5); DSW64(f6); DSW64(f7); WAITL; }   // 2 x (mfma + 4 ds_write)
        if constexpr (MODE == 21) { MFMA4(q0); FMA64(f0); FMA64(f1); MFMA4(q1); FMA64(f2); FMA64(f3); MFMA4(q2); FMA64(f4); FMA64(f5); MFMA4(q3); FMA64(f6); FMA64(f7); }   // 4 x (mfma4 + 2 fma64)
        if constexpr (MODE == 22) { FMA64(f0); ADD32(r0); ADD32(r1); FMA64(f1); ADD32(r2); ADD32(r3); FMA64(f2); ADD32(r4); ADD32(r5); FMA64(f3); ADD32(r6); ADD32(r7); }  // 4 x (fma64 + 2 add32)
        if constexpr (MODE == 23) { MFMA16(m0); MFMA16(m0); }                                                                                   // 2 dependent mfma
    }
    const long long t1 = __builtin_readcyclecounter();
    sink += f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + m0[0] + m0[1] + m0[2] + m0[3] + m1[0] + m1[1] + m1[2] + m1[3] + q0 + q1 + q2 + q3;
    if constexpr (MODE == 13 || MODE == 14 || MODE == 15 || MODE == 17) sink += x0 + x1 + x2 + x3;
    if constexpr (MODE == 16) { int t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(ar0)); isink += t + ar1 + ar2 + ar3; }
    isink += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    return t1 - t0;
}

constexpr int NCASE = 24;
static const char *NAMES[NCASE] = {
    "8 independent v_fma_f64", "8 v_add_u32", "8 fma64 + 8 add32 interleaved", "8 DEPENDENT v_fma_f64",
    "2 independent mfma_f64_16x16x4", "2 x (mfma16 + 8 add32)", "2 x (mfma16 + 12 add32)", "2 x (mfma16 + 4 dependent fma64)",
    "2 x (mfma16 + 4 independent fma64)", "4 independent mfma_f64_4x4x4", "4 x (mfma4 + 4 add32)", "2 x (mfma16 + 8 cndmask)",
    "2 x (mfma16 + 8 dpp mov)", "2 x (mfma16 + 4 ds_read_b64) + wait", "8 ds_read_b64 + wait", "8 fma64 + 8 ds_read_b64 interleaved + wait",
    "2 x (mfma16 + 8 accvgpr_write)", "8 v_rcp_f64", "8 v_mul_f64", "2 x (mfma16 + 16 add32)", "2 x (mfma16 + 4 ds_write_b64) + wait",
    "4 x (mfma4 + 2 fma64)", "4 x (fma64 + 2 add32)", "2 DEPENDENT mfma_f64_16x16x4"};

template <int M>
__device__ void all_cases(long long *cyc, double &sink, int &isink, double *lds)
{
    const long long c = run_case<M>(sink, isink, lds);
    if ((threadIdx.x & 63) == 0) cyc[(threadIdx.x >> 6) * NCASE + M] = c;
    if constexpr (M + 1 < NCASE) all_cases<M + 1>(cyc, sink, isink, lds);
}

__global__ void __launch_bounds__(512) probe(long long *cyc, double *out)
{
    __shared__ double lds[1024];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.0;
    __syncthreads();
    double sink = 0; int isink = 0;
    all_cases<0>(cyc, sink, isink, lds);
    out[threadIdx.x] = sink + isink;
}

int main()
{
    long long *c; double *o;
    hipMalloc(&c, 8 * NCASE * 8); hipMalloc(&o, 512 * 8);
    static long long h1[8 * NCASE], h2[8 * NCASE];
    probe<<<1, 64>>>(c, o); probe<<<1, 64>>>(c, o);
    hipMemcpy(h1, c, sizeof(h1), hipMemcpyDeviceToHost);
    probe<<<1, 512>>>(c, o); probe<<<1, 512>>>(c, o);          // eight wavefronts of one workgroup: two per SIMD
    hipMemcpy(h2, c, sizeof(h2), hipMemcpyDeviceToHost);
    printf("cycles per block (shader clock), %d blocks per case\n", REP);
    printf("%-48s %12s %22s\n", "block", "lone wave", "two waves per SIMD (max of 8)");
    for (int m = 0; m < NCASE; m++) {
        long long mx = 0;
        for (int w = 0; w < 8; w++) if (h2[w * NCASE + m] > mx) mx = h2[w * NCASE + m];
        printf("%-48s %12.1f %22.1f\n", NAMES[m], h1[m] / (double)REP, mx / (double)REP);
    }
    return 0;
}
