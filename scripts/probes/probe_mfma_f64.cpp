// Probe: v_mfma_f64_16x16x4_f64 operand/result layout and issue cost on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* D) {
    // A: 16x4 row-major, B: 4x16 row-major. lane l supplies A[l&15][l>>4], B[l>>4][l&15]
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; j++) D[l * 4 + j] = c[j];
}
__global__ void k_time(double* out, long long* cyc, int iters) {
    int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 2.0 - l * 1e-3;
    d4 c0 = {0,0,0,0}, c1 = {0,0,0,0}, c2 = {0,0,0,0}, c3 = {0,0,0,0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_time_dep(double* out, long long* cyc, int iters) {
    int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 2.0 - l * 1e-3;
    d4 c0 = {0,0,0,0};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = c0[0];
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_time_fma(double* out, long long* cyc, int iters) {
    int l = threadIdx.x;
    double a = 1.0 + l * 1e-9, b = 1e-9 * l;
    double c[8]; for (int j = 0; j < 8; j++) c[j] = j;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = __builtin_fma(c[j], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int j = 0; j < 8; j++) s += c[j];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i * 0.37 + k * 1.91;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 0.5 + k * 2.3 - j * 0.11 + (k * j) * 0.013;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD; long long* dC;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 64 * 4096 * 8); hipMalloc(&dC, 4096 * 8);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    // check documented layout: lane l, reg j -> row (l>>4)+4j, col l&15
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
        int row = (l >> 4) + 4 * j, col = l & 15;
        if (fabs(D[l * 4 + j] - ref[row * 16 + col]) > 1e-9) bad++;
    }
    printf("layout row=(l>>4)+4j col=l&15 : %s (bad=%d)\n", bad ? "MISMATCH" : "OK", bad);
    if (bad) { // search
        for (int l = 0; l < 8; l++) for (int j = 0; j < 4; j++) for (int e = 0; e < 256; e++) if (fabs(D[l*4+j]-ref[e])<1e-9) printf("lane %d reg %d -> row %d col %d\n", l, j, e/16, e%16);
    }
    int iters = 2000;
    std::vector<long long> cyc(4096);
    for (int nb : {1, 256, 1024, 2048}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_time<<<nb, 64>>>(dD, dC, iters); hipDeviceSynchronize();
        hipEventRecord(e0); k_time<<<nb, 64>>>(dD, dC, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(cyc.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("mfma indep x4: blocks %d  cyc/mfma(lane0 wave0) %.1f  wall %.3f ms -> %.2f TFLOP/s\n", nb, cyc[0] / (4.0 * iters), ms, nb * 4.0 * iters * 2048 / ms / 1e9);
        k_time_dep<<<nb, 64>>>(dD, dC, iters); hipDeviceSynchronize();
        hipEventRecord(e0); k_time_dep<<<nb, 64>>>(dD, dC, iters); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(cyc.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("mfma dependent: blocks %d  cyc/mfma %.1f  wall %.3f ms\n", nb, cyc[0] / (4.0 * iters), ms);
        k_time_fma<<<nb, 64>>>(dD, dC, iters); hipDeviceSynchronize();
        hipEventRecord(e0); k_time_fma<<<nb, 64>>>(dD, dC, iters); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(cyc.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        printf("dfma x8 indep: blocks %d  cyc/dfma %.2f  wall %.3f ms -> %.2f TFLOP/s\n", nb, cyc[0] / (8.0 * iters), ms, nb * 8.0 * iters * 128 / ms / 1e9);
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s CUs %d clock %d kHz LDS/block %zu\n", p.name, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
    return 0;
}
