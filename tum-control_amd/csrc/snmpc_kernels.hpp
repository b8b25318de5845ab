// snmpc_kernels.hpp -- the coupled SNMPC OCP around the SQP-RTI pipeline of pipe_kernels.hpp.
//
// Reference: Stochastic_NMPC/pred_model_dynamic_disc.py:121-220 (DISCRETE dynamics of the stacked state: nominal copy +
// n_s sample copies under ONE input), Stochastic_NMPC/SNMPC_acados_settings.py:57-194 (cost on the nominal copy, chance
// constraint E + kappa sqrt(Var) over the samples), Stochastic_NMPC/SNMPC_class.py:103-104,179-198 (stop flags of the
// uncertainty propagation horizon `uph`, solve).
//
// Structure that this build exploits instead of condensing an 88-state system:
//   * the stage matrix is block diagonal over the sample copies; the nominal copy of stage s <= uph is the PCE mean of
//     the sample copies (row 0 of A_pce), so  G_nom,s = sum_i a_i G^(i)_s  and the full-condensed QP keeps the shape
//     of the nominal one (2N variables, the same 3 soft rows per stage);
//   * from stage uph on the samples are frozen and never read again: the nominal recursion of the condensing kernel takes
//     over from G_nom,uph;
//   * at stage s the sample matrices G^(i)_s only have 2s <= 2 uph non-zero columns.
// So the solve is these launches on one stream (shipped library: the pipeline; the development build can put round 1's fused
// kernel nmpc_rti_kernel<., true> in the place of the four pipeline kernels):
//   snmpc_lin_kernel             sample stages: one RK4 step with sensitivities and the gg value / gradient per (instance, stage,
//                                sample) item, lane = item (full wavefronts; the register-heavy part, one wavefront per SIMD)
//   snmpc_prologue_kernel<NPM>   one wavefront per OCP: PCE weights of the chance rows, column recursions of all samples
//   / snmpc_prologue_cols_kernel (lane = (sample, column slot) with passes / lane = column, round 4); hands G_nom,s, g_nom,s and the
//                                chance-constraint rows of the stages 1..uph to the condensing kernel through `pro`
//   lin_kernel<true>, cond_kernel<., true>, ipm_kernel   cost rows / gg rows / Hessian of stages <= uph from `pro`, nominal recursion
//                                from stage uph, interior point
//   snmpc_epilogue_kernel        full step of the sample copies (lane = sample) and, as their PCE mean, of the nominal copy of the
//                                stages 1..uph
//   expand_kernel<., true>       the nominal recursion from stage uph to the end of the horizon, cost at the new iterate
#pragma once
#include "nmpc_device.hpp"

namespace tum {

constexpr int SN_NSMAX = 32;                 // max samples
constexpr int SN_LMAX = 32;                  // max PCE terms
// Compile-time bound of the short sums over samples / PCE terms and of the LDS tables sized by them: 16 wherever n_samples and the number of PCE
// terms are at most 16 (the shipped 10 x 10 and everything rounds 1-5 covered: unchanged kernels), 32 beyond (round 6: the column-slot prologue with its
// column state in LDS and the epilogue have a second instantiation; the matrix-core prologue takes at most ten samples anyway)
constexpr int SN_B16 = 16;
constexpr int SN_UPHMAX = 48;                // up to the whole horizon (the reference ran UPH = Tp, SNMPC_class.py:103-104)
constexpr int SN_UPHMAX_FUSED = 31;          // the fused kernel reads `pro` with a fixed pitch of 64 columns
// `pro` holds, per stage s <= uph, nine rows (8 rows of G_nom,s | the chance-constraint row) of 2 uph + 1 columns (the sample
// columns 0..2 uph-1 and the constant column g at index 2 uph). Row pitch: 64 doubles while the columns fit one wavefront
// (uph <= 31), 128 beyond (the consumers then fill their second register bank from the columns 64..).
__host__ __device__ inline int sn_pro_pitch(int uph) { return (2 * uph + 1 <= 64) ? 64 : 128; }
__host__ __device__ inline int sn_pro_stage(int uph) { return 9 * sn_pro_pitch(uph); }

struct SnArgs {
    int N, batch, ns, L, uph;
    double dt, kappa;
    Model mp;
    const double *X;          // [b][(N+1)*8]   nominal copy of the iterate
    const double *U;          // [b][N*2]
    double *XS;               // [b][N+1][ns][8] sample copies of the iterate
    const double *xs0;        // [b][ns][8]      initial condition of the samples (lbx_0 = ubx_0)
    const double *Apce;       // [L][ns]         PCE matrix (SNMPC_class.py:124), shared by the batch
    double *ws2;              // [b][uph*ns][ABS] sample linearisation records
    double *gh;               // [b][uph*ns][5]   gg value and its gradient (vl, vt, r, a) per (stage, sample) item
    double *pro;              // [b][uph][sn_pro_stage(uph)]
    const double *dv;         // [b][dv_stride]  QP solution of the fused kernel / of the pipeline's interior point kernel (epilogue)
    int dv_stride;
    const int *status;        // [b]
    int *xs_dirty;            // [b] 1: the sample copies of the stages > uph wait to be frozen at the value of stage uph (snmpc_freeze_kernel)
    double *Xn;               // pipeline only (else null): the epilogue also writes the NOMINAL copy of the stages 1..uph, [b][(N+1)*8]
    double *dxu;              // ... and the nominal step of stage uph, 8 doubles per instance at stride dv_stride (the expansion kernel starts from it)
    double *dbg;              // development aid: phase cycle counters of instance 0 (or null)
};

// gg circle with the limits looked up at |v| (SNMPC_acados_settings.py:60-67,100-113): value and d/d(vl, vt, r, a)
__device__ __forceinline__ void h_con_vabs(const Model &p, double vl, double vt, double r, double a,
                                           double &h, double &g3, double &g4, double &g5, double &g7)
{
    const double vabs = sqrt(vl * vl + vt * vt);
    double ax, dax, ay, day;
    interp_lin(p.n_ggv, p.ggv_v, p.ggv_ax, vabs, ax, dax);
    interp_lin(p.n_ggv, p.ggv_v, p.ggv_ay, vabs, ay, day);
    if (a < 0.0) { ax = p.ax_brake; dax = 0.0; }
    const double alat = vl * r, nlon = a / ax, nlat = alat / ay;
    h = nlon * nlon + nlat * nlat;
    const double dvv = -2.0 * nlat * alat / (ay * ay) * day - 2.0 * nlon * a / (ax * ax) * dax;
    const double iv = (vabs > 0.0) ? 1.0 / vabs : 0.0;
    g3 = 2.0 * nlat * r / ay + dvv * vl * iv;
    g4 = dvv * vt * iv;
    g5 = 2.0 * nlat * vl / ay;
    g7 = 2.0 * nlon / ax;
}

// dynamic LDS of the prologue (doubles): per (stage, sample) item the gg value, its 4 gradient entries and the PCE weight;
// the constraint value per stage; the records of one stage; the reduction buffer of P3; the nominal copy's own defect per
// stage; the column state of P3 (8 doubles per lane and pass); A_pce and the PCE coefficients per stage
__host__ __device__ inline int sn_prologue_passes(int uph, int ns) { const int cs = 64 / ns; return (2 * uph + 1 + cs - 1) / cs; }
// (NPM = 0: the column state of P3 waits in LDS between the stages -- 4 KiB per pass; NPM > 0: it lives in registers, see below)
__host__ __device__ inline int sn_prologue_lds_doubles(int uph, int ns, int npm = 0, int nsb = SN_B16)
{
    // (NPM > 0: reduction rows of pitch 128 whose upper halves stay zero: the sum over the samples reads 16 cells of a row without a mask)
    return (npm > 0 ? 2 : 6) * uph * ns + (uph + 1) + ns * ABS + (npm > 0 ? 9 * 130 : 9 * 64) + uph * 8 + (npm > 0 ? 0 : sn_prologue_passes(uph, ns) * 8 * 64) +
           nsb * nsb + uph * nsb;
}
// register-resident variants of the prologue: the number of passes NPM a lane's column state is held for (8 doubles each).
// Chosen on the host: the smallest instantiation that covers the passes of the last stage; 0 (column state in LDS) for short
// propagation horizons, where the LDS variant's five wavefronts per SIMD win, and beyond the largest instantiation.
__host__ inline int sn_prologue_variant(int uph, int ns)
{
    if (ns > SN_B16) return 0;          // (more than 16 samples: the LDS variant, instantiated with the 32-wide bounds)
    const int np = sn_prologue_passes(uph, ns);
    static const int forced = [] { const char *e = getenv("TUM_SN_PROLOGUE"); return e ? atoi(e) : -1; }();     // development aid
    if (forced == 0 || ((forced == 6 || forced == 9 || forced == 13 || forced == 17) && np <= forced && ns >= 8)) return forced;
    if (np <= 3 || ns < 8) return 0;         // (fewer than 8 samples: more than 8 column slots per pass, the LDS variant's general reduction)
    for (int v : {6, 9, 13, 17}) if (np <= v) return v;          // (beyond 17 passes the LDS variant runs)
    return 0;
}

// Which prologue runs: the matrix-core kernel (snmpc_prologue_mfma_kernel below) wherever the sample count allows (two wavefronts
// with SN_MFMA_NSW samples each: n_samples <= 10), the column-slot / pass variants above for more samples -- and as the second
// implementation the tests hold the new one against (per capsule: tum_ocp_set_kernel "prologue-passes" / "prologue-mfma";
// TUM_SN_PROLOGUE = 0 | 6 | 9 | 13 | 17 forces a pass variant, "mfma" the matrix-core kernel). Measured on 4096 instances, whole
// solve, matrix cores against passes: uph = 5: 1.31 / 1.31 ms, 9: 1.46 / 1.48, 15: 1.70 / 1.77, 24: 1.99 / 2.16, 38: 2.54 / 2.88.
constexpr int SN_MFMA_NSW = 5, SN_MFMA_NS = 2 * SN_MFMA_NSW;
// 0: column slots and passes, 2: the matrix-core kernel. `want` < 0: the library's choice.
__host__ inline int sn_prologue_kind(int uph, int ns, int want = -1)
{
    static const int forced = [] { const char *e = getenv("TUM_SN_PROLOGUE"); return !e ? -1 : (e[0] == 'm' ? 2 : 0); }();
    const int f = want >= 0 ? want : forced;
    (void)uph;
    if (ns > SN_MFMA_NS || f == 0) return 0;
    return 2;
}

// K-S1: linearisation of the sample stages. lane = (instance, stage k < uph, sample) item; the 64 records of a wavefront are
// one contiguous block of ws2 and go out transposed through LDS (416 contiguous bytes per store instruction instead of 64
// scattered 8-byte stores), like lin_kernel's.
__global__ void __launch_bounds__(64, 1) snmpc_lin_kernel(const SnArgs sa)
{
    __shared__ double sT[64 * ABS];
    const int N = sa.N, ns = sa.ns, nitem = sa.uph * ns;
    const long long total = (long long)sa.batch * nitem;
    const long long g0 = (long long)blockIdx.x * 64, gl = g0 + threadIdx.x;
    const bool live = gl < total;
    const long long g = live ? gl : total - 1;          // (lanes beyond the last item shadow it and store nothing)
    const int b = (int)(g / nitem), item = (int)(g - (long long)b * nitem);
    const int k = item / ns, i = item - k * ns;
    const double *gXS = sa.XS + (size_t)b * (N + 1) * ns * NX;
    const double *gU = sa.U + (size_t)b * N * NU;
    const double *xp = gXS + ((size_t)k * ns + i) * NX;
    double xk[8], uk[2] = {gU[2 * k], gU[2 * k + 1]};
#pragma unroll
    for (int r = 0; r < 8; r++) xk[r] = xp[r];
    double xn[8], Sp[2], S[6][7];
    rk4_sens(sa.mp, xk, uk, sa.dt, 1, xn, Sp, S);
    double *rec = sT + threadIdx.x * ABS;
    rec[0] = Sp[0]; rec[1] = Sp[1];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 7; c++) rec[2 + r * 7 + c] = S[r][c];
    const double *xq = gXS + ((size_t)(k + 1) * ns + i) * NX;
#pragma unroll
    for (int r = 0; r < 8; r++) rec[44 + r] = xn[r] - xq[r];
    double h = 0.0, g3 = 0.0, g4 = 0.0, g5 = 0.0, g7 = 0.0;
    if (k >= 1) h_con_vabs(sa.mp, xk[3], xk[4], xk[5], xk[7], h, g3, g4, g5, g7);
    if (live) {
        double *gh = sa.gh + (size_t)g * 5;
        gh[0] = h; gh[1] = g3; gh[2] = g4; gh[3] = g5; gh[4] = g7;
    }
    wsync();
    double *dst = sa.ws2 + (size_t)g0 * ABS;
    const int nit = (int)((total - g0 < 64) ? (total - g0) : 64);
    for (int it = 0; it < nit; it++)
        if ((int)threadIdx.x < 52) dst[(size_t)it * ABS + threadIdx.x] = sT[it * ABS + threadIdx.x];
}

// K-S1 for small batches: eight lanes per (instance, stage, sample) item -- one sensitivity column per lane, tyre chains split over
// each DPP quad (rk4_sens_col, nmpc_device.hpp; lin_cols_kernel is the nominal counterpart). The state step is bit-identical to
// snmpc_lin_kernel's, the sensitivities agree to 3e-15 relative (FMA contraction, see lin_cols_kernel).
constexpr int SLC_LANES = 8, SLC_ITEMS = 64 / SLC_LANES;
__global__ void __launch_bounds__(64, 1) snmpc_lin_cols_kernel(const SnArgs sa)
{
    __shared__ double sT[SLC_ITEMS * ABS];
    const int N = sa.N, ns = sa.ns, nitem = sa.uph * ns;
    const long long total = (long long)sa.batch * nitem;
    const int li = threadIdx.x / SLC_LANES, col = threadIdx.x % SLC_LANES;
    const long long g0 = (long long)blockIdx.x * SLC_ITEMS, gl = g0 + li;
    const bool live = gl < total;
    const long long g = live ? gl : total - 1;          // (groups beyond the last item shadow it and store nothing)
    const int b = (int)(g / nitem), item = (int)(g - (long long)b * nitem);
    const int k = item / ns, i = item - k * ns;
    const double *gXS = sa.XS + (size_t)b * (N + 1) * ns * NX;
    const double *gU = sa.U + (size_t)b * N * NU;
    const double *xp = gXS + ((size_t)k * ns + i) * NX;
    double xk[8], uk[2] = {gU[2 * k], gU[2 * k + 1]};
#pragma unroll
    for (int r = 0; r < 8; r++) xk[r] = xp[r];
    const TyreLane t = tyre_lane(sa.mp, col);
    double xn[8], Sc[6];
    rk4_sens_col(sa.mp, t, col, xk, uk, sa.dt, 1, xn, Sc);
    double *rec = sT + li * ABS;
    if (col == 7) { rec[0] = Sc[0]; rec[1] = Sc[1]; }
    else {
#pragma unroll
        for (int r = 0; r < 6; r++) rec[2 + r * 7 + col] = Sc[r];
    }
    const double *xq = gXS + ((size_t)(k + 1) * ns + i) * NX;
    double dn = xn[0] - xq[0];
#pragma unroll
    for (int r = 1; r < 8; r++) dn = (col == r) ? xn[r] - xq[r] : dn;
    rec[44 + col] = dn;
    double h = 0.0, g3 = 0.0, g4 = 0.0, g5 = 0.0, g7 = 0.0;
    if (k >= 1) h_con_vabs(sa.mp, xk[3], xk[4], xk[5], xk[7], h, g3, g4, g5, g7);
    if (live && col < 5) {
        const double v = (col == 0) ? h : (col == 1) ? g3 : (col == 2) ? g4 : (col == 3) ? g5 : g7;
        sa.gh[(size_t)g * 5 + col] = v;
    }
    wsync();
    double *dst = sa.ws2 + (size_t)g0 * ABS;
    const int nit = (int)((total - g0 < SLC_ITEMS) ? (total - g0) : SLC_ITEMS);
    for (int it = 0; it < nit; it++)
        if ((int)threadIdx.x < 52) dst[(size_t)it * ABS + threadIdx.x] = sT[it * ABS + threadIdx.x];
}

template <int NPM, int NSB = SN_B16>
__global__ void __launch_bounds__(64) snmpc_prologue_kernel(const SnArgs sa)
{
    static_assert(NSB == SN_B16 || (NSB == SN_NSMAX && NPM == 0), "32-wide bounds exist for the LDS variant only");
    extern __shared__ __attribute__((aligned(16))) double sn_lds[];
    const int lane = threadIdx.x, b = blockIdx.x;
    if (b >= sa.batch) return;
    const int N = sa.N, ns = sa.ns, L = sa.L, uph = sa.uph;
    const int nitem = uph * ns;
    // (NPM > 0: the gradients of the gg value are read from the workspace where P3 needs them -- four doubles per lane and stage --
    //  instead of waiting in LDS: with them the kernel needs 41.5 KiB at uph = 38 and only three instances share a CU)
    double *sH = sn_lds, *sGh = sH + nitem, *sCoef = sGh + (NPM > 0 ? 0 : 4 * nitem), *sHval = sCoef + nitem;
    constexpr int RP = (NPM > 0) ? 130 : 64;          // pitch of a reduction row (130: 128 cells, and the nine rows a reducing instruction reads do not start in the same bank)
    double *sRec = sHval + (uph + 1), *sRed = sRec + ns * ABS, *sDef = sRed + 9 * RP, *sW = sDef + uph * 8;
    double *sA = sW + (NPM > 0 ? 0 : sn_prologue_passes(uph, ns) * 8 * 64), *sC = sA + NSB * NSB;
    const double dt = sa.dt;
    const double *gX = sa.X + (size_t)b * (N + 1) * NX;
    const double *gXS = sa.XS + (size_t)b * (N + 1) * ns * NX;
    double *ws2 = sa.ws2 + (size_t)b * uph * ns * ABS;
    const int PP = sn_pro_pitch(uph), PSTAGE = 9 * PP;
    double *pro = sa.pro + (size_t)b * uph * PSTAGE;

    const long long t0 = __builtin_readcyclecounter();
    // ---- P1 (snmpc_lin_kernel, launched before this kernel): records in ws2, gg values and gradients in gh
    {
        const double *gh = sa.gh + (size_t)b * nitem * 5;
        for (int o = lane; o < nitem * 5; o += 64) {
            const int item = o / 5, c = o - item * 5;
            const double v = gh[o];
            if (c == 0) sH[item] = v; else if (NPM == 0) sGh[item * 4 + c - 1] = v;
        }
    }
    if constexpr (NPM > 0) {
#pragma unroll
        for (int r = 0; r < 9; r++) sRed[r * RP + 64 + lane] = 0.0;
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();

    // ---- P2: PCE coefficients c = A h of the sample values per stage, weights d(E + kappa sqrt(Var)) / d h_i (stages
    // 1..uph-1). A_pce sits in LDS; the short sums have compile-time bounds so that their operands are in flight together.
    for (int o = lane; o < L * ns; o += 64) sA[o] = sa.Apce[o];
    __syncthreads();
    for (int o = lane; o < uph * L; o += 64) {          // c[k][l]
        const int k = o / L, l = o - k * L;
        double cl = 0.0;
#pragma unroll
        for (int j = 0; j < NSB; j++) cl += (j < ns) ? sA[l * ns + j] * sH[k * ns + j] : 0.0;
        sC[o] = cl;
    }
    __syncthreads();
    for (int item = lane; item < nitem; item += 64) {
        const int k = item / ns, i = item - k * ns;
        double w = 0.0;
        if (k >= 1) {
            double var = 0.0, acc = 0.0;
#pragma unroll
            for (int l = 1; l < NSB; l++) {
                const double cl = (l < L) ? sC[k * L + l] : 0.0;
                var += cl * cl; acc += (l < L) ? cl * sA[l * ns + i] : 0.0;
            }
            const double sd = sqrt(var);
            w = sA[i] + ((sd > 0.0) ? sa.kappa * acc / sd : 0.0);
            if (i == 0) sHval[k] = sC[k * L] + sa.kappa * sd;
        }
        sCoef[item] = w;
    }

    // the nominal copy's own defect: sum_i a_i X^(i)_s - X_nom,s
    for (int o = lane; o < uph * 8; o += 64) {
        const int s = (o >> 3) + 1, r = o & 7;
        double acc = -gX[s * NX + r];
#pragma unroll
        for (int ii = 0; ii < NSB; ii++) acc += (ii < ns) ? sA[ii] * gXS[((size_t)s * ns + ii) * NX + r] : 0.0;
        sDef[o] = acc;
    }

    const long long t2 = __builtin_readcyclecounter();
    // ---- P3: column recursions G^(i)_{k+1} = A^(i)_k G^(i)_k + B^(i)_k e_k of all samples. lane = (sample i, column slot c):
    // floor(64 / ns) columns of every sample advance together; slot 0 is the constant column g, slot q >= 1 column q-1 of G.
    // Stage k only has 2(k+1) live columns, so it takes ceil((2k+3) / slots) passes; the column state of the passes waits in
    // LDS. The PCE mean over the samples (G_nom) and the weighted sum of the chance-constraint row are reductions over the
    // sample lanes through LDS. The records of a stage are contiguous in ws2 and staged through LDS one stage ahead.
    // (wsync, not __syncthreads: one wavefront per workgroup, and a barrier would drain the outstanding stores to `pro`
    // and the record prefetch in every pass)
    const int CS = 64 / ns;
    const int si = lane / CS, sc = lane - si * CS;
    const bool act = si < ns;
    const int i = act ? si : 0;
    const int nrec = ns * ABS;
    constexpr int NCH = (NSB * ABS + 63) / 64;
    const double ai = sA[i];
    const int npass = sn_prologue_passes(uph, ns);
    // NPM > 0: the column state of this lane's slot in every pass stays in registers (8 NPM doubles) and the record of the lane's
    // sample is read from LDS once per STAGE instead of once per pass (52 of the ~90 LDS instructions of a pass), the reduction
    // over the samples reads its operands together (compile-time bound). The LDS variant needs 4 KiB per pass -- 65 KiB at
    // uph = 38, two wavefronts per CU -- and was 6.2 of the 8.2 ms of a 4096-instance solve there.
    double W[NPM > 0 ? NPM : 1][8];
    if constexpr (NPM > 0) {
#pragma unroll
        for (int pass = 0; pass < NPM; pass++) {
            const bool isg = act && pass == 0 && sc == 0;
#pragma unroll
            for (int r = 0; r < 8; r++) W[pass][r] = isg ? sa.xs0[((size_t)b * ns + i) * NX + r] - gXS[(size_t)i * NX + r] : 0.0;
        }
    } else {
    for (int pass = 0; pass < npass; pass++) {
        const bool isg = act && pass == 0 && sc == 0;
#pragma unroll
        for (int r = 0; r < 8; r++)
            sW[(pass * 8 + r) * 64 + lane] = isg ? sa.xs0[((size_t)b * ns + i) * NX + r] - gXS[(size_t)i * NX + r] : 0.0;
    }
    }
    double pre[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; pre[c] = (uph > 0 && idx < nrec) ? ws2[idx] : 0.0; }
    // (NPM > 0: the gg gradients of the lane's sample at stage s, requested one stage ahead like the records)
    double ghn[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (NPM > 0) {
        if (1 < uph) {
#pragma unroll
            for (int e = 0; e < 4; e++) ghn[e] = sa.gh[((size_t)b * nitem + 1 * ns + i) * 5 + 1 + e];
        }
    }
    for (int k = 0; k < uph; k++) {
        const int s = k + 1;
        wsync();
#pragma unroll
        for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; if (idx < nrec) sRec[idx] = pre[c]; }
        wsync();
        if (k + 1 < uph) {
#pragma unroll
            for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; pre[c] = (idx < nrec) ? ws2[(size_t)(k + 1) * nrec + idx] : 0.0; }
        }
        double *pg = pro + (size_t)k * PSTAGE;
        const int np_k = (2 * k + 3 + CS - 1) / CS;
        if constexpr (NPM > 0) {
            // (the slot of the lane is derived again in every stage from a copy the optimiser cannot see through: the per-pass lane
            //  predicates would otherwise all be hoisted out of the stage loop and held in scalar registers -- 137 of them spilled)
            int lane_k = lane;
            asm volatile("" : "+v"(lane_k));
            const int si = lane_k / CS, sc = lane_k - si * CS;
            const bool act = si < ns;
            const int i = act ? si : 0;
            double rec[52];
#pragma unroll
            for (int f = 0; f < 52; f++) rec[f] = sRec[i * ABS + f];
            double gh4[4], coefk = 0.0;                        // chance row of this stage's item (sample i): weight and gradient
#pragma unroll
            for (int e = 0; e < 4; e++) gh4[e] = ghn[e];
            if (s < uph) coefk = sCoef[s * ns + i];
            if (s + 1 < uph) {
#pragma unroll
                for (int e = 0; e < 4; e++) ghn[e] = sa.gh[((size_t)b * nitem + (s + 1) * ns + i) * 5 + 1 + e];
            }
#pragma unroll
            for (int pass = 0; pass < NPM; pass++) {
                if (pass < np_k) {
                    asm volatile("" ::: "memory");          // (keeps this a BRANCH: without it the compiler predicates all NPM passes into one block and
                                                             //  every stage pays for the passes its columns do not reach yet)
                    const int q = pass * CS + sc;
                    const bool isg = act && q == 0, valid = act && q <= 2 * uph;
                    const int col = q - 1, jst = col >> 1, r0 = col & 1;
                    double *w = W[pass];
                    const double sel = (valid && !isg && jst == k) ? 1.0 : 0.0, selg = isg ? 1.0 : 0.0;
                    apply_A(rec, w);
#pragma unroll
                    for (int r = 0; r < 6; r++) w[r] += sel * (r0 ? rec[2 + r * 7 + 6] : rec[2 + r * 7 + 5]);      // (a lane-dependent index would send the record to scratch)
                    w[6] += sel * (r0 ? dt : 0.0);
                    w[7] += sel * (r0 ? 0.0 : dt);
#pragma unroll
                    for (int r = 0; r < 8; r++) w[r] += selg * rec[44 + r];
#pragma unroll
                    for (int r = 0; r < 8; r++) sRed[r * RP + lane] = valid ? ai * w[r] : 0.0;
                    const double rowv = coefk * (gh4[0] * w[3] + gh4[1] * w[4] + gh4[2] * w[5] + gh4[3] * w[7]);
                    sRed[8 * RP + lane] = (valid && s < uph) ? rowv : 0.0;
                    wsync();
                    // lane (si, sc) with si < 9 sums row si, column slot sc over the samples: cells sc, CS + sc, ... (CS <= 8 here: 16 cells
                    // stay inside the 128-wide row; those of lanes without a sample and the upper half hold zeros). With eight samples
                    // (CS = 8) the lanes only reach the rows 0..7: the lanes of row 0 then sum the chance row (row 8) in a second round.
                    auto reduce_row = [&](const int row, const int cc) {
                        const double *rp = sRed + row * RP + cc;
                        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                        for (int ii = 0; ii < SN_B16; ii += 4) { a0 += rp[ii * CS]; a1 += rp[(ii + 1) * CS]; a2 += rp[(ii + 2) * CS]; a3 += rp[(ii + 3) * CS]; }
                        const double acc = (a0 + a1) + (a2 + a3);
                        const int qo = pass * CS + cc;
                        if (qo <= 2 * uph) {
                            const bool og = (qo == 0);
                            const int colo = og ? 2 * uph : qo - 1;
                            if (row < 8) pg[row * PP + colo] = acc + (og ? sDef[k * 8 + row] : 0.0);
                            else pg[8 * PP + colo] = acc + ((og && s < uph) ? sHval[s] : 0.0);
                        }
                    };
                    if (si < 9) reduce_row(si, sc);
                    if (9 * CS > 64 && si == 0) reduce_row(8, sc);
                    wsync();
                }
            }
        } else {
        const double *rec = sRec + i * ABS;
        for (int pass = 0; pass < np_k; pass++) {
            const int q = pass * CS + sc;
            const bool isg = act && q == 0, valid = act && q <= 2 * uph;
            const int col = q - 1, jst = col >> 1, r0 = col & 1;
            double w[8];
#pragma unroll
            for (int r = 0; r < 8; r++) w[r] = sW[(pass * 8 + r) * 64 + lane];
            const double sel = (valid && !isg && jst == k) ? 1.0 : 0.0, selg = isg ? 1.0 : 0.0;
            apply_A(rec, w);
#pragma unroll
            for (int r = 0; r < 6; r++) w[r] += sel * rec[2 + r * 7 + 5 + r0];
            w[6] += sel * (r0 ? dt : 0.0);
            w[7] += sel * (r0 ? 0.0 : dt);
#pragma unroll
            for (int r = 0; r < 8; r++) w[r] += selg * rec[44 + r];
#pragma unroll
            for (int r = 0; r < 8; r++) { sW[(pass * 8 + r) * 64 + lane] = w[r]; sRed[r * 64 + lane] = valid ? ai * w[r] : 0.0; }
            double rowv = 0.0;
            if (s < uph) {
                const int it = s * ns + i;
                rowv = sCoef[it] * (sGh[it * 4 + 0] * w[3] + sGh[it * 4 + 1] * w[4] + sGh[it * 4 + 2] * w[5] + sGh[it * 4 + 3] * w[7]);
            }
            sRed[8 * 64 + lane] = valid ? rowv : 0.0;
            wsync();
            for (int o = lane; o < 9 * CS; o += 64) {
                const int r = o / CS, cc = o - r * CS, qo = pass * CS + cc;
                double acc = 0.0;
                for (int ii = 0; ii < ns; ii++) acc += sRed[r * 64 + ii * CS + cc];
                if (qo <= 2 * uph) {
                    const bool og = (qo == 0);
                    const int colo = og ? 2 * uph : qo - 1;
                    if (r < 8) pg[r * PP + colo] = acc + (og ? sDef[k * 8 + r] : 0.0);
                    else pg[8 * PP + colo] = acc + ((og && s < uph) ? sHval[s] : 0.0);
                }
            }
            wsync();
        }
        }
    }
    if (sa.dbg && b == 0 && lane == 0) {
        const long long t3 = __builtin_readcyclecounter();
        sa.dbg[0] = (double)(t1 - t0); sa.dbg[1] = (double)(t2 - t1); sa.dbg[2] = (double)(t3 - t2);
    }
}

// ---- the prologue on the matrix cores (round 4): the column recursions G^(i) <- A^(i) G^(i) as v_mfma_f64_4x4x4_4b products.
// The instruction computes four independent 4 x 4 x 4 products, one per 4-lane block of a DPP row: lane = 16 q + 4 blk + x,
// A_blk[i][k] on lane (q = k, x = i), B_blk[k][j] and D_blk[i][j] on lane (q = row, x = j) (pipe_kernels.hpp, probe_mfma_4x4x4.cpp).
// A sample's G (8 rows x 64 columns) lives in that B / D layout: register W[rb][cg] of lane (q, blk, x) is row 4 rb + q of column
// 16 cg + 4 blk + x (column 0: the constant column g, column c >= 1: input column c - 1) -- eight doubles per sample, as before;
// the output of a product is the input of the next stage's. The A operand of block (rb, kb) is the SAME for all columns: lane
// (q, x) reads entry (4 rb + x, 4 kb + q) of the sample's 8 x 8 stage matrix straight out of the compact record in LDS (the
// structural zeros, ones and dt are three constants appended to the record's image, so it is a plain load at a lane-constant
// offset): FOUR LDS reads per (stage, sample) where a lane-per-column kernel (built first in round 4, HISTORY.md) needs 45
// broadcasts and is bound by its LDS instructions. And the matrix instructions only run over the column groups that hold a live column
// (stage k has 2 k + 3: one group of 16 for the first seven stages, two for the next eight, ...), which a lane mapping cannot
// do. Per (stage, sample, group): 4 products (2 x 2 blocks of the 8 x 8 matrix), the input column, the PCE mean and the chance row
// as 6 FMAs. Two wavefronts per OCP share the samples (five each: 80 registers of column state) and meet through LDS once per stage; the
// chance row is reduced over the four row lanes (q) of a column once per stage (quad_sum).
constexpr int SN_MREC = 56;                  // LDS image of a record: ABS doubles | 0.0 | 1.0 | dt
__host__ __device__ inline int sn_mfma_lds_doubles(int uph, int ns)
{
    return 2 * uph * ns + (uph + 1) + uph * 8 + SN_B16 * SN_B16 + uph * SN_B16 + 2 * ns * SN_MREC + 2 * ns * 5 + 12 * 64;
}
// NWV wavefronts per OCP, NSW samples per wavefront (NWV x NSW >= n_samples). (A variant without the per-sample guards for the
// reference's ten samples was built in round 5: 30 % fewer vector instructions in the stage loops, and 31 registers spilled around
// them -- +35 us per launch; not kept.)
template <int NSW, int NWV>
__global__ void __launch_bounds__(64 * NWV, 2) snmpc_prologue_mfma_kernel(const SnArgs sa)
{
    extern __shared__ __attribute__((aligned(16))) double sn_lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    constexpr int NT = 64 * NWV;
    auto sync = [] { if (NWV > 1) __syncthreads(); else wsync(); };
    if (b >= sa.batch) return;
    const int N = sa.N, ns = sa.ns, L = sa.L, uph = sa.uph;
    const int nitem = uph * ns;
    double *sH = sn_lds, *sCoef = sH + nitem, *sHval = sCoef + nitem, *sDef = sHval + (uph + 1), *sA = sDef + uph * 8;
    double *sC = sA + SN_B16 * SN_B16, *sRec = sC + uph * SN_B16, *sG4 = sRec + 2 * ns * SN_MREC, *sX = sG4 + 2 * ns * 5;
    const double dt = sa.dt;
    const double *gX = sa.X + (size_t)b * (N + 1) * NX;
    const double *gXS = sa.XS + (size_t)b * (N + 1) * ns * NX;
    const double *ws2 = sa.ws2 + (size_t)b * uph * ns * ABS;
    const double *ggh = sa.gh + (size_t)b * nitem * 5;
    const int PP = sn_pro_pitch(uph), PSTAGE = 9 * PP;
    double *pro = sa.pro + (size_t)b * uph * PSTAGE;

    // ---- gg values of the items; PCE coefficients c = A h, weights d(E + kappa sqrt(Var)) / d h_i; the nominal copy's own defect
    //      (the arithmetic of snmpc_prologue_kernel's P1 / P2, term by term)
    for (int o = tid; o < nitem; o += NT) sH[o] = ggh[o * 5];
    for (int o = tid; o < L * ns; o += NT) sA[o] = sa.Apce[o];
    // the constants behind every record image and gradient table (both buffers), written once
    for (int o = tid; o < 2 * ns; o += NT) {
        sRec[o * SN_MREC + ABS] = 0.0; sRec[o * SN_MREC + ABS + 1] = 1.0; sRec[o * SN_MREC + ABS + 2] = dt;
        sG4[o * 5 + 4] = 0.0;
    }
    __syncthreads();
    for (int o = tid; o < uph * L; o += NT) {
        const int k = o / L, l = o - k * L;
        double cl = 0.0;
#pragma unroll
        for (int j = 0; j < SN_B16; j++) cl += (j < ns) ? sA[l * ns + j] * sH[k * ns + j] : 0.0;
        sC[o] = cl;
    }
    __syncthreads();
    for (int item = tid; item < nitem; item += NT) {
        const int k = item / ns, i = item - k * ns;
        double w = 0.0;
        if (k >= 1) {
            double var = 0.0, acc = 0.0;
#pragma unroll
            for (int l = 1; l < SN_B16; l++) {
                const double cl = (l < L) ? sC[k * L + l] : 0.0;
                var += cl * cl; acc += (l < L) ? cl * sA[l * ns + i] : 0.0;
            }
            const double sd = sqrt(var);
            w = sA[i] + ((sd > 0.0) ? sa.kappa * acc / sd : 0.0);
            if (i == 0) sHval[k] = sC[k * L] + sa.kappa * sd;
        }
        sCoef[item] = w;
    }
    for (int o = tid; o < uph * 8; o += NT) {
        const int s = (o >> 3) + 1, r = o & 7;
        double acc = -gX[s * NX + r];
#pragma unroll
        for (int ii = 0; ii < SN_B16; ii++) acc += (ii < ns) ? sA[ii] * gXS[((size_t)s * ns + ii) * NX + r] : 0.0;
        sDef[o] = acc;
    }

    // ---- lane roles
    const int lane = tid & 63, grp = tid >> 6;
    const int q = lane >> 4, blk = (lane >> 2) & 3, x = lane & 3;
    const int i0 = grp * NSW;                          // first sample of this wavefront
    const int nrec = ns * ABS, ngh = ns * 4;
    // entry (r, c) of a stage matrix as an index into the record image: Sp / S entries, or the constants 0.0 / 1.0 behind the record
    auto aidx = [](int r, int c) -> int {
        if (c < 2) return (r == c) ? ABS + 1 : ABS;
        if (c == 2) return (r < 2) ? r : ((r == 2) ? ABS + 1 : ABS);
        if (r < 6) return 2 + 7 * r + (c - 3);
        return (r == c) ? ABS + 1 : ABS;
    };
    int oA[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int kb = 0; kb < 2; kb++) oA[rb][kb] = aidx(4 * rb + x, 4 * kb + q) + i0 * SN_MREC;          // (+ the wavefront's first sample: sample i of it is an immediate offset)
    // the input column B[:, r0] (rows 6, 7: dt on the row the input integrates into) and the defect b, rows 4 rb + q
    const int r0 = (x + 1) & 1;                        // parity of the input column c - 1 of column c = 16 cg + 4 blk + x
    int oB[2], oG[2], oV[2];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
        const int row = 4 * rb + q;
        oB[rb] = ((row < 6) ? 2 + 7 * row + 5 + r0 : ((row == 6) ? (r0 ? ABS + 2 : ABS) : (r0 ? ABS : ABS + 2))) + i0 * SN_MREC;
        oG[rb] = 44 + row + i0 * SN_MREC;
    }
    // gradient of the gg value w.r.t. (vl, vt, r, a) = rows 3, 4, 5, 7: index into the sample's table [g3, g4, g5, g7, 0.0]
    oV[0] = ((q == 3) ? 0 : 4) + i0 * 5;
    oV[1] = ((q == 0) ? 1 : (q == 1) ? 2 : (q == 3) ? 3 : 4) + i0 * 5;
    constexpr int NCH = (NWV * NSW * ABS + NT - 1) / NT;

    // phase cycle counters (scripts/dev/sn_mfma_phases.py): in the development build only -- six 64-bit accumulators are 14 registers the
    // shipped kernel has no room for
#ifdef TUM_DEV_KERNELS
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define SN_TICK(j) do { if (sa.dbg && b == 256) { const long long t_ = __builtin_readcyclecounter(); tacc[j] += t_ - tprev; tprev = t_; } } while (0)
#else
#define SN_TICK(j) do { } while (0)
#endif
    for (int phase = 0; phase < 2; phase++) {
        if (phase == 1 && 2 * uph < 64) break;         // (no column beyond 63)
        const int kbeg = phase ? 31 : 0;               // input column 63 belongs to stage 31
        const int cbase = 64 * phase + 4 * blk + x;    // column of this lane in group 0
        const bool isg0 = (phase == 0) && blk == 0 && x == 0;      // (group 0 only) the constant column
        double W[NSW][2][4];
#pragma unroll
        for (int i = 0; i < NSW; i++) {
            const int ii = (i0 + i < ns) ? i0 + i : ns - 1;
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const int row = 4 * rb + q;
                const double d0 = sa.xs0[((size_t)b * ns + ii) * NX + row] - gXS[(size_t)ii * NX + row];
                W[i][rb][0] = (isg0 && i0 + i < ns) ? d0 : 0.0;
#pragma unroll
                for (int cg = 1; cg < 4; cg++) W[i][rb][cg] = 0.0;
            }
        }
        double pre[NCH], preg = 0.0;
        auto fetch = [&](int k) {
#pragma unroll
            for (int c = 0; c < NCH; c++) { const int idx = tid + NT * c; pre[c] = (idx < nrec) ? ws2[(size_t)k * nrec + idx] : 0.0; }
            if (tid < ngh && k + 1 < uph) { const int i = tid >> 2, e = tid & 3; preg = ggh[((size_t)(k + 1) * ns + i) * 5 + 1 + e]; }
        };
        auto stash = [&](int k) {
            double *dr = sRec + (k & 1) * ns * SN_MREC, *dg = sG4 + (k & 1) * ns * 5;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int idx = tid + NT * c, smp = idx / ABS, f = idx - smp * ABS;
                if (idx < nrec) dr[smp * SN_MREC + f] = pre[c];
            }
            if (tid < ngh) dg[(tid >> 2) * 5 + (tid & 3)] = preg;
        };
        sync();                                        // (phase 1: everybody is through with the buffers of phase 0)
        if (kbeg < uph) { fetch(kbeg); stash(kbeg); }
        // the number of column groups with a live column is a compile-time constant of the stage body (one instantiation per count,
        // the stages of a phase in four runs): with it a run-time value every `if (cg < ncg)` was a branch whose join copied the column
        // state -- ~100 of the ~500 vector instructions of a stage were such moves
        auto stage = [&](const int k, auto ncgc) __attribute__((always_inline)) {
            constexpr int ncg = decltype(ncgc)::value;
            const int s = k + 1;
            SN_TICK(5);
            sync();
            SN_TICK(0);
            if (k + 1 < uph) fetch(k + 1);
            const double *recs = sRec + (k & 1) * ns * SN_MREC, *g4s = sG4 + (k & 1) * ns * 5;
            double sel[4];
#pragma unroll
            for (int cg = 0; cg < 4; cg++) {
                const int c = cbase + 16 * cg;
                sel[cg] = ((cg == 0 && isg0) || (c >= 1 && c <= 2 * uph && ((c - 1) >> 1) == k)) ? 1.0 : 0.0;
            }
            double M[2][4], Mc[4];
#pragma unroll
            for (int cg = 0; cg < 4; cg++) { M[0][cg] = 0.0; M[1][cg] = 0.0; Mc[cg] = 0.0; }
#pragma unroll
            for (int i = 0; i < NSW; i++) {
                if (i0 + i < ns) {
                    asm volatile("" ::: "memory");      // (the operand reads of a sample stay behind the arithmetic of the one before)
                    const int gi = i0 + i;
                    const double *rec = recs + i * SN_MREC;          // (the wavefront's first sample is in the lane offsets)
                    const double a00 = rec[oA[0][0]], a01 = rec[oA[0][1]], a10 = rec[oA[1][0]], a11 = rec[oA[1][1]];
                    const double bn0 = rec[oB[0]], bn1 = rec[oB[1]];
                    const double bg0 = isg0 ? rec[oG[0]] : bn0, bg1 = isg0 ? rec[oG[1]] : bn1;      // (group 0: the g lanes take the defect)
                    const double ai = sA[gi];
                    const double cf = (s < uph) ? sCoef[s * ns + gi] : 0.0;
                    const double gv0 = cf * g4s[i * 5 + oV[0]], gv1 = cf * g4s[i * 5 + oV[1]];
#pragma unroll
                    for (int cg = 0; cg < ncg; cg++) {
                        double d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a00, W[i][0][cg], 0.0, 0, 0, 0);
                        double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a10, W[i][0][cg], 0.0, 0, 0, 0);
                        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a01, W[i][1][cg], d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a11, W[i][1][cg], d1, 0, 0, 0);
                        d0 += sel[cg] * (cg == 0 ? bg0 : bn0);
                        d1 += sel[cg] * (cg == 0 ? bg1 : bn1);
                        W[i][0][cg] = d0; W[i][1][cg] = d1;
                        M[0][cg] += ai * d0; M[1][cg] += ai * d1;
                        Mc[cg] += gv0 * d0 + gv1 * d1;
                    }
                }
            }
            SN_TICK(1);
            // the chance row: sum over the four row lanes of a column
#pragma unroll
            for (int cg = 0; cg < ncg; cg++) Mc[cg] = quad_sum(Mc[cg]);
            // the two sample halves meet: wavefront 1 hands its partial sums over, wavefront 0 adds and stores the stage
            if (NWV > 1 && grp == 1) {
#pragma unroll
                for (int cg = 0; cg < 4; cg++) { sX[(3 * cg + 0) * 64 + lane] = M[0][cg]; sX[(3 * cg + 1) * 64 + lane] = M[1][cg]; sX[(3 * cg + 2) * 64 + lane] = Mc[cg]; }
            }
            if (NWV > 1) __syncthreads();
            SN_TICK(2);
            // the next stage's records go to LDS BEFORE this stage's results go out: stores count in the same counter as loads on
            // this part, and a wait for the records behind the stores would wait for the stores' round trip to L2 as well
            if (k + 1 < uph) stash(k + 1);
            SN_TICK(3);
            if (grp == 0) {
                double *pg = pro + (size_t)k * PSTAGE;
#pragma unroll
                for (int cg = 0; cg < ncg; cg++) {
                    const int c = cbase + 16 * cg;
                    const bool isg = (cg == 0) && isg0;
                    if (c <= 2 * k + 2 && c <= 2 * uph) {      // (the live columns only: the rest of the buffer is zero and stays zero)
                        const int colo = isg ? 2 * uph : c - 1;
#pragma unroll
                        for (int rb = 0; rb < 2; rb++) {
                            const int row = 4 * rb + q;
                            pg[row * PP + colo] = M[rb][cg] + (NWV > 1 ? sX[(3 * cg + rb) * 64 + lane] : 0.0) + (isg ? sDef[k * 8 + row] : 0.0);
                        }
                        if (q == 0) pg[8 * PP + colo] = Mc[cg] + (NWV > 1 ? sX[(3 * cg + 2) * 64 + lane] : 0.0) + ((isg && s < uph) ? sHval[s] : 0.0);
                    }
                }
            }
            SN_TICK(4);
        };
        // column groups with a live column in stage k (live columns: 0 .. 2 k + 2): nlive = 2 k + 3 - 64 phase, one group per 16
        static_for<1, 4>([&](auto ncgc) {
            constexpr int ncg = decltype(ncgc)::value;
            const int klo = (ncg == 1) ? kbeg : (16 * (ncg - 1) + 64 * phase - 1) / 2;          // first k with nlive >= 16 (ncg - 1) + 1
            const int khi = (ncg == 4) ? uph : (16 * ncg + 64 * phase - 1) / 2;                  // first k with nlive >= 16 ncg + 1
            for (int k = (klo > kbeg ? klo : kbeg); k < khi && k < uph; k++) stage(k, ncgc);
        });
    }
#ifdef TUM_DEV_KERNELS
    if (sa.dbg && b == 256 && lane == 0)
        for (int j = 0; j < 6; j++) sa.dbg[10 * grp + j] = (double)tacc[j];
#endif
#undef SN_TICK
}

// full step of the sample copies: dx^(i)_0 = xs0 - X^(i)_0, dx^(i)_{k+1} = A dx^(i)_k + B du_k + b for k < uph; frozen
// afterwards (X^(i)_k = X^(i)_uph for k > uph, pred_model_dynamic_disc.py:203). One lane per sample. The frozen copies
// are NOT written here: nothing on the solve path reads the sample copies of a stage > uph, and writing them is 22 KB per
// instance and solve (N = 40, uph = 5, ten samples; 92 MB per 4096-instance solve). The instance is flagged instead and
// snmpc_freeze_kernel brings the stages > uph up to date when somebody asks for them (get / set of a stacked state of
// such a stage, a change of uph).
// Pipeline (sa.Xn set): the NOMINAL copy of the stages 1..uph is written here as well. In the stacked model it is the PCE mean of
// the sample copies (x0+ = sum_i A[0, i] x_i+, pred_model_dynamic_disc.py:208-210), and the linearised step keeps that:
// X_nom,s + dx_nom,s = sum_i a_i (X^(i)_s + dx^(i)_s) -- a 16-lane sum of what the lanes have just computed. Round 3's expansion
// kernel formed the same step as G_nom,s dU + g_nom,s from the prologue's matrices: 2 s products per row and stage on eight
// lanes and 200 KB of hand-over buffer read again per instance at UPH = Tp (0.21 ms per 4096 instances, now gone).
template <int NSB = SN_B16>
__global__ void __launch_bounds__(64) snmpc_epilogue_kernel(const SnArgs sa)
{
    // the ns records of a stage are contiguous in ws2: all 64 lanes fetch them (coalesced, one stage ahead) and the sample
    // lanes read theirs from LDS -- a lane reading its own 424-byte record field by field touches ns sectors per load
    __shared__ double sRec[NSB * ABS];
    const int lane = threadIdx.x, b = blockIdx.x;
    if (b >= sa.batch || sa.status[b] != 0) return;
    const int N = sa.N, ns = sa.ns, uph = sa.uph;
    if (lane == 0 && uph < N) sa.xs_dirty[b] = 1;
    const bool act = lane < ns;
    const int i = act ? lane : 0;
    double *gXS = sa.XS + (size_t)b * (N + 1) * ns * NX;
    double *gXn = sa.Xn ? sa.Xn + (size_t)b * (N + 1) * NX : nullptr;
    const double ai = (sa.Xn && act) ? sa.Apce[i] : 0.0;                 // row 0 of A_pce: the PCE mean
    const double *ws2 = sa.ws2 + (size_t)b * uph * ns * ABS;
    const double *dv = sa.dv + (size_t)b * sa.dv_stride;
    const int nrec = ns * ABS;
    constexpr int NCH = (NSB * ABS + 63) / 64;
    double pre[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; pre[c] = (uph > 0 && idx < nrec) ? ws2[idx] : 0.0; }
    double dx[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double x = gXS[(size_t)i * NX + r];
        dx[r] = sa.xs0[((size_t)b * ns + i) * NX + r] - x;
        if (act) gXS[(size_t)i * NX + r] = x + dx[r];
    }
    // the old sample copies of the next stage, requested one stage ahead like the records
    double xo[8];
#pragma unroll
    for (int r = 0; r < 8; r++) xo[r] = (uph > 0) ? gXS[((size_t)1 * ns + i) * NX + r] : 0.0;
    for (int k = 0; k < uph; k++) {
        wsync();
#pragma unroll
        for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; if (idx < nrec) sRec[idx] = pre[c]; }
        wsync();
        if (k + 1 < uph) {
#pragma unroll
            for (int c = 0; c < NCH; c++) { const int idx = lane + 64 * c; pre[c] = (idx < nrec) ? ws2[(size_t)(k + 1) * nrec + idx] : 0.0; }
        }
        const double *rec = sRec + i * ABS;
        const double du0 = dv[2 * k], du1 = dv[2 * k + 1];
        apply_A(rec, dx);
#pragma unroll
        for (int r = 0; r < 6; r++) dx[r] += rec[2 + r * 7 + 5] * du0 + rec[2 + r * 7 + 6] * du1;
        dx[6] += sa.dt * du1; dx[7] += sa.dt * du0;
#pragma unroll
        for (int r = 0; r < 8; r++) dx[r] += rec[44 + r];
        double *xq = gXS + ((size_t)(k + 1) * ns + i) * NX;
        double xn[8];
#pragma unroll
        for (int r = 0; r < 8; r++) xn[r] = xo[r] + dx[r];
        if (k + 2 <= uph) {
#pragma unroll
            for (int r = 0; r < 8; r++) xo[r] = gXS[((size_t)(k + 2) * ns + i) * NX + r];
        }
        if (act) {
#pragma unroll
            for (int r = 0; r < 8; r++) xq[r] = xn[r];
        }
        if (gXn) {
            // PCE mean of the new sample copies: sum over the sample lanes of DPP row 0, total on lane 15 (more than 16 samples: row 1's total, on
            // lane 31, joins it)
            double m[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                double v = ai * xn[r];
                v += row_shr<1>(v); v += row_shr<2>(v); v += row_shr<4>(v); v += row_shr<8>(v);
                if constexpr (NSB > 16) v += rl(v, 31);
                m[r] = v;
            }
            if (lane == 15) {
                double *xd = gXn + (size_t)(k + 1) * NX;
                if (k + 1 == uph) {
                    double *du = sa.dxu + (size_t)b * sa.dv_stride;
#pragma unroll
                    for (int r = 0; r < 8; r++) du[r] = m[r] - xd[r];
                }
#pragma unroll
                for (int r = 0; r < 8; r++) xd[r] = m[r];
            }
        }
    }
}

// the deferred part of the epilogue: X^(i)_k = X^(i)_uph for k > uph on the flagged instances (contiguous, coalesced)
__global__ void __launch_bounds__(256) snmpc_freeze_kernel(double *XS, int *dirty, int N, int ns, int uph, int batch)
{
    const int b = blockIdx.x;
    if (b >= batch || !dirty[b]) return;
    const int per = ns * NX;
    double *g = XS + (size_t)b * (N + 1) * per;
    for (int o = threadIdx.x; o < (N - uph) * per; o += blockDim.x) g[(size_t)(uph + 1) * per + o] = g[(size_t)uph * per + o % per];
    __syncthreads();
    if (threadIdx.x == 0) dirty[b] = 0;
}

// sample initial conditions from the nominal one: xs0[b][i] = x0[b] + offs[i]  (compute_x0dist, stochastic_mpc_utils.py:78-91)
__global__ void snmpc_fanout_kernel(double *xs0, const double *x0, const double *offs, int ns, int batch)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * ns * NX) return;
    const int r = t & 7, bi = t >> 3, b = bi / ns, i = bi - b * ns;
    xs0[t] = x0[b * NX + r] + offs[i * NX + r];
}

// cold start of the sample copies: X^(i)_k = xs0^(i) for all k (SNMPC_class.py:126-127)
__global__ void snmpc_cold_start_kernel(double *XS, const double *xs0, int N, int ns, int batch)
{
    const int b = blockIdx.x;
    if (b >= batch) return;
    const int per = ns * NX;
    for (int i = threadIdx.x; i < (N + 1) * per; i += blockDim.x) XS[(size_t)b * (N + 1) * per + i] = xs0[(size_t)b * per + i % per];
}

}  // namespace tum
