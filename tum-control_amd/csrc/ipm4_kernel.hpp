// ipm4_kernel.hpp -- the interior point kernel of the pipeline with FOUR wavefronts per OCP.
//
// ipm_kernel (pipe_kernels.hpp) gives an OCP one wavefront with the whole register file of its SIMD: four OCPs per CU, one
// wavefront per SIMD, which issues in about half of its cycles (f64 VALU at 7.5 cycles per instruction from a lone wavefront,
// LDS / cross-lane round trips and MFMA results waited for with nothing else to run). Here an OCP is a 256-thread workgroup
// whose four wavefronts each hold a QUARTER of the live state -- below 128 registers -- so that four workgroups (sixteen
// wavefronts) share a CU: the same four OCPs per CU (40 KiB of LDS each), but every SIMD has four wavefronts to pick from.
//
//   rows        one row SIDE per lane (240 of 256 lanes): lane pair (2r, 2r+1) = lower / upper side of row r,
//               rows 0..39 steering-rate boxes, 40..79 steering-angle rows, 80..119 gg rows: six doubles of state per lane
//   v space     block T (variables 16T..16T+15) lives on wavefront T mod 4, replicated over the four DPP rows
//   gg rows     wavefront w keeps the MFMA operands of tile column w (and 4 on wavefront 0): <= 12 doubles
//   KKT matrix  tile row I is assembled and updated by wavefront I mod 4; the 4x4 pivot chain of the diagonal tile (J, J)
//               runs on wavefront J mod 4, which publishes the two MFMA operands of every micro-panel (P = L^-T D^-1 and
//               -L D) through LDS; the tiles below scale and update themselves from those, one barrier later
//   solves      one wavefront per solve (the substitutions are a dependency chain), taken in turn
// Same method, same numbers as ipm_kernel up to the order of the wave-level reductions.
#pragma once
#include "pipe_kernels.hpp"

namespace tum {

struct I4 {     // LDS carve (doubles), NT = 5
    static constexpr int LPK = PD<5>::LPK;
    static constexpr int L_M = 0;                    // LPK     KKT matrix / L D L' factor (packed lower triangle)
    static constexpr int L_P = L_M + LPK + 16;       // 4 x 64  P operand of the four micro-panels of the current diagonal tile (+16: slack behind the last packed row)
    static constexpr int L_BD = L_P + 256;           // 4 x 64  -L D operand
    static constexpr int L_AS = L_BD + 256;          // 10 x 64 gamma-scaled gg chunks of the current block column
    static constexpr int L_ROW = L_AS + 640;         // 128     one scalar per row (gamma / weight)
    static constexpr int L_SFX = L_ROW + 128;        // 48      suffix / prefix sums over the steering-angle rows
    static constexpr int L_V = L_SFX + 48;           // 80      a v-space vector (rhs, then dv)
    static constexpr int L_PART = L_V + 80;          // 5 x 40  C dv of the gg rows, partial sums per tile column
    static constexpr int L_RED = L_PART + 200;       // 32      cross-wavefront reductions
    static constexpr int L_END = L_RED + 32;
    static constexpr int BYTES = L_END * 8;
};
static_assert(I4::BYTES <= 40 * 1024, "four workgroups per CU");

template <int CTRL> __device__ __forceinline__ double quad_perm(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// the body of one wavefront; W = its index in the workgroup (compile-time: the tiles, gg chunks and v blocks it owns are fixed)
// PROF: cycle counters per phase and wavefront into ka.prof[b][12] (slot 3 W + {0: row phases, 1: assembly + factorisation
// incl. waiting at its barriers, 2: solves / waiting for the solving wavefront}; development aid)
#define I4_TICK(slot) do { if (PROF) { const long long t_ = __builtin_readcyclecounter(); pacc[slot] += t_ - tprev; tprev = t_; } } while (0)
template <int W, bool PROF>
__device__ __forceinline__ void ipm4_body(const PArgs &pa, double *lds)
{
    long long pacc[3] = {0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
    using D = PD<5>;
    constexpr int NT = 5, NVP = 80, NMAX = 40, NTT = 15, NC = 10;
    constexpr int PV_Q = D::PV_Q, PV_D = D::PV_D, PV_DV = D::PV_DV, PV_SC = D::PV_SC, PVEC = D::PVEC;
    const KArgs &ka = pa.ka;
    const int b = ka.order ? ka.order[blockIdx.x] : (int)blockIdx.x;
    constexpr int w = W;
    const int l_outer = threadIdx.x & 63;
    const int N = ka.N, nv = 2 * N, NB = N + 1;
    const double dt = ka.dt, dt2 = dt * dt;
    double *sM = lds + I4::L_M, *sP = lds + I4::L_P, *sBd = lds + I4::L_BD, *sAs = lds + I4::L_AS, *sRow = lds + I4::L_ROW,
           *sSfx = lds + I4::L_SFX, *sV = lds + I4::L_V, *sPart = lds + I4::L_PART, *sRed = lds + I4::L_RED;
    const double *gU = ka.U + (size_t)b * N * NU;
    const double *gpen = ka.pen + (size_t)b * 36;
    const double *gbnd = ka.bnd + (size_t)b * 6 * NB;
    double *gvec = pa.vec + (size_t)b * PVEC;
    auto tidx = [](int K, int I) { return D::tidx(K, I); };
    auto cidx = [](int c, int T) { return D::cidx(c, T); };

    // Everything derived from the lane id. Instantiated from an OPAQUE copy of the lane id inside the iteration loop (as in
    // ipm_kernel): otherwise the loop-invariant code motion hoists every lane predicate, packed-row offset and LDS address of
    // the whole iteration into the loop preheader and the register allocator spills them (measured: 58 spill stores there).
#define I4_DEFS(LANE) \
    const int l = (LANE), tid = 64 * W + l, lq = l >> 4, lc = l & 15; \
    const int jA = 16 * W + lc, jB = 64 + lc; \
    const bool vAon = jA < nv, vBon = (W == 0) && (jB < nv); \
    const int r = tid >> 1, sd = tid & 1; \
    const int ty = (r < NMAX) ? 0 : (r < 2 * NMAX) ? 1 : (r < 3 * NMAX) ? 2 : 3; \
    const int kst = (ty == 0) ? r : (ty == 1) ? r - (NMAX - 1) : r - (2 * NMAX - 1);      /* stage of the row */ \
    const bool on = (ty == 0) ? (kst < N) : (ty < 3 && kst <= N); \
    const double eps = sd ? -1.0 : 1.0; \
    const int rbA = lpk(16 * W + lc, 0), rbB = lpk(64 + lc, 0);      /* packed rows of this lane in tile row W / tile row 4 */ \
    /* one scalar per ROW from the two lanes of a pair (x: this side's signed contribution) */ \
    auto row_publish = [&](double x) { \
        const double tot = x + quad_perm<0xB1>(x);          /* lane ^ 1 */ \
        if (sd == 0 && r < 128) sRow[r] = on ? tot : 0.0; \
    }; \
    /* suffix sums over the steering-angle rows: sSfx[k] = sum_{stage s >= k} sRow[40 + s - 1]   (one wavefront) */ \
    auto steer_suffix = [&]() { \
        const double x = (l < N) ? sRow[NMAX + l] : 0.0; \
        const double sfx = wave_suffix(x, l); \
        if (l <= NMAX) sSfx[l + 1] = (l < N) ? sfx : 0.0; \
    }; \
    /* C' w for the v blocks of this wavefront (sRow = row weights, sSfx = their suffix sums over the steering rows) */ \
    auto ctw = [&](double &oA, double &oB) { \
        double eA = 0.0, eA1 = 0.0; \
    _Pragma("unroll") \
        for (int c = 2 * W; c < NC; c++) { const double wv = sRow[2 * NMAX + 4 * c + lq]; if (c & 1) eA1 += cA[c] * wv; else eA += cA[c] * wv; } \
        const double tA = quad_sum(eA + eA1); \
        const double tB = (W == 0) ? quad_sum(cB[0] * sRow[2 * NMAX + 32 + lq] + cB[1] * sRow[2 * NMAX + 36 + lq]) : 0.0; \
        const double sA = (jA & 1) ? sRow[jA >> 1] + dt * sSfx[(jA >> 1) + 1] : 0.0; \
        const double sB = (jB & 1) ? sRow[(jB >> 1) < NMAX ? (jB >> 1) : 0] + dt * sSfx[(jB >> 1) + 1] : 0.0; \
        oA = vAon ? tA + sA : 0.0; oB = vBon ? tB + sB : 0.0; \
    }; \
    /* workgroup reductions of up to four values per lane (three maxima, one sum): two barriers inside */ \
    auto wg_reduce4 = [&](double m0, double m1, double m2, double sm, double &o0, double &o1, double &o2, double &os) { \
        m0 = wave_max(m0); m1 = wave_max(m1); m2 = wave_max(m2); sm = wave_sum(sm); \
        __syncthreads(); \
        if (l == 0) { sRed[W] = m0; sRed[4 + W] = m1; sRed[8 + W] = m2; sRed[12 + W] = sm; } \
        __syncthreads(); \
        o0 = fmax(fmax(sRed[0], sRed[1]), fmax(sRed[2], sRed[3])); \
        o1 = fmax(fmax(sRed[4], sRed[5]), fmax(sRed[6], sRed[7])); \
        o2 = fmax(fmax(sRed[8], sRed[9]), fmax(sRed[10], sRed[11])); \
        os = (sRed[12] + sRed[13]) + (sRed[14] + sRed[15]); \
    }; \
    (void)tid; (void)lq; (void)lc; (void)jA; (void)jB; (void)vAon; (void)vBon; (void)r; (void)sd; (void)ty; (void)kst; (void)on; (void)eps; \
    (void)rbA; (void)rbB; (void)row_publish; (void)steer_suffix; (void)ctw; (void)wg_reduce4;

    // ---- this wavefront's share of the gg rows: operands of tile column W (chunks 2W..9); wavefront 0 also tile column 4
    double cA[NC], cB[2];          // (cA[c] exists for c >= 2 W only: the lower entries are never touched and cost no register)
    double vA = 0.0, vB = 0.0, rvA, rvB, qn;
    double zp, Zp, st_s, st_t, st_l, st_m, st_rs, st_rt;
    {
        I4_DEFS(l_outer)
        const double *gcw = pa.cws + (size_t)b * D::NCH * 64 + l;
#pragma unroll
        for (int c = 2 * W; c < NC; c++) cA[c] = gcw[cidx(c, W) * 64];
        if (W == 0) { cB[0] = gcw[cidx(8, 4) * 64]; cB[1] = gcw[cidx(9, 4) * 64]; } else { cB[0] = 0.0; cB[1] = 0.0; }
        // v space: block W on every wavefront, block 4 on wavefront 0 (replicated over the four DPP rows)
        const double qA = vAon ? gvec[PV_Q + jA] : 0.0, qB = vBon ? gvec[PV_Q + jB] : 0.0;
        {   // this lane's row side: initial point
            const int ks = on ? kst : ((ty == 0) ? 0 : 1);
            const int t_ = (ty < 3) ? ty : 2;
            const int pc = (t_ == 0) ? ((ks == 0) ? 0 : 1) : ((ks < N) ? 1 : 2);
            const double psc = (t_ == 0) ? dt : ((ks < N) ? dt : 1.0);
            const int pix = (pc * 3 + t_) * 4;
            zp = psc * gpen[pix + sd]; Zp = psc * gpen[pix + 2 + sd];
            const double dval = (t_ == 0) ? gU[2 * ks + 1] : gvec[PV_D + 2 * (ks - 1) + ((t_ == 2) ? 1 : 0)];
            const double bnd = gbnd[(2 * t_ + sd) * NB + ks];
            const double r0v = eps * (dval - bnd);
            const double s0 = ka.mu0 / (zp > 1e-6 ? zp : 1e-6);
            double t0_ = r0v + s0;
            if (t0_ < ka.t0) t0_ = ka.t0;
            const double lam = ka.mu0 / t0_;
            double ms = zp + Zp * s0 - lam;
            const double msf = 1e-2 * ka.mu0 / s0;
            if (ms < msf) ms = msf;
            st_s = on ? s0 : 1.0; st_t = on ? t0_ : 1.0; st_l = on ? lam : 1.0; st_m = on ? ms : 1.0;
            st_rs = on ? zp + Zp * s0 - lam - ms : 0.0;
            st_rt = on ? t0_ - r0v - s0 : 0.0;
        }
        for (int i = tid; i < I4::L_END - I4::L_P; i += 256) lds[I4::L_P + i] = 0.0;
        __syncthreads();
        // initial stationarity residual rv = q - C'(lam_l - lam_u), qn
        row_publish(sd ? -st_l : st_l);
        __syncthreads();
        if (W == 3) steer_suffix();
        __syncthreads();
        {
            double cA_, cB_;
            ctw(cA_, cB_);
            rvA = vAon ? qA - cA_ : 0.0; rvB = vBon ? qB - cB_ : 0.0;
        }
        {
            double d0, d1, d2;
            wg_reduce4(fmax(fabs(qA), fabs(qB)), 0.0, 0.0, 0.0, qn, d0, d1, d2);
            if (qn < 1.0) qn = 1.0;
        }
    }
    const double npairs = 12.0 * N, inv_npairs = 1.0 / npairs;
    int it = 0, qp_status = 1;
    double res_stat = 0.0, res_ineq = 0.0, res_comp = 0.0;
    const d4 *ghws0 = reinterpret_cast<const d4 *>(pa.hws) + (size_t)b * NTT * 64;

    for (;; it++) {
        int lane_v = l_outer;
        asm volatile("" : "+v"(lane_v));
        I4_DEFS(lane_v)
        const d4 *ghws = ghws0 + l;
        // ================================================================ row phase A: norms, convergence, gamma
        double gap;
        {
            const double c1 = st_t * st_l, c2 = st_s * st_m;
            double ls = on ? fabs(st_rs) : 0.0;
            if (lq == 0) ls = fmax(ls, fmax(fabs(rvA), fabs(rvB)));
            const double li = on ? fabs(st_rt) : 0.0, lcmp = on ? fmax(c1, c2) : 0.0, lg = on ? c1 + c2 : 0.0;
            const bool lane_nan = !(ls == ls) || !(li == li) || !(lcmp == lcmp);
            double gs;
            wg_reduce4(lane_nan ? __builtin_inf() : ls, li, lcmp, lg, res_stat, res_ineq, res_comp, gs);
            gap = gs * inv_npairs;
            if (!(res_stat < __builtin_inf()) || !(gap == gap) || !(res_ineq == res_ineq) || !(res_comp == res_comp)) { qp_status = 3; break; }
            if (!((res_stat > ka.tol_stat * qn) || (res_ineq > ka.tol_ineq) || (res_comp > ka.tol_comp))) { qp_status = 0; break; }
            if (it >= ka.iter_max) { qp_status = 1; break; }
        }
        const double rD = frcp(Zp * st_s + st_m);                     // D = 1 / (Z s + mu)
        const double rG = frcp(st_t + st_l * st_s * rD);              // G = 1 / (t + lam s D)
        row_publish(st_l * rG);                                       // gamma of the row = sum over its two sides
        __syncthreads();
        if (w == 3) steer_suffix();
        I4_TICK(0);
        // ================================================================ assembly + blocked L D L' over the four wavefronts
        int dmin_hi = 0x3ff00000;
        {
            if constexpr (W == 0) {      // gamma-scaled column operand of block column 0
#pragma unroll
                for (int c = 0; c < NC; c++) sAs[c * 64 + l] = cA[c] * sRow[2 * NMAX + 4 * c + lq];
            }
            __syncthreads();
            const double eu0 = (lq == 0) ? 1.0 : 0.0, eu1 = (lq == 1) ? 1.0 : 0.0, eu2 = (lq == 2) ? 1.0 : 0.0, eu3 = (lq == 3) ? 1.0 : 0.0;
#pragma unroll 1
            for (int J = 0; J < NT; J++) {
                const int dw = J & 3;                           // wavefront of the diagonal tile
                // tiles of block column J owned by this wavefront: I1 = first tile row >= J with I1 % 4 == w, I2 = 4 on wavefront 0
                const int I1 = (w >= J) ? w : ((w == 0 && J <= 4) ? 4 : -1);
                const bool two = (w == 0 && J == 0);            // wavefront 0 owns rows 0 and 4 of block column 0
                d4 T1 = {0.0, 0.0, 0.0, 0.0}, T2 = {0.0, 0.0, 0.0, 0.0};
                auto assemble = [&](const int I, const bool rowB) -> d4 {
                    d4 acc = ghws[tidx(J, I) * 64];
                    const int colv = 16 * I + lc;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int rowv = 16 * J + lq + 4 * jj;
                        const int mx = (rowv > colv) ? rowv : colv;
                        double add = ((rowv & 1) && (colv & 1) && mx < nv) ? dt2 * sSfx[(mx >> 1) + 1] : 0.0;
                        if (rowv == colv) add += ka.reg + (((rowv & 1) && rowv < nv) ? sRow[rowv >> 1] : 0.0);
                        acc[jj] += add;
                    }
                    // gg rows: chunks c >= 2 I (rows that reach tile column I); column operand from LDS, row operand from registers
                    if (rowB) {
                        acc = mfma(sAs[(8 - 2 * J) * 64 + l], cB[0], acc);
                        acc = mfma(sAs[(9 - 2 * J) * 64 + l], cB[1], acc);
                    } else {
#pragma unroll
                        for (int c = 2 * W; c < NC; c++) acc = mfma(sAs[(c - 2 * J) * 64 + l], cA[c], acc);      // (I == W here)
                    }
                    // left-looking update with the finished block columns K < J
                    const int rbI = rowB ? rbB : rbA, rbJ = lpk(16 * J + lc, 0);
                    for (int K = 0; K < J; K++)
#pragma unroll
                        for (int kc = 0; kc < 4; kc++) {
                            const int kk = 16 * K + 4 * kc + lq;
                            const double aJ = -sM[rbJ + kk] * sM[lpk(kk, kk)];
                            acc = mfma(aJ, sM[rbI + kk], acc);
                        }
                    return acc;
                };
                if (I1 >= 0) T1 = assemble(I1, I1 == 4);
                if (two) T2 = assemble(4, true);
                // ---- diagonal tile: four micro-panels on wavefront dw (T1 is the diagonal tile there)
                if (w == dw) {
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const int c0 = 16 * J + 4 * m;
                        const double a00 = readlane_f64(T1[m], 4 * m);
                        const double a10 = readlane_f64(T1[m], 4 * m + 1), a11 = readlane_f64(T1[m], 16 + 4 * m + 1);
                        const double a20 = readlane_f64(T1[m], 4 * m + 2), a21 = readlane_f64(T1[m], 16 + 4 * m + 2),
                                     a22 = readlane_f64(T1[m], 32 + 4 * m + 2);
                        const double a30 = readlane_f64(T1[m], 4 * m + 3), a31 = readlane_f64(T1[m], 16 + 4 * m + 3),
                                     a32 = readlane_f64(T1[m], 32 + 4 * m + 3), a33 = readlane_f64(T1[m], 48 + 4 * m + 3);
                        const double d0 = a00, i0 = frcp(d0);
                        const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
                        const double d1 = a11 - l10 * a10, i1 = frcp(d1);
                        const double y21 = a21 - l20 * a10, y31 = a31 - l30 * a10;
                        const double l21 = y21 * i1, l31 = y31 * i1;
                        const double d2 = a22 - l20 * a20 - l21 * y21, i2 = frcp(d2);
                        const double y32 = a32 - l30 * a20 - l31 * y21;
                        const double l32 = y32 * i2;
                        const double d3 = a33 - l30 * a30 - l31 * y31 - l32 * y32, i3 = frcp(d3);
                        dmin_hi = min(min(min(dmin_hi, __double2hiint(d0)), min(__double2hiint(d1), __double2hiint(d2))), __double2hiint(d3));
                        // P[k][x] = (L^-1)[x][k] / d_x on lane (k, x) = (lq, lc), x < 4
                        const double X1 = eu1 - l10 * eu0;
                        const double X2 = eu2 - l20 * eu0 - l21 * X1;
                        const double X3 = eu3 - l30 * eu0 - l31 * X1 - l32 * X2;
                        const double Xx = (lc == 0) ? eu0 : (lc == 1) ? X1 : (lc == 2) ? X2 : X3;
                        const double ix = (lc == 0) ? i0 : (lc == 1) ? i1 : (lc == 2) ? i2 : i3;
                        const double pop = (lc < 4) ? Xx * ix : 0.0;
                        const double dsel = (lq == 0) ? d0 : (lq == 1) ? d1 : (lq == 2) ? d2 : d3;
                        const int rel = lc - (4 * m + lq);              // row - column inside the diagonal tile
                        d4 z = {0.0, 0.0, 0.0, 0.0};
                        z = mfma(pop, T1[m], z);
                        const double Lcd = z[0];
                        if (rel == 0) sM[lpk(16 * J + lc, 0) + c0 + lq] = dsel;
                        const double bval = (rel > 0) ? Lcd : ((rel == 0) ? 1.0 : 0.0);
                        const double bd = -bval * dsel;
                        sP[m * 64 + l] = pop; sBd[m * 64 + l] = bd;
                        if (m < 3) T1 = mfma(bd, bval, T1);
                    }
                }
                __syncthreads();
                // ---- tiles below the diagonal one, and the identity tile on wavefront dw (-> inverse of the diagonal block)
                auto panel = [&](d4 T, const int rb_) {
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        d4 z = {0.0, 0.0, 0.0, 0.0};
                        z = mfma(sP[m * 64 + l], T[m], z);
                        const double Lc = z[0];
                        sM[rb_ + 16 * J + 4 * m + lq] = Lc;
                        if (m < 3) T = mfma(sBd[m * 64 + l], Lc, T);
                    }
                };
                if (w == dw) {
                    d4 Ti;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) Ti[jj] = (lc == lq + 4 * jj) ? 1.0 : 0.0;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const int c0 = 16 * J + 4 * m;
                        d4 z = {0.0, 0.0, 0.0, 0.0};
                        z = mfma(sP[m * 64 + l], Ti[m], z);
                        const double Lc = z[0];
                        const double dsl = sM[lpk(c0 + lq, c0 + lq)];
                        if (4 * m + lq > lc) sM[lpk(c0 + lq, 16 * J + lc)] = Lc * dsl;
                        if (m < 3) Ti = mfma(sBd[m * 64 + l], Lc, Ti);
                    }
                    if (two) panel(T2, rbB);
                } else if (I1 > J) panel(T1, (I1 == 4) ? rbB : rbA);
                if (w == ((J + 1) & 3) && J + 1 < NT) {     // column operand of the next block column (chunks 2(J+1)..9)
                    if (J + 1 == 4) { sAs[0 * 64 + l] = cB[0] * sRow[2 * NMAX + 32 + lq]; sAs[1 * 64 + l] = cB[1] * sRow[2 * NMAX + 36 + lq]; }
                    else {      // (J + 1 == W here: chunks 2 W .. 9)
#pragma unroll
                        for (int c = 2 * W; c < NC; c++) sAs[(c - 2 * W) * 64 + l] = cA[c] * sRow[2 * NMAX + 4 * c + lq];
                    }
                }
                __syncthreads();
            }
        }
        {   // a pivot that is not positive (or below 1e-300): the factorisation failed
            int mn = dmin_hi;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) mn = min(mn, __shfl_xor(mn, o, 64));
            if (l == 0) sRed[16 + w] = (double)mn;
            __syncthreads();
            const double mall = fmin(fmin(sRed[16], sRed[17]), fmin(sRed[18], sRed[19]));
            if (mall < (double)0x01a56e1f) { qp_status = 3; break; }
        }
        I4_TICK(1);
        // ================================================================ predictor / corrector
        double cross1 = 0.0, cross2 = 0.0, dvA = 0.0, dvB = 0.0, alpha = 1.0, sigma = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            int lane_p = l_outer;
            asm volatile("" : "+v"(lane_p));
            I4_DEFS(lane_p)
            const double tau = (pass == 1) ? fmax(sigma * gap, 0.1 * ka.tol_comp) : 0.0;
            double rc1 = st_t * st_l, rc2 = st_s * st_m;
            if (pass == 1) { rc1 += cross1 - tau; rc2 += cross2 - tau; }
            {
                const double gr = rG * (rc1 - st_l * (st_rt + (st_rs * st_s + rc2) * rD));
                __syncthreads();                                  // (everybody is done with sRow / sSfx of the phase before)
                row_publish(sd ? -gr : gr);
            }
            __syncthreads();
            if (w == 3) steer_suffix();
            __syncthreads();
            double bA, bB;
            ctw(bA, bB);
            bA = vAon ? -rvA - bA : 0.0; bB = vBon ? -rvB - bB : 0.0;
            if (lq == 0) { sV[jA] = bA; if (w == 0) sV[jB] = bB; }
            __syncthreads();
            I4_TICK(0);
            // ---- the solve: one wavefront, taken in turn
            if (w == ((2 * it + pass) & 3)) {
                int lane_s = l_outer;
                asm volatile("" : "+v"(lane_s));
                const int l = lane_s, lq = l >> 4, lc = l & 15;
                int ga[4];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) ga[jj] = ((l & 48) | ((lq + 4 * jj) & 15)) << 2;
                const bool ondiag = (lc >= lq) && (((lc - lq) & 3) == 0);
                // (the solved blocks go through the v-space buffer: block K is written by the lanes of DPP row 0 and read back as
                //  the four values lq + 4 jj a lane multiplies -- LDS reads instead of 20 more live registers of lane gathers)
                double bj[NT];
                int rb[NT];
#pragma unroll
                for (int J = 0; J < NT; J++) { bj[J] = sV[16 * J + lc]; rb[J] = lpk(16 * J + lc, 0); }
                wsync();
#pragma unroll
                for (int J = 0; J < NT; J++) {
                    double t = bj[J];
                    if (J > 0) {
                        double acc = 0.0, acc1 = 0.0;
#pragma unroll
                        for (int K = 0; K < J; K++)
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const double lv = sM[rb[J] + 16 * K + lq + 4 * jj], yv = sV[16 * K + lq + 4 * jj];
                                if (jj & 1) acc1 += lv * yv; else acc += lv * yv;
                            }
                        t -= quad_sum(acc + acc1);
                    }
                    double a2 = ondiag ? t : 0.0;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const double lv = sM[rb[J] + 16 * J + lq + 4 * jj];
                        const double tv = lane_gather(t, ga[jj]);
                        a2 += ((lq + 4 * jj < lc) ? lv : 0.0) * tv;
                    }
                    const double y = quad_sum(a2);
                    bj[J] = y * frcp(sM[rb[J] + 16 * J + lc]);
                    if (J < NT - 1) { if (lq == 0) sV[16 * J + lc] = y; wsync(); }
                }
#pragma unroll
                for (int J = NT - 1; J >= 0; J--) {
                    double t = bj[J];
                    if (J < NT - 1) {
                        double acc = 0.0, acc1 = 0.0;
#pragma unroll
                        for (int I = J + 1; I < NT; I++)
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const double lv = sM[lpk(16 * I + lq + 4 * jj, 0) + 16 * J + lc], xv = sV[16 * I + lq + 4 * jj];
                                if (jj & 1) acc1 += lv * xv; else acc += lv * xv;
                            }
                        t -= quad_sum(acc + acc1);
                    }
                    double a2 = ondiag ? t : 0.0;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const double lv = sM[lpk(16 * J + lq + 4 * jj, 0) + 16 * J + lc];
                        const double tv = lane_gather(t, ga[jj]);
                        a2 += ((lq + 4 * jj > lc) ? lv : 0.0) * tv;
                    }
                    const double x = quad_sum(a2);
                    if (lq == 0) sV[16 * J + lc] = x;
                    wsync();
                }
            }
            __syncthreads();
            I4_TICK(2);
            dvA = vAon ? sV[jA] : 0.0; dvB = vBon ? sV[jB] : 0.0;
            // ---- C dv: steering prefix sums on wavefront 3, gg partial sums per tile column on its wavefront
            if (w == 3) {
                const double x = (l < N) ? sV[2 * l + 1] : 0.0;
                const double pf = wave_prefix(x, l);
                if (l < NMAX) sSfx[l + 1] = pf;
            }
            {
                const double dA = sV[jA];
#pragma unroll
                for (int c = 2 * W; c < NC; c++) {
                    double a = cA[c] * dA;
                    a += row_shr<8>(a); a += row_shr<4>(a); a += row_shr<2>(a); a += row_shr<1>(a);
                    if (lc == 15) sPart[w * NMAX + 4 * c + lq] = a;
                }
                if (w == 0) {
                    const double dB = sV[jB];
#pragma unroll
                    for (int c = 8; c < NC; c++) {
                        double a = cB[c - 8] * dB;
                        a += row_shr<8>(a); a += row_shr<4>(a); a += row_shr<2>(a); a += row_shr<1>(a);
                        if (lc == 15) sPart[4 * NMAX + 4 * c + lq] = a;
                    }
                }
            }
            __syncthreads();
            double cdv;
            {
                const int ks = on ? kst : 1;
                if (ty == 0) cdv = sV[2 * (on ? kst : 0) + 1];
                else if (ty == 1) cdv = dt * sSfx[ks];
                else cdv = ((sPart[ks - 1] + sPart[NMAX + ks - 1]) + (sPart[2 * NMAX + ks - 1] + sPart[3 * NMAX + ks - 1])) + sPart[4 * NMAX + ks - 1];
            }
            // ---- step of this row side, step length
            double dsl, dtt, dl, dm, q;
            {
                const double is_ = frcp(st_s), il_ = frcp(st_l), it_ = frcp(st_t), im_ = frcp(st_m);
                const double iDs = st_s * rD, gam = st_l * rG;
                const double rho = -st_rt + rc1 * il_ - (st_rs + rc2 * is_) * iDs;
                dl = -gam * (eps * cdv + rho);
                dsl = (dl - st_rs - rc2 * is_) * iDs;
                dm = (-rc2 - st_m * dsl) * is_;
                dtt = (-rc1 - st_t * dl) * il_;
                if (!on) { dsl = 0.0; dtt = 0.0; dl = 0.0; dm = 0.0; }
                q = fmax(fmax(-dsl * is_, -dtt * it_), fmax(-dl * il_, -dm * im_));
                q = on ? q : 0.0;
            }
            double amax, u1, u2, musum;
            wg_reduce4(fmax(q, 1.0), 0.0, 0.0, 0.0, amax, u1, u2, musum);
            amax = frcp(amax);
            if (pass == 0) {
                const double pr_ = (st_t + amax * dtt) * (st_l + amax * dl) + (st_s + amax * dsl) * (st_m + amax * dm);
                wg_reduce4(0.0, 0.0, 0.0, on ? pr_ : 0.0, u1, u2, q, musum);
                const double ratio = musum * inv_npairs * frcp(gap);
                sigma = ratio * ratio * ratio;
                cross1 = (on && !(amax < 0.1)) ? dtt * dl : 0.0;
                cross2 = (on && !(amax < 0.1)) ? dsl * dm : 0.0;
            } else {
                alpha = (amax >= 1.0) ? 1.0 : 0.995 * amax;
                if (alpha >= 1e-12) {
                    const double om_ = 1.0 - alpha;
                    st_s += alpha * dsl; st_t += alpha * dtt; st_l += alpha * dl; st_m += alpha * dm;
                    st_rs *= om_; st_rt *= om_;
                }
            }
        }
        if (alpha < 1e-12) { qp_status = 2; break; }
        vA += alpha * dvA; vB += alpha * dvB;
        const double om = 1.0 - alpha;
        rvA *= om; rvB *= om;
        __syncthreads();
    }
    // ---- outputs: slack part of the cost, slack values, the step of the inputs (expansion kernel), status
    const int status = acados_status(qp_status);
    {
        I4_DEFS(l_outer)
        const double c = zp * st_s + 0.5 * Zp * st_s * st_s;
        double u0, u1, u2, scost;
        wg_reduce4(0.0, 0.0, 0.0, on ? c : 0.0, u0, u1, u2, scost);
        if (ka.slack && on) {
            double *sl = ka.slack + (size_t)b * 6 * N;
            const int idx = (ty == 0) ? kst : N + 2 * (kst - 1) + ((ty == 2) ? 1 : 0);
            sl[sd * 3 * N + idx] = st_s;
        }
        if (lq == 0) {
            gvec[PV_DV + jA] = vA;
            if (w == 0) gvec[PV_DV + jB] = vB;
        }
        I4_TICK(0);
        if (PROF && l == 0)
            for (int i = 0; i < 3; i++) ka.prof[(size_t)b * 12 + 3 * W + i] = pacc[i];
        if (tid == 0) {
            gvec[PV_SC] = scost;
            ka.status[b] = status;
            ka.qp_iter[b] = it;
            ka.qp_status[b] = qp_status;
            ka.res[b * 3 + 0] = res_stat; ka.res[b * 3 + 1] = res_ineq; ka.res[b * 3 + 2] = res_comp;
        }
    }
}

template <bool PROF>
__global__ void __launch_bounds__(256, 4) ipm4_kernel(const PArgs pa)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if ((int)blockIdx.x >= pa.ka.batch) return;
    const int w = threadIdx.x >> 6;          // (wave-uniform: every wavefront runs its own instantiation, the barriers pair up)
#ifdef IPM4_ONLY      // (development: register use of one instantiation alone)
    ipm4_body<IPM4_ONLY, PROF>(pa, lds);
#else
    if (w == 0) ipm4_body<0, PROF>(pa, lds);
    else if (w == 1) ipm4_body<1, PROF>(pa, lds);
    else if (w == 2) ipm4_body<2, PROF>(pa, lds);
    else ipm4_body<3, PROF>(pa, lds);
#endif
}

}  // namespace tum
