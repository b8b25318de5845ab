// pipe_kernels.hpp -- the SQP-RTI step as a PIPELINE of four kernels, each at the occupancy its phase allows
// (the fused kernel of nmpc_kernel.hpp runs every phase at the occupancy of its most demanding one: one wavefront per SIMD).
//
//   K1  lin_kernel      lane = (instance, stage): ERK4 x nsub with forward sensitivities, cost residuals, gg constraint.
//                       Full lanes (the fused kernel has 41 of 64 busy), LDS only for the transposed store of the records;
//                       390 registers: one wavefront per SIMD.
//   K2  cond_kernel     one wavefront per OCP: column recursion G_{k+1} = A_k G_k, Gauss-Newton SYRK on the matrix cores
//                       (15 register tiles), gg rows; hands H (tiles), C (MFMA operand layout), q, d to the workspace.
//   K3  ipm_kernel      one wavefront per OCP with the whole register file (450 of 512 registers, four OCPs per CU): LDS holds
//                       the factor of the KKT matrix -- five-tile build: as XOR-swizzled 16 x 16 tiles, the blocks below the
//                       diagonal, the inverse diagonal blocks and the pivot vector, 31 KiB (34 KiB in all); six tiles: packed by
//                       rows, 26 KiB -- the gg rows live in registers in MFMA operand layout, H is streamed tile by tile from
//                       the workspace (L2); the tile operands of the substitutions are read from LDS straight into accumulator
//                       registers. (Bounded to 256 registers -- two wavefronts per SIMD -- the same method was measured slower:
//                       HISTORY.md, round-4 document, section 7; the build switch went with round 6's clean-up.)
//   K3' ipm4_kernel     (ipm4_kernel.hpp) the same method with FOUR wavefronts per OCP, each below 128 registers.
//   K4  expand_kernel   one wavefront per OCP: dx recursion, full step, cost at the new iterate. For batches of at most one
//                       round of resident wavefronts the nominal OCP runs it as the tail of K3 instead (ipm_kernel<., ., true>).
//
// Workspace per instance (HBM/L2): stage records 41 x 64 doubles, H 15 x 64 x 4, C 30 x 64, q | d | dv 3 x 80.
// Reference semantics as in nmpc_kernel.hpp; the arithmetic of every phase is the fused kernel's.
#pragma once
#include "common_kernels.hpp"
#include "snmpc_kernels.hpp"

// wavefronts per SIMD the interior point kernel is bounded to (1: the whole register file; 2: an experiment build, HISTORY.md (round-4 document, section 7))

namespace tum {

constexpr int PREC = 64;                        // doubles per stage record
// record fields: [0,1] Sp | [2..43] S[6][7] | [44..51] defect b | [52..55] cost residuals | [56..59] g3 g5 g7 h | [60] delta_f
constexpr int PR_RES = 52, PR_GH = 56, PR_XD = 60;
// coupled SNMPC OCP only: [61, 62] gradient (vl, vt)/|v| of the speed row of the cost, [63] d h / d vt of the gg row at |v|
constexpr int PR_CV = 61, PR_G4 = 63;
// Everything that depends on the horizon limit is a function of NT_, the number of 16-wide MFMA tiles of the condensed QP:
// NT_ = 5 -> N <= 40 (what the fused kernel covers as well), NT_ = 6 -> N <= 48 (pipeline only).
template <int NT_> struct PD {
    static constexpr int NT = NT_, NVP = 16 * NT_, NMAX = 8 * NT_, NTT = NT_ * (NT_ + 1) / 2, LPK = NVP * (NVP + 1) / 2;
    static constexpr int NC = 2 * NT_;              // chunks of 4 gg rows
    static constexpr int NCH = NT_ * NT_ + NT_;     // (chunk c, tile column T) pairs with c >= 2T
    static constexpr int NB1 = NVP - 64;            // condensed variables beyond the first 64: "bank 1", lanes 0..NB1-1
    // rows of the interior point method per lane: NT_ = 5: two slots (box + steering on lanes 0..N-1, two gg rows on each of
    // the lanes 40..59); more tiles: three slots on the lanes 0..N-1 (box of stage l, steering and gg row of stage l+1)
    static constexpr int SLOTS = (NT_ == 5) ? 2 : 3;
    static constexpr int PV_Q = 0, PV_D = NVP, PV_DV = 2 * NVP, PV_SC = 3 * NVP, PVEC = 3 * NVP + 16;   // q | d | dv | slack cost
    // (C_WT: the weights of every stage; C_PARK: the Hessian tiles of the last block column wait here between the stages of the
    //  last segment -- d4 per lane and tile)
    static constexpr int C_NPARK = 3;
    static constexpr int C_REC = 0, C_STAGE = 2 * PREC, C_GS = C_STAGE + 4 * NVP, C_U0 = C_GS + 8,
                         C_WT = C_U0 + NVP, C_PARK = (C_WT + (NMAX + 1) * 6 + 1) & ~1, C_LDS = C_PARK + C_NPARK * 256;
    static constexpr int E_REC = 0, E_X = 2 * PREC, E_U = E_X + (NMAX + 1) * NX, E_DV = E_U + NVP, E_LDS = E_DV + NVP;
    // interior point kernel: the small vectors first (the diagonal-block substitution reads up to 15 doubles in front of a
    // packed row with a zero multiplier: in front of row 0 that lands on them, finite data), then the KKT matrix
    static constexpr int I_WH = 0;                  // NMAX     gamma / weights / C dv of the gg rows (index stage - 1)
    static constexpr int I_WB = I_WH + NMAX;        // NMAX     box-row scalars
    static constexpr int I_SFX = I_WB + NMAX;       // NMAX+2   suffix sums over the steering-angle rows
    static constexpr int I_DV = I_WB;               //   (alias: a v-space vector lives here between a solve and the next publish)
    static constexpr int I_DUMMY = I_SFX + NMAX + 2;
    static constexpr int I_PZ = I_DUMMY + 1;        // 18       quadratic slack penalties Zl, Zu of the 9 (class, row type) pairs
    // five-tile build: the factor is stored as TILES -- the NT (NT - 1) / 2 blocks below the diagonal and the NT inverse diagonal
    // blocks (unit diagonal, zeros above it, both written once) as 16 x 16 tiles of pitch 16 with the column index XOR-swizzled by
    // a bit permutation of the row (tile_swz below), the pivots as a vector of their own; L of a diagonal block itself is never
    // read again and is not stored. Every access pattern of the kernel then hits each LDS bank once: the column stores of a
    // micro-panel, the operand reads of the left-looking update and the tile operands of the forward substitution (row = lane
    // column) are conflict-free, the transposed reads of the backward substitution 2-way for two of their four rotations.
    // (Rounds 2-3 kept the factor packed-triangular by rows: 3-way conflicts on the substitutions' operands, 2-way on the
    // update's and on the pitch-17 inverse blocks, 850 conflict cycles per iteration; and every row start was a multiply.)
    // Six tiles (N = 41..48): no room for the inverse blocks as tiles of their own -- the factor stays packed, the inverse blocks
    // in the strict lower triangle of its diagonal tiles, read with masks.
    static constexpr bool DENSE_W = (NT_ == 5);
    static constexpr bool TILED = DENSE_W;
    static constexpr int NOT = NT_ * (NT_ - 1) / 2, W_TILE = 256;
    static constexpr int I_M = TILED ? ((I_PZ + 18 + 31) & ~31) : I_PZ + 18;      // LPK (packed) | NOT tiles, on a 256-byte boundary
    static constexpr int I_W = I_M + (TILED ? NOT * 256 : LPK);                    // NT tiles: the inverse diagonal blocks
    static constexpr int I_D = I_W + (TILED ? NT_ * W_TILE : 0);                   // NVP: the pivots
    static constexpr int I_BK = (I_D + (TILED ? NVP : 0) + 1) & ~1;      // 64 (16-byte aligned): the strip of a micro-panel
    static constexpr int I_ZERO = I_BK + 64;        // 1        0.0, written once
    // five-tile build: the steering and the box term of every v-space index, expanded per iteration (0 where an index has none):
    // the diagonal tiles gather their terms with one address register and instruction offsets
    static constexpr bool PRE_DIAG = (NT_ == 5);    // (the six-tile build has neither the registers nor the LDS for it)
    static constexpr int I_XS = I_ZERO + 2, I_XB = I_XS + (PRE_DIAG ? NVP : 0);
    static constexpr int I_LDS = I_XB + (PRE_DIAG ? NVP : 0), I_LDS_BYTES = I_LDS * 8;
    static_assert(NVP <= 2 * NMAX + 2, "the v-space alias must fit in the box / suffix buffers");
    static_assert(I_LDS_BYTES <= (NT_ <= 6 ? 40 : 64) * 1024, "four workgroups per CU (seven tiles: the packed factor alone is 51 KB -- three)");
    static __host__ __device__ constexpr int tidx(int K, int I) { return K * NT_ - K * (K - 1) / 2 + (I - K); }   // K <= I
    static __host__ __device__ constexpr int offt(int I, int K) { return I * (I - 1) / 2 + K; }                 // tile (I, K), I > K
    // physical column of (row, col) inside a tile: col ^ tile_swz(row); bits of the row: [3 2 1 0] -> [1 3 2 0]
    static __host__ __device__ constexpr int tile_swz(int row) { return ((row & 2) << 2) | ((row & 12) >> 1) | (row & 1); }
    static __host__ __device__ constexpr int cidx(int c, int T) { return T * (NC - T - 1) + c; }              // c >= 2T
};
static_assert(PD<5>::cidx(9, 4) == chidx(9, 4) && PD<5>::cidx(2, 1) == chidx(2, 1) && PD<5>::NCH == NCHV, "operand layout of the fused kernel's tile count");
// local names of the constants inside a kernel templated on NT_ (they hide the namespace-level ones of the fused kernel)
#define PD_LOCALS \
    using D = PD<NT_>; \
    constexpr int NT = D::NT, NVP = D::NVP, NMAX = D::NMAX, NTT = D::NTT, LPK = D::LPK, NC = D::NC, NCH = D::NCH, NB1 = D::NB1; \
    constexpr int SLOTS = D::SLOTS, PV_Q = D::PV_Q, PV_D = D::PV_D, PV_DV = D::PV_DV, PV_SC = D::PV_SC, PVEC = D::PVEC; \
    auto tidx = [](int K, int I) { return D::tidx(K, I); }; \
    auto cidx = [](int c, int T) { return D::cidx(c, T); }; \
    (void)NT; (void)NVP; (void)NMAX; (void)NTT; (void)LPK; (void)NC; (void)NCH; (void)NB1; (void)SLOTS; \
    (void)PV_Q; (void)PV_D; (void)PV_DV; (void)PV_SC; (void)PVEC; (void)tidx; (void)cidx;

struct PArgs {
    KArgs ka;
    double *rec;          // [b][N+1][PREC]
    double *hws;          // [b][NTT][64][4]
    double *cws;          // [b][NCH][64]
    double *vec;          // [b][PVEC]
};

// ---------------------------------------------------------------------------------------------------------------- K1
constexpr int L_PITCH = PREC + 1;               // LDS pitch of one item's record (odd: lane-strided writes hit distinct banks)
// SN = true: the nominal copy of the coupled SNMPC OCP (speed row |v|, gg limits looked up at |v|; the stages below the
// uncertainty propagation horizon are condensed by the prologue kernel and need no record here)
template <bool SN>
__global__ void __launch_bounds__(64, 1) lin_kernel(const PArgs pa)
{
    // the 64 records of a wavefront are one contiguous 32 KiB block of the workspace: they are transposed through LDS so that
    // every store instruction writes 512 contiguous bytes (a lane writing its own record would touch 64 segments per store)
    __shared__ double sT[64 * L_PITCH];
    const KArgs &ka = pa.ka;
    const int N = ka.N, NB = N + 1;
    const long long total = (long long)ka.batch * NB;
    const long long g0 = (long long)blockIdx.x * 64;
    const long long gl = g0 + threadIdx.x;
    const bool live = gl < total;
    const long long g = live ? gl : total - 1;       // (lanes beyond the last item shadow it and store nothing)
    const int b = (int)(g / NB), k = (int)(g - (long long)b * NB);
    const double *gX = ka.X + ((size_t)b * NB + k) * NX;
    double xk[8];
#pragma unroll
    for (int i = 0; i < 8; i++) xk[i] = gX[i];
    const double *yr = ka.yref + ((size_t)b * NB + k) * 6;
    double r[PREC];
#pragma unroll
    for (int i = 0; i < PREC; i++) r[i] = 0.0;
    r[PR_RES + 0] = xk[0] - yr[0];
    r[PR_RES + 1] = xk[1] - yr[1];
    r[PR_RES + 2] = wrap_yaw(xk[2]) - yr[2];
    if (SN) {
        const double vabs = sqrt(xk[3] * xk[3] + xk[4] * xk[4]), iv = (vabs > 0.0) ? 1.0 / vabs : 0.0;
        r[PR_RES + 3] = vabs - yr[3];
        r[PR_CV] = xk[3] * iv; r[PR_CV + 1] = xk[4] * iv;
    } else r[PR_RES + 3] = xk[3] - yr[3];
    r[PR_XD] = xk[6];
    if (k >= 1) {
        double h, g3, g5, g7;
        if (SN) {
            double g4;
            h_con_vabs(ka.mp, xk[3], xk[4], xk[5], xk[7], h, g3, g4, g5, g7);
            r[PR_G4] = g4;
        } else h_con(ka.mp, xk[3], xk[5], xk[7], h, g3, g5, g7);
        r[PR_GH + 0] = g3; r[PR_GH + 1] = g5; r[PR_GH + 2] = g7; r[PR_GH + 3] = h;
    }
    if (k < N && (!SN || k >= ka.uph)) {
        const double *gU = ka.U + ((size_t)b * N + k) * NU;
        double uk[2] = {gU[0], gU[1]};
        double xn[8], Sp[2], S[6][7];
        rk4_sens(ka.mp, xk, uk, ka.dt, ka.nsub, xn, Sp, S);
        r[0] = Sp[0]; r[1] = Sp[1];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int c = 0; c < 7; c++) r[2 + i * 7 + c] = S[i][c];
#pragma unroll
        for (int i = 0; i < 8; i++) r[44 + i] = xn[i] - gX[NX + i];
        // (get_from_qp_in and the R2 back-off read A_k, B_k, b_k of a pipeline solve from these records: no separate copy)
    }
    constexpr int NF = SN ? PREC : PR_XD + 1;          // fields in use
#pragma unroll
    for (int i = 0; i < NF; i++) sT[threadIdx.x * L_PITCH + i] = r[i];
    wsync();
    double *dst = pa.rec + (size_t)g0 * PREC;
    const int nitem = (int)((total - g0 < 64) ? (total - g0) : 64);
    for (int it = 0; it < nitem; it++)
        if ((int)threadIdx.x < NF) dst[(size_t)it * PREC + threadIdx.x] = sT[it * L_PITCH + threadIdx.x];
}

// ---- K1 for small batches: EIGHT lanes per (instance, stage).
// lin_kernel is one pass of ~13 000 instructions per wavefront whatever the batch: with a few dozen instances that pass IS
// the kernel's time (33 us at 26 instances). Here the work of an item is spread over a group of eight lanes so that the pass is
// shorter, every number still produced by the same operations on the same operands:
//   - the eight columns of the sensitivity recursion (seven of S, the psi0 column Sp) are independent given the state
//     trajectory: lane c carries column c through the twelve Runge-Kutta stages (one column update per stage instead of eight);
//   - inside each DPP quad the transcendental chains of the model are split as in plant_xdot (loop_kernels.hpp): lane 0 the
//     front tyre (slip angle and its partials, magic formula with derivative, friction-circle factor), lane 1 the rear tyre,
//     lane 2 sin / cos of psi, lane 3 sin / cos of delta; ONE sincos serves all four (the magic formula's on lanes 0, 1). The
//     tyre lanes reduce their results to the five numbers the Jacobian needs (Fy and its partials) and the quad exchanges
//     them by DPP broadcasts; the remainder (rolling resistance, Jacobian assembly, the state's RK4 update) is computed
//     redundantly by every lane.
// The state recursion comes out bit-identical to lin_kernel's (b_k); the sensitivity columns agree to <= 3e-15 relative, not
// to the bit: lin_kernel's column loop is unrolled with the column index known, so a factor 1.0 folds away and the product
// behind it is contracted into an FMA there and not here (tests/test_gpu_parity.py::test_linearisation_eight_lanes_...).
// Launched when batch x (N + 1) x 8 lanes still fit one round of the chip (tum_nmpc.hip: launch_pipeline).
constexpr int LC_LANES = 8, LC_ITEMS = 64 / LC_LANES;
template <bool SN>
__global__ void __launch_bounds__(64, 1) lin_cols_kernel(const PArgs pa)
{
    __shared__ double sT[LC_ITEMS * L_PITCH];
    const KArgs &ka = pa.ka;
    const int N = ka.N, NB = N + 1;
    const long long total = (long long)ka.batch * NB;
    const long long g0 = (long long)blockIdx.x * LC_ITEMS;
    const int li = threadIdx.x / LC_LANES, col = threadIdx.x % LC_LANES;
    const long long gl = g0 + li;
    const long long g = (gl < total) ? gl : total - 1;       // (groups beyond the last item shadow it; their records are not stored)
    const int b = (int)(g / NB), k = (int)(g - (long long)b * NB);
    const TyreLane t = tyre_lane(ka.mp, col);
    const double *gX = ka.X + ((size_t)b * NB + k) * NX;
    double xk[8];
#pragma unroll
    for (int i = 0; i < 8; i++) xk[i] = gX[i];
    // flags & 8 (the device closed loop, nominal OCP): this launch runs BESIDE the planner that writes yref for this solve -- the
    // residuals of the cost are left to cond_wide_kernel, which forms them while it loads the records, and yref is not touched
    const bool late_res = !SN && (ka.flags & 8);
    const double *yr = ka.yref + ((size_t)b * NB + k) * 6;
    double *rw = sT + li * L_PITCH;
    // the fields outside the sensitivity block are computed by every lane of the group (same loads, same operations) and
    // stored by lane 0
    double res0 = 0.0, res1 = 0.0, res2 = 0.0, res3 = 0.0, cv0 = 0.0, cv1 = 0.0;
    if (!late_res) { res0 = xk[0] - yr[0]; res1 = xk[1] - yr[1]; res2 = wrap_yaw(xk[2]) - yr[2]; res3 = xk[3] - yr[3]; }
    if (SN) {           // the speed row |v| of the coupled SNMPC OCP and its gradient (lin_kernel<true>)
        const double vabs = sqrt(xk[3] * xk[3] + xk[4] * xk[4]), iv = (vabs > 0.0) ? 1.0 / vabs : 0.0;
        res3 = vabs - yr[3];
        cv0 = xk[3] * iv; cv1 = xk[4] * iv;
    }
    double h = 0.0, g3 = 0.0, g4 = 0.0, g5 = 0.0, g7 = 0.0;
    if (k >= 1) {
        if (SN) h_con_vabs(ka.mp, xk[3], xk[4], xk[5], xk[7], h, g3, g4, g5, g7);
        else h_con(ka.mp, xk[3], xk[5], xk[7], h, g3, g5, g7);
    }
    if (col == 0) {
        rw[PR_RES + 0] = res0; rw[PR_RES + 1] = res1; rw[PR_RES + 2] = res2; rw[PR_RES + 3] = res3;
        rw[PR_GH + 0] = g3; rw[PR_GH + 1] = g5; rw[PR_GH + 2] = g7; rw[PR_GH + 3] = h;
        rw[PR_XD] = xk[6];
        if (SN) { rw[PR_CV] = cv0; rw[PR_CV + 1] = cv1; rw[PR_G4] = g4; }
    }
    if (k < N && (!SN || k >= ka.uph)) {        // (uniform over the eight lanes of an item; the DPP exchanges stay inside a quad)
        const double *gU = ka.U + ((size_t)b * N + k) * NU;
        double uk[2] = {gU[0], gU[1]};
        double xn[8], Sc[6];
        rk4_sens_col(ka.mp, t, col, xk, uk, ka.dt, ka.nsub, xn, Sc);
        if (col == 7) { rw[0] = Sc[0]; rw[1] = Sc[1]; }
        else {
#pragma unroll
            for (int i = 0; i < 6; i++) rw[2 + i * 7 + col] = Sc[i];
        }
        double dn = xn[0] - gX[NX + 0];
#pragma unroll
        for (int i = 1; i < 8; i++) dn = (col == i) ? xn[i] - gX[NX + i] : dn;
        rw[44 + col] = dn;
    } else {
        for (int i = col; i < PR_RES; i += LC_LANES) rw[i] = 0.0;
    }
    wsync();
    constexpr int NF = SN ? PREC : PR_XD + 1;          // fields in use
    double *dst = pa.rec + (size_t)g0 * PREC;
    const int nitem = (int)((total - g0 < LC_ITEMS) ? (total - g0) : LC_ITEMS);
    for (int it = 0; it < nitem; it++)
        if ((int)threadIdx.x < NF) dst[(size_t)it * PREC + threadIdx.x] = sT[it * L_PITCH + threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------------- K2
// (Round 5 also built the stage record through the SCALAR data path -- address space 4, s_load, the fields as scalar operands of the FMAs:
//  +38 us, every stage waits for its loads -- and three wavefronts per SIMD: slower in every form; profiles/r05_ab_cond_variants.txt,
//  HISTORY.md. What replaced the LDS broadcasts is the DPP form below.)
template <typename R>
__device__ __forceinline__ void apply_A_rec(const R &rec, double w[8])
{
    double n[6];
    n[0] = w[0] + rec[0] * w[2];
    n[1] = w[1] + rec[1] * w[2];
    n[2] = w[2];
    n[3] = 0.0; n[4] = 0.0; n[5] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int c = 0; c < 5; c++) n[i] += rec[2 + i * 7 + c] * w[3 + c];
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = n[i];
}
template <typename R>
__device__ __forceinline__ void apply_A2_rec(const R &rec, double w[8], double v[8])
{
    const double sp0 = rec[0], sp1 = rec[1];
    double n[6], m[6];
    n[0] = w[0] + sp0 * w[2]; m[0] = v[0] + sp0 * v[2];
    n[1] = w[1] + sp1 * w[2]; m[1] = v[1] + sp1 * v[2];
    n[2] = w[2]; m[2] = v[2];
    n[3] = 0.0; n[4] = 0.0; n[5] = 0.0; m[3] = 0.0; m[4] = 0.0; m[5] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int c = 0; c < 5; c++) { const double sv = rec[2 + i * 7 + c]; n[i] += sv * w[3 + c]; m[i] += sv * v[3 + c]; }
#pragma unroll
    for (int i = 0; i < 6; i++) { w[i] = n[i]; v[i] = m[i]; }
}

// The stage record in REGISTERS, read by the 64-bit DPP broadcast (round 5). A record is wave-uniform data; rounds 2-5 staged it in LDS
// and broadcast it back with ~25 ds_read per stage. gfx90a+ vector FP64 takes ONE DPP control, row_newbcast:n -- every lane of a 16-lane
// row reads lane n of its row (probed: scripts/probes/probe_dpp64_bcast.cpp, same issue cost as a plain v_fma_f64) -- on v_fmac_f64 only.
// So the 64 fields of a stage slot live in four registers, field 16 r + n on lane n of EVERY row of register r (four coalesced 128-byte
// loads per stage, straight from the workspace, two stages ahead in two register sets), and every use of a field is `acc += field * x`
// with the field as the DPP operand: no LDS read, no scalar register, no extra instruction. The sums are formed in the order the
// compiler's contraction of the LDS form forms them (x * 1.0 and + 0.0 are exact), so the results are the same to the bit (the tests
// hold this kernel against cond_wide_kernel, which kept the LDS form). With 30 registers fewer all 15 Hessian tiles stay in registers
// (no tile parked in LDS), lane-derived values are kept instead of re-derived per stage, a lane's input column is fetched once, and the
// weights are scaled by dt once, in LDS, where a lane reads the one of its operand row (no select chain).
// (Measured and dropped, profiles/r05_ab_cond_dpp.txt and HISTORY.md: the four cost rows to the MFMA operand layout by gfx950 row swaps
// instead of through LDS, and the first rows of g_s by v_readlane -- a stage without any LDS traffic, bit-identical, 1-2 us slower.)
struct RecRows {
    double g[4];
    // acc += field F * x
    template <int F> __device__ __forceinline__ void fmac(double &acc, double x) const
    {
        static_assert(F >= 0 && F < 64, "record field");
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g[F >> 4]), "v"(x), "n"(F & 15));
    }
    // (a DPP operand must not be read within two wait states of a vector instruction that wrote it -- the compiler does not look into
    //  the asm above: one s_nop behind whatever produced the registers, which every later use depends on)
    __device__ __forceinline__ void settle() { asm volatile("s_nop 1" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3])); }
};
// ---- lane-distributed 4 x 4 pivot blocks of the interior point kernel's micro-panels (round 6): a value that lives on ONE lane of every
// 16-lane row is consumed as the `row_newbcast` operand of an FP64 FMA / reciprocal -- no wave-uniform copy of it, no lane select
// acc -= (lane L of the row of b) * x
template <int L> __device__ __forceinline__ void fnmac_bc(double &acc, double b, double x)
{
    static_assert(L >= 0 && L < 16, "lane of the row");
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(x), "n"(L));
}
// acc -= (lane L of the row of acc) * x
// SETTLE: two wait states behind it. The compiler's hazard recogniser does not know that the asm is a vector instruction: a MATRIX instruction
// that takes acc as an operand right behind it read a stale register (found at N = 48: the last micro-panel of the last block column hands its
// X straight to a 4x4x4 product -- 3e-4 on the last stage's jerk, everything else exact). Results that go to plain vector instructions first
// need nothing: the hardware interlocks those.
// FRESH: acc comes straight from a plain vector instruction: two wait states in FRONT of the DPP read (as mul_bc_fresh below)
template <int L, bool SETTLE = false, bool FRESH = false> __device__ __forceinline__ void fnmac_bc_self(double &acc, double x)
{
    if constexpr (FRESH) asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "n"(L));
    else if constexpr (SETTLE) asm("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(acc) : "v"(x), "n"(L));
    else asm("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "n"(L));
}
// lm = (lane L of the row of r) * (a * m)
// (v_rcp_f64_dpp assembles but does not work: scripts/probes/probe_dpp_f64.cpp returns inf -- the reciprocal of a pivot is taken on every lane
//  and its lane reaches the others through this FMA.) r has JUST been produced by a vector instruction (the Newton step of the reciprocal): a DPP
// read needs two wait states behind the write of its source, which the compiler does not know it owes to the asm -- the product a * m and the zero of
// the accumulator stand there (while matrix instructions sat between the reciprocal and its use -- rounds 2-6a -- the distance was there by accident;
// the disassembly test of tests/test_host_logic.py caught it when they left)
template <int L> __device__ __forceinline__ double mul_bc_fresh(double r, double a, double m)
{
    double lm, am;
    asm("v_mul_f64 %1, %3, %4\n\tv_mov_b64 %0, 0\n\tv_fmac_f64_dpp %0, %2, %1 row_newbcast:%5 row_mask:0xf bank_mask:0xf" : "=&v"(lm), "=&v"(am) : "v"(r), "v"(a), "v"(m), "n"(L));
    return lm;
}
// w <- A_k w (+ the same for a second bank v): apply_A's sums, term by term in its order
// (the rows 0..2 -- px, py, psi -- are no operands of any row: updated in place, row 2 behind the two rows that read it; only the
//  rows 3..5 need fresh accumulators. The instruction is destructive -- acc is source and destination --, every copy is a vector move)
// INS: the rows 3..5 of bank w start from ins[0..2] (the lane's input column times its stage selector: a product that is zero wherever
// the sum behind it is not, so the order of the two does not matter) instead of from zero -- three vector moves and three FMAs less
template <bool TWO, bool INS = false>
__device__ __forceinline__ void apply_A_rows(const RecRows &R, double w[8], double v[8], const double *ins = nullptr)
{
    double n[3] = {0.0, 0.0, 0.0}, m[3] = {0.0, 0.0, 0.0};
    if constexpr (INS) { n[0] = ins[0]; n[1] = ins[1]; n[2] = ins[2]; }
    R.fmac<0>(w[0], w[2]); R.fmac<1>(w[1], w[2]);
    if constexpr (TWO) { R.fmac<0>(v[0], v[2]); R.fmac<1>(v[1], v[2]); }
    static_for<0, 5>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, 4>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            R.fmac<2 + i * 7 + c>(i < 3 ? w[i] : n[i - 3], w[3 + c]);
            if constexpr (TWO) R.fmac<2 + i * 7 + c>(i < 3 ? v[i] : m[i - 3], v[3 + c]);
        });
    });
#pragma unroll
    for (int i = 0; i < 3; i++) { w[3 + i] = n[i]; if constexpr (TWO) v[3 + i] = m[i]; }
}

// LDS of the condensing kernel (PD::C_*): stage record double buffer | 4 staging rows of the SYRK | g_s of the current stage |
// iterate U | packed gg rows (staging for the operand layout)
// REGF: the register form of the stage record (RecRows); the LDS form is kept for the coupled SNMPC OCP at long propagation horizons
// (the host decides, tum_nmpc.hip: launch_pipeline)
template <int NT_, bool SN, bool REGF = !SN>
__global__ void __launch_bounds__(64, (NT_ == 5) ? 2 : 1) cond_kernel(const PArgs pa)
{
    PD_LOCALS
    constexpr int C_REC = D::C_REC, C_STAGE = D::C_STAGE, C_GS = D::C_GS, C_U0 = D::C_U0, C_WT = D::C_WT;
    __shared__ __attribute__((aligned(16))) double lds[D::C_LDS];
    const KArgs &ka = pa.ka;
    const int lane = threadIdx.x, b = blockIdx.x;
    if (b >= ka.batch) return;
    const int N = ka.N, nv = 2 * N;
    const double dt = ka.dt;
    double *sRec = lds + C_REC, *sStage = lds + C_STAGE, *sGs = lds + C_GS, *sU0 = lds + C_U0, *sWt = lds + C_WT;
    const double *grec = pa.rec + (size_t)b * (N + 1) * PREC;
    const double *gx0 = ka.x0 + (size_t)b * NX;
    const double *gX = ka.X + (size_t)b * (N + 1) * NX;
    const double *gU = ka.U + (size_t)b * N * NU;
    const double *gyref = ka.yref + (size_t)b * (N + 1) * 6;
    const double *gW = ka.W + (size_t)b * (N + 1) * 6;      // diagonal of W per stage (cost_set(i, 'W', ...), NMPC_class.py:294-296)
    double *gvec = pa.vec + (size_t)b * PVEC;
    // the weights as the sums take them: dt W_s of the stages s < N, W_e of stage N (one product per weight, here instead of in every stage)
    for (int i = lane; i < (N + 1) * 6; i += 64) sWt[i] = ((i < 6 * N) ? dt : 1.0) * gW[i];

    // The nominal OCP reads its stage records from REGISTERS by DPP broadcasts (RecRows above). The coupled SNMPC OCP takes the columns
    // of its first uph stages from the prologue's buffer and uses nine fields of a record there: at UPH = Tp the register form costs it
    // 6 % (four loads a stage instead of one), behind stage uph it gains what the nominal OCP gains -- REGF is the host's choice by uph
    // Hessian tiles of the last block column parked in LDS between the stages of the last segment: none in the register form (it needs
    // 30 registers fewer), D::C_NPARK in the LDS form
    constexpr int NPARK = ((REGF && !SN) || NT_ != 5) ? 0 : D::C_NPARK;
    // stage slot k & 1: fields 0..51 of record k (A_k, B_k, b_k), fields 52..60 of record k+1 (residuals, gg row, delta of stage k+1)
    auto fetch = [&](int k) -> double { return (lane < PR_RES) ? grec[(size_t)k * PREC + lane] : ((lane <= (SN ? PR_G4 : PR_XD)) ? grec[(size_t)(k + 1) * PREC + lane] : 0.0); };
    // field 16 r + (lane & 15) of the stage slot k, and the weight of cost row lane >> 4 of stage k + 1
    auto fetch_rows = [&](int k, RecRows &R) {
        const int p_ = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = 16 * r + p_;
            R.g[r] = (f < PR_RES) ? grec[(size_t)k * PREC + f] : ((f <= (SN ? PR_G4 : PR_XD)) ? grec[(size_t)(k + 1) * PREC + f] : 0.0);
        }
    };
    const int uph = SN ? ka.uph : 0;
    const int PP = SN ? sn_pro_pitch(uph) : 64, PSTAGE = 9 * PP;
    const double *gpro = SN ? ka.pro + (size_t)b * uph * PSTAGE : nullptr;
    sU0[lane] = (lane < nv) ? gU[lane] : 0.0;
    if (lane < NB1) sU0[64 + lane] = (64 + lane < nv) ? gU[64 + lane] : 0.0;
    // register form: two register sets, the even stages' and the odd stages': a stage reads its own and, done with it, requests the
    // record two stages on into it (a rotation of the sets would be ten vector moves per stage, on the issue port this kernel is bound by)
    RecRows Ra{}, Rb{};
    double pre = 0.0;
    // register form: the input column of a lane -- column (lane & 1) of B_j, j the stage of the lane's variable -- enters the recursion
    // once, at stage j: fetched once per bank (bank 1 in front of its segment) instead of broadcast to every lane in every stage
    double bcol[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    auto fetch_bcol = [&](int j) {
#pragma unroll
        for (int i = 0; i < 6; i++) bcol[i] = (j < N) ? grec[(size_t)j * PREC + 2 + i * 7 + 5 + (lane & 1)] : 0.0;
    };
    const double notg63 = (lane == 63) ? 0.0 : 1.0;
    const double b6c = (lane & 1) ? dt : 0.0, b7c = (lane & 1) ? 0.0 : dt;          // (rows 6, 7 of that column: the integrators of the two inputs)
    if constexpr (REGF) {
        fetch_rows(0, Ra);
        if (N > 1) fetch_rows(1, Rb); else Rb = Ra;
        fetch_bcol(lane >> 1);
    } else {
        pre = fetch(0);
        sRec[lane] = pre;
        if (N > 1) pre = fetch(1);
    }
    wsync();

    const int j0 = lane >> 1, r0 = lane & 1, j1 = 32 + (lane >> 1);
    const bool isg = (lane == NB1);              // bank 1: columns 64.. on the lanes 0..NB1-1, the constant column g on lane NB1
    const int lq = lane >> 4, lc = lane & 15;
    double q0 = 0.0, q1 = 0.0;
    d4 Ht[NTT];
#pragma unroll
    for (int i = 0; i < NTT; i++) Ht[i] = d4{0.0, 0.0, 0.0, 0.0};
    if constexpr (NT_ == 5) {
#pragma unroll
        for (int K = 0; K < NPARK; K++) reinterpret_cast<d4 *>(lds + D::C_PARK)[K * 64 + lane] = d4{0.0, 0.0, 0.0, 0.0};
    }
    {
        double w0[8], w1[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { w0[i] = 0.0; w1[i] = 0.0; }
        // The constant column g of the recursion rides in bank 0 -- on lane 63 -- while the stages leave that lane free (stage s has 2 s
        // columns: s <= 24, the first three segments) and moves to its lane of bank 1 in front of the fourth: until then bank 1 holds
        // nothing, and the second copy of the column update, the gg row, the staging stores and the gradient terms of bank 1 are not
        // issued at all (55 of the ~130 FP64 instructions of a stage; the same operations on the same operands for every column that
        // exists -- results unchanged to the bit, tests/test_gpu_parity.py::test_condensing_six_wavefronts_is_the_same_arithmetic).
        // (The coupled SNMPC OCP loads its first stages from the prologue's buffer in the two-bank layout and keeps it.)
        constexpr int G0_SEGS = SN ? 0 : 3;
        if (G0_SEGS > 0 ? lane == 63 : isg) {
#pragma unroll
            for (int i = 0; i < 8; i++) (G0_SEGS > 0 ? w0[i] : w1[i]) = gx0[i] - gX[i];
        }
        const int lane_cond = lane;
        // coupled SNMPC OCP: the columns G_nom,s / g_nom,s of a stage s <= uph and its chance-constraint row come from the prologue's
        // buffer -- requested ONE STAGE AHEAD (18 doubles per lane): fetched where they are used, every one of the uph stages waited a
        // memory round trip for them
        double nP0[8], nP1[8], nH0 = 0.0, nH1 = 0.0;
        auto fetch_pro = [&](const int k, const int ln) {
            const double *pg = gpro + (size_t)k * PSTAGE;
            const bool b0 = ln < 2 * (k + 1), b1 = ln < NB1 && 64 + ln < 2 * (k + 1), isg_ = (ln == NB1);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // (only the 2 (k + 1) live columns of the stage are fetched: the rest of a row is zero, and at UPH = Tp the full rows
                //  were 200 KB of the 350 KB this kernel read per instance)
                const double gv = b0 ? pg[i * PP + ln] : 0.0, gg = pg[i * PP + 2 * uph];
                const double g1 = b1 ? pg[i * PP + 64 + ln] : 0.0;
                nP0[i] = gv;
                nP1[i] = isg_ ? gg : g1;
            }
            if (k + 1 < uph) {   // chance-constraint row E + kappa sqrt(Var) over the samples of stage s = k + 1
                const double *pr = pg + 8 * PP;
                const int s_ = k + 1;
                const double rv = (ln < 2 * s_) ? pr[ln] : 0.0, rg = pr[2 * uph];
                const double r1 = (ln < NB1 && 64 + ln < 2 * s_) ? pr[64 + ln] : 0.0;
                nH0 = rv; nH1 = isg_ ? rg : r1;
            }
        };
        // (the LDS form, which is the one for long propagation horizons; the register form has no 36 registers to spare and few such stages)
        constexpr bool PRO_AHEAD = SN && !REGF;
        if constexpr (PRO_AHEAD) { if (uph > 0) fetch_pro(0, lane); }
        // (two call sites per segment: inlined by force, or every captured array lives in scratch)
        auto stage_body = [&](const int k, auto tsc, auto g0c, RecRows &R) __attribute__((always_inline)) {
            constexpr int Ts = decltype(tsc)::value;
            // G0: the constant column rides on lane 63 of bank 0 and bank 1 is not issued. LATE: ... in a segment whose tiles reach
            // column 63 (the register form keeps g there through stage 31, whose input column is the first to need the lane): what
            // lane 63 stages and stores as its share of the gg row is masked to the zeros of the column that is not there yet
            constexpr bool G0 = decltype(g0c)::value, LATE = G0 && Ts > G0_SEGS;
            // (everything derived from the lane id is derived again in every stage, from a copy the optimiser cannot see through:
            //  held across the stage loop these values cost the registers the last segment lacks)
            int lane_s = lane_cond;
            if constexpr (!REGF || SN) asm volatile("" : "+v"(lane_s));          // (the SNMPC LDS form with them kept: 246 registers, no change in time -- its stages wait for the prologue's buffer)          // (the register form has ~40 registers to spare: there the optimiser may keep what it likes -- 45 of 270 vector instructions per pair of stages)
            const int lane = lane_s;
            const int j0 = lane >> 1, r0 = lane & 1, j1 = 32 + (lane >> 1);
            const bool isg = (lane == NB1);
            const int lq = lane >> 4, lc = lane & 15;
            const double *rec = sRec + (k & 1) * PREC;          // (LDS form)
            const double one = 1.0;
            if constexpr (REGF) R.settle();
            double cr0 = 0.0, cr1 = 0.0;          // (SNMPC: the chance-constraint row of this stage)
            if (SN && k < uph) {
                // stage s = k+1 <= uph: G_nom,s and g_nom,s are PCE means of the sample recursions (prologue kernel)
                // (columns 0..63 in bank 0, 64..2 uph-1 on the lanes of bank 1, the constant column on the g lane)
                if constexpr (!PRO_AHEAD) fetch_pro(k, lane);
#pragma unroll
                for (int i = 0; i < 8; i++) { w0[i] = nP0[i]; w1[i] = nP1[i]; }
                cr0 = nH0; cr1 = nH1;
                if constexpr (PRO_AHEAD) { if (k + 1 < uph) fetch_pro(k + 1, lane); }
            } else if constexpr (REGF) {
                // (the input column of a lane: B_k's column r0 -- two accumulations, the one of the other parity adds 0.0 * field)
                if constexpr (G0) {
                    const double sel0 = (j0 == k) ? 1.0 : 0.0, selg = (lane == 63) ? 1.0 : 0.0;
                    const double ins[3] = {sel0 * bcol[3], sel0 * bcol[4], sel0 * bcol[5]};
                    apply_A_rows<false, true>(R, w0, w0, ins);
#pragma unroll
                    for (int i = 0; i < 3; i++) w0[i] += sel0 * bcol[i];
                    w0[6] += sel0 * b6c; w0[7] += sel0 * b7c;
                    static_for<0, 7>([&](auto ic) { constexpr int i = decltype(ic)::value; R.fmac<44 + i>(w0[i], selg); });
                } else {
                    apply_A_rows<true>(R, w0, w1);
                    // (the input columns of stage k sit in bank 0 while k < 32 -- the segments 1..4 -- and in bank 1 behind: the other bank's
                    //  accumulations would add 0.0 * entry)
                    constexpr bool IN0 = Ts <= 4;
                    const double selk = IN0 ? ((j0 == k) ? 1.0 : 0.0) : ((lane < NB1 && j1 == k) ? 1.0 : 0.0), selg = isg ? 1.0 : 0.0;
                    double *wi = IN0 ? w0 : w1;
#pragma unroll
                    for (int i = 0; i < 6; i++) wi[i] += selk * bcol[i];
                    wi[6] += selk * b6c; wi[7] += selk * b7c;
                    static_for<0, 7>([&](auto ic) { constexpr int i = decltype(ic)::value; R.fmac<44 + i>(w1[i], selg); });
                }
            } else {
                if constexpr (G0) {
                    apply_A_rec(rec, w0);
                    const double sel0 = (j0 == k) ? 1.0 : 0.0, selg = (lane == 63) ? 1.0 : 0.0;
#pragma unroll
                    for (int i = 0; i < 6; i++) w0[i] += sel0 * rec[2 + i * 7 + 5 + r0];
                    w0[6] += sel0 * (r0 ? dt : 0.0); w0[7] += sel0 * (r0 ? 0.0 : dt);
#pragma unroll
                    for (int i = 0; i < 8; i++) w0[i] += selg * rec[44 + i];
                } else {
                    apply_A2_rec(rec, w0, w1);
                    const double sel0 = (j0 == k) ? 1.0 : 0.0, sel1 = (lane < NB1 && j1 == k) ? 1.0 : 0.0, selg = isg ? 1.0 : 0.0;
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        const double bc = rec[2 + i * 7 + 5 + r0];
                        w0[i] += sel0 * bc; w1[i] += sel1 * bc;
                    }
                    const double b6 = r0 ? dt : 0.0, b7 = r0 ? 0.0 : dt;
                    w0[6] += sel0 * b6; w0[7] += sel0 * b7; w1[6] += sel1 * b6; w1[7] += sel1 * b7;
#pragma unroll
                    for (int i = 0; i < 8; i++) w1[i] += selg * rec[44 + i];
                }
            }
            const int s = k + 1;                         // stage whose G_s the lanes now hold
            double hr0 = 0.0, hr1 = 0.0, c30, c31;          // gg row and speed row of stage s (per bank)
            double cvl = 1.0, cvt = 0.0;                     // (LDS form, SNMPC: gradient of |v|)
            if constexpr (REGF) {
                // (g3 w3 + g5 w5 + g7 w7 as the compiler contracts the LDS form's expression: the SECOND product is rounded, the first and
                //  the third are fused onto it)
                R.fmac<PR_GH + 1>(hr0, w0[5]); R.fmac<PR_GH + 0>(hr0, w0[3]); R.fmac<PR_GH + 2>(hr0, w0[7]);
                if constexpr (!G0) { R.fmac<PR_GH + 1>(hr1, w1[5]); R.fmac<PR_GH + 0>(hr1, w1[3]); R.fmac<PR_GH + 2>(hr1, w1[7]); }
                double hdon = 1.0;          // (the constant of the gg row, field PR_GH + 3: dropped where the chance-constraint row stands in)
                c30 = w0[3]; c31 = w1[3];          // the speed row of the cost: vl (nominal OCP) ...
                if constexpr (SN) {
                    R.fmac<PR_G4>(hr0, w0[4]); R.fmac<PR_G4>(hr1, w1[4]);
                    if (s < uph) { hr0 = cr0; hr1 = cr1; hdon = 0.0; }   // chance-constraint row E + kappa sqrt(Var) over the samples (prologue kernel)
                    // ... or |v| (SNMPC: gradient (vl, vt)/|v|; the second product rounded, the first fused onto it)
                    c30 = 0.0; c31 = 0.0;
                    R.fmac<PR_CV + 1>(c30, w0[4]); R.fmac<PR_CV>(c30, w0[3]);
                    R.fmac<PR_CV + 1>(c31, w1[4]); R.fmac<PR_CV>(c31, w1[3]);
                }
                // d of the two rows of stage s on the lane that holds the constant column: field + column entry
                double dbx = G0 ? w0[6] : w1[6], dh = G0 ? hr0 : hr1;
                R.fmac<PR_XD>(dbx, one); R.fmac<PR_GH + 3>(dh, hdon);
                if (G0 ? lane == 63 : isg) {
                    {
#pragma unroll
                        for (int i = 0; i < (SN ? 5 : 4); i++) sGs[i] = G0 ? w0[i] : w1[i];
                    }
                    gvec[PV_D + 2 * (s - 1)] = dbx;
                    gvec[PV_D + 2 * (s - 1) + 1] = dh;
                }
            } else {
                const double g3 = rec[PR_GH + 0], g5 = rec[PR_GH + 1], g7 = rec[PR_GH + 2];
                const double g4 = SN ? rec[PR_G4] : 0.0;
                if (SN) { cvl = rec[PR_CV]; cvt = rec[PR_CV + 1]; }
                hr0 = g3 * w0[3] + g5 * w0[5] + g7 * w0[7]; hr1 = G0 ? 0.0 : g3 * w1[3] + g5 * w1[5] + g7 * w1[7];
                double hd = rec[PR_GH + 3];
                if (SN) {
                    hr0 += g4 * w0[4]; hr1 += g4 * w1[4];
                    if (s < uph) { hr0 = cr0; hr1 = cr1; hd = 0.0; }   // chance-constraint row E + kappa sqrt(Var) over the samples (prologue kernel)
                }
                // the speed row of the cost: vl (nominal OCP) or |v| (SNMPC: gradient (vl, vt)/|v|)
                c30 = SN ? cvl * w0[3] + cvt * w0[4] : w0[3]; c31 = SN ? cvl * w1[3] + cvt * w1[4] : w1[3];
                if (G0 ? lane == 63 : isg) {
#pragma unroll
                    for (int i = 0; i < 8; i++) sGs[i] = G0 ? w0[i] : w1[i];
                    gvec[PV_D + 2 * (s - 1)] = rec[PR_XD] + (G0 ? w0[6] : w1[6]);
                    gvec[PV_D + 2 * (s - 1) + 1] = hd + (G0 ? hr0 : hr1);
                }
            }
            // the gg row of stage s goes straight to the workspace in the MFMA operand layout the interior point kernel loads it in:
            // row s = 4 c + lq + 1 of chunk c, the 16 columns of tile T at cidx(c, T) -- one store per bank; the columns right of the
            // row's end hold zeros already. (Rounds 2-3 staged all rows in LDS, 13 KB, and re-laid them out behind the last stage.)
            {
                const int c_ = (s - 1) >> 2;
                double *gcs = pa.cws + (size_t)b * NCH * 64 + 16 * ((s - 1) & 3) + lc;
                if (2 * lq <= c_) gcs[cidx(c_, lq) * 64] = LATE ? hr0 * notg63 : hr0;          // (G0: the lanes 48..63 do not store before stage 25)
                const int T1 = 4 + lq;
                if (!G0 && lane < NB1 && 2 * T1 <= c_) gcs[cidx(c_, T1) * 64] = hr1;
            }
            // the four cost rows to the MFMA operand layout, the stage's weights, the first rows of g_s
            {
#pragma unroll
                for (int r = 0; r < 3; r++) sStage[r * NVP + lane] = LATE ? w0[r] * notg63 : w0[r];
                sStage[3 * NVP + lane] = LATE ? c30 * notg63 : c30;          // (G0: what lane 63 stages is g, in a column the tiles of the first three segments do not read)
                if (!G0 && lane < NB1) {
#pragma unroll
                    for (int r = 0; r < 3; r++) sStage[r * NVP + 64 + lane] = w1[r];
                    sStage[3 * NVP + 64 + lane] = c31;
                }
            }
            wsync();
            double wl, wr[4], gsr[5];
#pragma unroll
            for (int r = 0; r < 4; r++) wr[r] = sWt[s * 6 + r];          // (the stage's own weights, scaled; stage N: W_e)
            if constexpr (REGF) wl = sWt[s * 6 + lq];                    // (an LDS read instead of a select chain)
            else wl = (lq == 0) ? wr[0] : (lq == 1) ? wr[1] : (lq == 2) ? wr[2] : wr[3];
#pragma unroll
            for (int r = 0; r < (SN ? 5 : 4); r++) gsr[r] = sGs[r];
            {
                double a0 = 0.0, a1 = 0.0;
                static_for<0, 3>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    double gs = gsr[r];
                    if constexpr (REGF) {
                        if constexpr (SN && r == 3) { gs = 0.0; R.fmac<PR_CV + 1>(gs, gsr[4]); R.fmac<PR_CV>(gs, gsr[3]); }
                        R.fmac<PR_RES + r>(gs, one);          // residual + g_s
                    } else gs = rec[PR_RES + r] + ((SN && r == 3) ? cvl * gsr[3] + cvt * gsr[4] : gsr[r]);
                    const double e = wr[r] * gs;
                    a0 += e * ((r == 3) ? c30 : w0[r]);
                    if constexpr (!G0) a1 += e * ((r == 3) ? c31 : w1[r]);
                });
                if constexpr (REGF && G0) q0 += a0 * notg63;          // (lane 63 holds g, not a column)
                else q0 += (G0 && lane == 63) ? 0.0 : a0;
                if constexpr (!G0) q1 += (lane < NB1) ? a1 : 0.0;
            }
            double bop[Ts];
#pragma unroll
            for (int T = 0; T < Ts; T++) bop[T] = sStage[lq * NVP + 16 * T + lc];
            // Last segment of the five-tile build when tiles are parked (the LDS form; rounds 2-5 until the stage record left the
            // register file's competitors): the first C_NPARK tiles of the last block column -- touched by this segment only -- wait in
            // LDS between the stages: loaded, updated, stored again.
            constexpr bool PARK = (NT_ == 5) && (Ts == NT_);
            d4 *sPark = reinterpret_cast<d4 *>(lds + D::C_PARK) + lane;
            // the weighted operands of the stage: all of them AHEAD of the matrix instructions (the nominal OCP: one change between the vector and the
            // matrix pipe per stage instead of one per tile row, -2 us on 4096 instances, bit-identical -- round 6, profiles/r06_ab_mp_slots.txt); the
            // coupled SNMPC OCP's instantiations have no registers for the five of them and form one at a time
            constexpr bool AOPV = !SN;
            double aopv[AOPV ? Ts : 1];
            if constexpr (AOPV) {
#pragma unroll
                for (int K = 0; K < Ts; K++) aopv[K] = bop[K] * wl;
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int K = 0; K < Ts; K++) {
                double aop;
                if constexpr (AOPV) aop = aopv[K]; else aop = bop[K] * wl;
#pragma unroll
                for (int I = K; I < Ts; I++) {
                    if constexpr (PARK) {
                        if (I == NT_ - 1 && K < NPARK) { sPark[K * 64] = mfma(aop, bop[I], sPark[K * 64]); continue; }
                    }
                    Ht[tidx(K, I)] = mfma(aop, bop[I], Ht[tidx(K, I)]);
                }
            }
            if constexpr (REGF) {
                // the record two stages on, into the register set this stage is done with
                if (k + 2 < N) fetch_rows(k + 2, R);
            } else {
                // next stage's record into the other slot (its global load has been in flight for a whole stage)
                sRec[((k + 1) & 1) * PREC + lane] = pre;
                if (k + 2 < N) pre = fetch(k + 2);
            }
            wsync();
        };
        // stage s = k+1 touches columns < 2s, i.e. ceil(s/8) tiles: one instantiation of the stage per segment of 8 stages
        static_for<1, NT>([&](auto tsc) {
            constexpr int Ts = decltype(tsc)::value;
            constexpr bool G0S = Ts <= G0_SEGS;
            auto move_g = [&] {          // g moves from lane 63 of bank 0 to its lane of bank 1
#pragma unroll
                for (int i = 0; i < 8; i++) { const double gv = rl(w0[i], 63); w1[i] = isg ? gv : 0.0; w0[i] = (lane == 63) ? 0.0 : w0[i]; }
            };
            using yes = std::integral_constant<bool, true>;
            using no = std::integral_constant<bool, false>;
            if constexpr (REGF && G0_SEGS > 0 && Ts == G0_SEGS + 1) {
                // the fourth segment of the register form: lane 63 is free until stage 31 inserts its input column -- seven more stages
                // without bank 1 (its column update, gg row, staging stores and gradient terms: ~60 of ~150 vector instructions a stage)
                constexpr int kl = 8 * Ts - 1;          // the stage that needs the lane
                for (int k = 8 * (Ts - 1); k < N && k < kl - 1; k += 2) {
                    stage_body(k, tsc, yes(), Ra);
                    if (k + 1 < N) stage_body(k + 1, tsc, yes(), Rb);
                }
                if (kl - 1 < N) stage_body(kl - 1, tsc, yes(), Ra);
                move_g();
                if (kl < N) stage_body(kl, tsc, no(), Rb);
            } else {
                if constexpr (G0_SEGS > 0 && Ts == G0_SEGS + 1) move_g();
                if constexpr (REGF) {
                    if constexpr (Ts == 5) fetch_bcol(32 + (lane >> 1));          // (bank 1's input columns: the stages from 32 on)
                    for (int k = 8 * (Ts - 1); k < N && k < 8 * Ts; k += 2) {
                        stage_body(k, tsc, std::integral_constant<bool, G0S>(), Ra);
                        if (k + 1 < N) stage_body(k + 1, tsc, std::integral_constant<bool, G0S>(), Rb);
                    }
                } else {
                    for (int k = 8 * (Ts - 1); k < N && k < 8 * Ts; k++) stage_body(k, tsc, std::integral_constant<bool, G0S>(), Ra);
                }
            }
        });
        for (int s = N + 1; s <= NMAX; s++) {          // rows beyond the horizon: zeros
            const int c_ = (s - 1) >> 2;
            double *gcs = pa.cws + (size_t)b * NCH * 64 + 16 * ((s - 1) & 3) + lc;
            if (2 * lq <= c_) gcs[cidx(c_, lq) * 64] = 0.0;
            const int T1 = 4 + lq;
            if (lane < NB1 && 2 * T1 <= c_) gcs[cidx(c_, T1) * 64] = 0.0;
        }
    }
    if constexpr (NT_ == 5) {      // (the parked tiles come back)
#pragma unroll
        for (int K = 0; K < NPARK; K++) Ht[tidx(K, NT - 1)] = reinterpret_cast<d4 *>(lds + D::C_PARK)[K * 64 + lane];
    }
    // input cost (R) and padding on the diagonal, gradient of the input cost
#pragma unroll
    for (int K = 0; K < NT; K++)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int row = lq + 4 * jj;
            if (row == lc) {
                const int idx = 16 * K + row;
                Ht[tidx(K, K)][jj] += (idx < nv) ? sWt[(idx >> 1) * 6 + 4 + (idx & 1)] : 1.0;
            }
        }
    if (lane < nv) q0 += sWt[j0 * 6 + 4 + r0] * (sU0[lane] - gyref[j0 * 6 + 4 + r0]);
    {   // (the stage of a bank-1 column is recomputed from an opaque copy of the lane id: held since the top of the kernel it was the
        //  one value the register allocator sent to scratch)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int j1e = 32 + (lane_e >> 1);
        if (lane < NB1 && 64 + lane < nv) q1 += sWt[j1e * 6 + 4 + r0] * (sU0[64 + lane] - gyref[j1e * 6 + 4 + r0]);
    }
    // ---- hand-over: H tiles, q (the gg rows went out stage by stage)
    {
        d4 *gh = reinterpret_cast<d4 *>(pa.hws) + (size_t)b * NTT * 64 + lane;
#pragma unroll
        for (int t = 0; t < NTT; t++) gh[t * 64] = Ht[t];
        gvec[PV_Q + lane] = q0;
        if (lane < NB1) gvec[PV_Q + 64 + lane] = q1;
    }
}

// ---- K2 for small batches: six (N <= 40) or seven wavefronts per OCP.
// cond_kernel is one wavefront walking the N stages, each stage the column recursion, an exchange through LDS and the stage's
// share of the SYRK on the matrix cores: ~2400 cycles a stage, 41 us at N = 40 whatever the batch. Only the recursion is
// sequential in the stages. Here
//   phase 1  the recursion alone, FOUR lanes (a DPP quad) per column of G -- lane 0 the rows px and vl of A_k w, lane 1 py and vt,
//            lane 2 psi and r, each row's sum in apply_A's order; the new vl, vt, r, psi go round the quad by DPP broadcasts; lane 3
//            forms the gg row -- ~40 instructions a stage instead of ~150. All records are preloaded into LDS by the whole
//            workgroup and read one stage ahead (two register sets). Left behind: the four cost rows of every G_s in LDS (stage
//            s: 16 ceil(s/8) live columns), the g column, the gg rows and d (straight to the workspace, as cond_kernel does);
//   phase 2  (one workgroup barrier later) wavefronts 0..3 accumulate the Hessian tiles -- every tile by ONE wavefront, over
//            the stages in order, with the operands cond_kernel forms: the same sums in the same order -- and wavefronts 4, 5
//            the gradient q of the columns 0..63 / 64...
// Tiles are dealt to the four wavefronts longest-first (tile (K, I) is touched by the 8 (NT - I) stages behind block column I).
// The results are the ones cond_kernel writes, to the last bit (tests/test_gpu_parity.py::test_condensing_six_wavefronts_...).
// Launched while the batch is at most one workgroup per CU (tum_nmpc.hip: launch_pipeline); the nominal OCP only.
constexpr int CW_SYRK = 4;
template <int NT_> constexpr int cw_waves() { return (4 * (PD<NT_>::NVP + 1) + 63) / 64 > CW_SYRK + 2 ? (4 * (PD<NT_>::NVP + 1) + 63) / 64 : CW_SYRK + 2; }
template <int NT_> struct CondWideTab { int own[PD<NT_>::NTT]; };
template <int NT_> constexpr CondWideTab<NT_> cond_wide_deal()
{
    CondWideTab<NT_> t{};
    int load[CW_SYRK] = {};
    for (int I = 0; I < NT_; I++)
        for (int K = 0; K <= I; K++) {
            int best = 0;
            for (int w = 1; w < CW_SYRK; w++) if (load[w] < load[best]) best = w;
            t.own[PD<NT_>::tidx(K, I)] = best; load[best] += 8 * (NT_ - I);
        }
    return t;
}
template <int NT_> struct CondWide {
    using D = PD<NT_>;
    static constexpr CondWideTab<NT_> tab = cond_wide_deal<NT_>();
    static constexpr int ROWS = 512 * NT_ * (NT_ + 1) / 2;          // doubles: 4 rows x 16 Ts columns x 8 stages per segment Ts
    // first double of stage s (1-based) in the row store; the stage holds 4 rows of 16 ceil(s/8) columns
    static __device__ __forceinline__ int rowoff(int s) { const int Ts = (s + 7) >> 3; return 64 * (4 * Ts * (Ts - 1) + (s - 1 - 8 * (Ts - 1)) * Ts); }
};

// SN = true: the nominal copy of the coupled SNMPC OCP -- G_s, g_s and the chance-constraint rows of the stages s <= uph come
// from the prologue kernel's hand-over buffer, the recursion starts behind them; speed row |v|, gg row with its vt partial
// (the same differences as between cond_kernel<., false> and cond_kernel<., true>)
// FULLW (round 6, nominal OCP only): the stage cost takes a full symmetric W (KArgs::Wf; acados' cost_set(i, 'W', W) accepts any matrix,
// NMPC_class.py:290-296). With the output Jacobian of stage s in the condensed variables J = [rows 0..3 of G_s ; unit rows of the inputs 2s, 2s+1] and
// W = [Q S; S' R]: the SYRK's A operand becomes Q-mixed rows (Q J_x instead of w_r J_r), the 2 x 2 block R goes onto the diagonal tile, the cross term
// S' J_x is added to the rows 2s, 2s+1 of H (and, in the diagonal tiles, to their mirror image: the interior point kernel reads both triangles), and
// the gradient takes W (res + g) over all six outputs. The diagonal instantiation is untouched.
template <int NT_, bool SN, bool FULLW = false>
__global__ void __launch_bounds__(64 * cw_waves<NT_>()) cond_wide_kernel(const PArgs pa)
{
    static_assert(!(SN && FULLW), "the coupled SNMPC OCP takes a diagonal W");
    PD_LOCALS
    using CW = CondWide<NT_>;
    constexpr int CW_WAVES = cw_waves<NT_>();
    constexpr int GSTR = SN ? 8 : 4;            // doubles of the g column kept per stage (rows px, py, psi, vl; SN: vt too)
    __shared__ __attribute__((aligned(16))) double sRec[(NMAX + 3) * PREC];       // (+2: the read-ahead of the last stage stays inside)
    __shared__ __attribute__((aligned(16))) double sRows[CW::ROWS];
    __shared__ double sG[(NMAX + 1) * GSTR], sWt[(NMAX + 1) * 6], sU0[NVP], sEq[2 * ((NMAX + 1) * 4 + 2)];
    __shared__ double sWf[FULLW ? (NMAX + 1) * 36 : 1];          // full W per stage, scaled by dt (stage N: 1)
    const KArgs &ka = pa.ka;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, b = blockIdx.x;
    if (b >= ka.batch) return;
    const int N = ka.N, nv = 2 * N;
    const double dt = ka.dt;
    const double *grec = pa.rec + (size_t)b * (N + 1) * PREC;
    const double *gx0 = ka.x0 + (size_t)b * NX;
    const double *gX = ka.X + (size_t)b * (N + 1) * NX;
    const double *gU = ka.U + (size_t)b * N * NU;
    const double *gyref = ka.yref + (size_t)b * (N + 1) * 6;
    const double *gW = ka.W + (size_t)b * (N + 1) * 6;
    double *gvec = pa.vec + (size_t)b * PVEC;
    const int uph = SN ? ka.uph : 0;
    const int PP = SN ? sn_pro_pitch(uph) : 64, PSTAGE = 9 * PP;
    const double *gpro = SN ? ka.pro + (size_t)b * uph * PSTAGE : nullptr;
    // (flags & 8: the linearisation ran beside the planner and left the residuals of the cost to this kernel -- the same
    //  differences lin_cols_kernel forms, from the iterate and the reference of THIS solve)
    const bool late_res = !SN && (ka.flags & 8);
    for (int i = tid; i < (N + 1) * PREC; i += 64 * CW_WAVES) {
        double v = grec[i];
        const int f = i & (PREC - 1), r = f - PR_RES;
        if (late_res && r >= 0 && r < 4) {
            const int st = i / PREC;
            const double xv = gX[(size_t)st * NX + r], yv = gyref[(size_t)st * 6 + r];
            v = ((r == 2) ? wrap_yaw(xv) : xv) - yv;
        }
        sRec[i] = v;
    }
    for (int i = tid; i < (N + 1) * 6; i += 64 * CW_WAVES) sWt[i] = gW[i];
    if constexpr (FULLW) {
        const double *gWf = ka.Wf + (size_t)b * (N + 1) * 36;
        for (int i = tid; i < (N + 1) * 36; i += 64 * CW_WAVES) sWf[i] = ((i < 36 * N) ? dt : 1.0) * gWf[i];
    }
    for (int i = tid; i < NVP; i += 64 * CW_WAVES) sU0[i] = (i < nv) ? gU[i] : 0.0;
    __syncthreads();

    // ---- phase 1: the column recursion, four lanes per column (column slot NVP: the constant column g)
    if (tid < 4 * (NVP + 1)) {
        const int cs = tid >> 2, rho = tid & 3;
        const bool isg = (cs == NVP), colv = (cs < NVP);
        const int j = cs >> 1, r0 = cs & 1, T = cs >> 4, lc = cs & 15;
        const int ra = (rho == 3) ? 2 : rho, rb = ra + 3;         // rows of A_k w this lane forms (lane 3 repeats lane 2's and stores neither)
        // wa, wb: rows ra, rb of the column; w2..w5 as the quad last exchanged them; w6, w7 (delta, a: integrators of the inputs) on every lane
        double wa = 0.0, wb = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0, w5 = 0.0, w6 = 0.0, w7 = 0.0;
        if (isg) {
            wa = gx0[ra] - gX[ra]; wb = gx0[rb] - gX[rb];
            w2 = gx0[2] - gX[2]; w3 = gx0[3] - gX[3]; w4 = gx0[4] - gX[4]; w5 = gx0[5] - gX[5]; w6 = gx0[6] - gX[6]; w7 = gx0[7] - gX[7];
        }
        // what a stage leaves behind (na: row ra as this lane formed or fetched it)
        auto leave = [&](int s, double na, double nb, double hr, double hd, double xd, double cvl, double cvt) {
            const int Ts = (s + 7) >> 3;
            const double c3 = SN ? cvl * w3 + cvt * w4 : w3;    // the speed row of the cost: vl (nominal OCP) or |v|
            const double wrow = (rho == 3) ? c3 : na;           // row rho of the cost rows of G_s
            if (isg) {
                sG[s * GSTR + rho] = (rho == 3) ? w3 : na;
                if (SN && rho == 1) sG[s * GSTR + 4] = nb;
                if (rho == 3) {
                    gvec[PV_D + 2 * (s - 1)] = xd + w6;
                    gvec[PV_D + 2 * (s - 1) + 1] = hd + hr;
                }
            }
            {   // the gg row of stage s in the operand layout of the interior point kernel (see cond_kernel)
                const int c_ = (s - 1) >> 2;
                double *gcs = pa.cws + (size_t)b * NCH * 64 + 16 * ((s - 1) & 3) + lc;
                if (rho == 3 && colv && 2 * T <= c_) gcs[(T * (NC - T - 1) + c_) * 64] = hr;
            }
            if (colv && T < Ts) sRows[CW::rowoff(s) + rho * (16 * Ts) + cs] = wrow;
        };
        if constexpr (SN) {
            // stages s = k + 1 <= uph: PCE means of the sample recursions (prologue kernel), live columns only
            for (int k = 0; k < uph && k < N; k++) {
                const double *pg = gpro + (size_t)k * PSTAGE, *recn = sRec + (k + 1) * PREC;
                const int s = k + 1;
                const bool live = isg || (colv && cs < 2 * s);
                const int idx = isg ? 2 * uph : (live ? cs : 0);
                const double na = live ? pg[ra * PP + idx] : 0.0, nb = live ? pg[rb * PP + idx] : 0.0;
                w2 = live ? pg[2 * PP + idx] : 0.0; w3 = live ? pg[3 * PP + idx] : 0.0; w4 = live ? pg[4 * PP + idx] : 0.0;
                w5 = live ? pg[5 * PP + idx] : 0.0; w6 = live ? pg[6 * PP + idx] : 0.0; w7 = live ? pg[7 * PP + idx] : 0.0;
                wa = na; wb = nb;
                double hr = recn[PR_GH + 0] * w3 + recn[PR_GH + 1] * w5 + recn[PR_GH + 2] * w7, hd = recn[PR_GH + 3];
                hr += recn[PR_G4] * w4;
                if (s < uph) { hr = live ? pg[8 * PP + idx] : 0.0; hd = 0.0; }       // chance-constraint row E + kappa sqrt(Var) over the samples
                leave(s, na, nb, hr, hd, recn[PR_XD], recn[PR_CV], recn[PR_CV + 1]);
            }
        }
        struct Rk { double Sa[5], Sb[5], sp, Ba, Bb, ba, bb, b6, b7, g3, g5, g7, hd, xd, g4, cvl, cvt; };
        auto fetch = [&](int k, Rk &r) {
            const double *rec = sRec + k * PREC;
#pragma unroll
            for (int c = 0; c < 5; c++) { r.Sa[c] = rec[2 + ra * 7 + c]; r.Sb[c] = rec[2 + rb * 7 + c]; }
            r.sp = rec[ra & 1];                                 // (used by the rows px, py only)
            r.Ba = rec[2 + ra * 7 + 5 + r0]; r.Bb = rec[2 + rb * 7 + 5 + r0];
            r.ba = rec[44 + ra]; r.bb = rec[44 + rb]; r.b6 = rec[44 + 6]; r.b7 = rec[44 + 7];
            r.g3 = rec[PREC + PR_GH + 0]; r.g5 = rec[PREC + PR_GH + 1]; r.g7 = rec[PREC + PR_GH + 2]; r.hd = rec[PREC + PR_GH + 3];
            r.xd = rec[PREC + PR_XD];
            if constexpr (SN) { r.g4 = rec[PREC + PR_G4]; r.cvl = rec[PREC + PR_CV]; r.cvt = rec[PREC + PR_CV + 1]; }
            else { r.g4 = 0.0; r.cvl = 1.0; r.cvt = 0.0; }
        };
        auto stage = [&](int k, const Rk &r, Rk &nxt) {
            fetch(k + 1, nxt);      // (unconditional: behind a branch the wait for the previous stage's reads would cover these too; sRec holds N + 2 records)
            // w <- A_k w (apply_A, nmpc_device.hpp: rows px, py start from w + Sp w_psi, row psi from w_psi, the rest from 0)
            double na = (ra < 2) ? wa + r.sp * w2 : wa, nb = 0.0;
            na += r.Sa[0] * w3; na += r.Sa[1] * w4; na += r.Sa[2] * w5; na += r.Sa[3] * w6; na += r.Sa[4] * w7;
            nb += r.Sb[0] * w3; nb += r.Sb[1] * w4; nb += r.Sb[2] * w5; nb += r.Sb[3] * w6; nb += r.Sb[4] * w7;
            // the column of B_k (b_k for g) this column takes up at stage k
            const double sel = (colv && j == k) ? 1.0 : 0.0, selg = isg ? 1.0 : 0.0;
            na += sel * r.Ba; nb += sel * r.Bb;
            const double b6 = r0 ? dt : 0.0, b7 = r0 ? 0.0 : dt;
            w6 += sel * b6; w7 += sel * b7;
            na += selg * r.ba; nb += selg * r.bb; w6 += selg * r.b6; w7 += selg * r.b7;
            wa = na; wb = nb;
            w2 = quad_bcast<2>(na); w3 = quad_bcast<0>(nb); w4 = quad_bcast<1>(nb); w5 = quad_bcast<2>(nb);
            double hr = r.g3 * w3 + r.g5 * w5 + r.g7 * w7;
            if constexpr (SN) hr += r.g4 * w4;
            leave(k + 1, na, nb, hr, r.hd, r.xd, r.cvl, r.cvt);
        };
        {
            Rk ra_, rb_;
            const int k0 = (uph < N) ? uph : N;
            fetch(k0, ra_);
            for (int k = k0; k < N; k += 2) {
                stage(k, ra_, rb_);
                if (k + 1 < N) stage(k + 1, rb_, ra_);
            }
        }
        for (int s = N + 1; s <= NMAX; s++) {          // rows beyond the horizon: zeros
            const int c_ = (s - 1) >> 2;
            double *gcs = pa.cws + (size_t)b * NCH * 64 + 16 * ((s - 1) & 3) + lc;
            if (rho == 3 && colv && 2 * T <= c_) gcs[(T * (NC - T - 1) + c_) * 64] = 0.0;
        }
    }
    __syncthreads();

    // ---- phase 2
    const int lq = lane >> 4, lc = lane & 15;
    if (wv < CW_SYRK) {
        auto syrk = [&](auto wvc) {
            constexpr int WV = decltype(wvc)::value;
            d4 Ht[NTT];
#pragma unroll
            for (int i = 0; i < NTT; i++) Ht[i] = d4{0.0, 0.0, 0.0, 0.0};
            static_for<1, NT>([&](auto tsc) {
                constexpr int Ts = decltype(tsc)::value;
                // (the operands of the next stage are read while the matrix cores work on this one: two register sets, as in phase 1)
                // (FULLW: ao = row lq of Q J_x of the stage -- four reads and four FMAs per tile column instead of one product)
                double ao_a[FULLW ? Ts : 1], ao_b[FULLW ? Ts : 1];
                auto ldop = [&](int s, double (&bo)[Ts], double &wl, double *ao) {
                    const double sc = (s < N) ? dt : 1.0;
                    wl = sc * sWt[s * 6 + lq];
                    const double *row = sRows + CW::rowoff(s) + lq * (16 * Ts) + lc;
#pragma unroll
                    for (int T = 0; T < Ts; T++) bo[T] = row[16 * T];
                    if constexpr (FULLW) {
                        const double *r0_ = sRows + CW::rowoff(s) + lc, *wq = sWf + s * 36 + lq * 6;
#pragma unroll
                        for (int T = 0; T < Ts; T++) {
                            double a = 0.0;
#pragma unroll
                            for (int r = 0; r < 4; r++) a += wq[r] * r0_[r * (16 * Ts) + 16 * T];
                            ao[T] = a;
                        }
                    }
                };
                auto mm = [&](const double (&bop)[Ts], const double wl, const double *ao) {
                    static_for<0, Ts - 1>([&](auto Kc) {
                        constexpr int K = decltype(Kc)::value;
                        const double aop = FULLW ? ao[K] : bop[K] * wl;
                        static_for<K, Ts - 1>([&](auto Ic) {
                            constexpr int I = decltype(Ic)::value;
                            if constexpr (CW::tab.own[D::tidx(K, I)] == WV) Ht[D::tidx(K, I)] = mfma(aop, bop[I], Ht[D::tidx(K, I)]);
                        });
                    });
                };
                const int se = (N < 8 * Ts) ? N : 8 * Ts;
                int s = 8 * (Ts - 1) + 1;
                double ba[Ts], bb[Ts], wa = 0.0, wb = 0.0;
                if (s <= se) ldop(s, ba, wa, ao_a);
                for (; s <= se; s += 2) {
                    if (s + 1 <= se) ldop(s + 1, bb, wb, ao_b);
                    mm(ba, wa, ao_a);
                    if (s + 1 <= se) {
                        if (s + 2 <= se) ldop(s + 2, ba, wa, ao_a);
                        mm(bb, wb, ao_b);
                    }
                }
            });
            // input cost (R) and padding on the diagonal; hand-over of this wavefront's tiles
            d4 *gh = reinterpret_cast<d4 *>(pa.hws) + (size_t)b * NTT * 64 + lane;
            static_for<0, NT - 1>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                static_for<0, I>([&](auto Kc) {
                    constexpr int K = decltype(Kc)::value;
                    if constexpr (CW::tab.own[D::tidx(K, I)] == WV) {
                        if constexpr (FULLW) {
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                // component jj of lane (lq, lc) of tile (K, I) is H[16 I + lc][16 K + lq + 4 jj] (the matrix instruction's D[i][j]: i from the
                                // A operand = column tile K, j from the B operand = row tile I)
                                const int ir = 16 * I + lc, ic = 16 * K + lq + 4 * jj;
                                double add = 0.0;
                                // the 2 x 2 input block R of the stage both indices belong to; padding beyond the horizon
                                if (K == I && (ir >> 1) == (ic >> 1)) add = (ir < nv) ? sWf[(ir >> 1) * 36 + (4 + (ir & 1)) * 6 + 4 + (ic & 1)] : ((ir == ic) ? 1.0 : 0.0);
                                // the cross term S' J_x: row 2s + j of H against the columns below 2s -- and its mirror image in a diagonal tile
                                const int hi = (ir > ic) ? ir : ic, lo = (ir > ic) ? ic : ir;
                                const int s_ = hi >> 1, j_ = hi & 1;
                                if (s_ >= 1 && s_ < N && lo < 2 * s_) {
                                    const int Tss = (s_ + 7) >> 3;
                                    const double *rw_ = sRows + CW::rowoff(s_) + lo, *ws = sWf + s_ * 36 + 4 + j_;
#pragma unroll
                                    for (int r = 0; r < 4; r++) add += ws[r * 6] * rw_[r * (16 * Tss)];
                                }
                                Ht[tidx(K, I)][jj] += add;
                            }
                        } else if constexpr (K == I) {
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const int rw = lq + 4 * jj;
                                if (rw == lc) {
                                    const int idx = 16 * K + rw;
                                    Ht[tidx(K, K)][jj] += (idx < nv) ? dt * sWt[(idx >> 1) * 6 + 4 + (idx & 1)] : 1.0;
                                }
                            }
                        }
                        gh[tidx(K, I) * 64] = Ht[tidx(K, I)];
                    }
                });
            });
        };
        if (wv == 0) syrk(std::integral_constant<int, 0>());
        else if (wv == 1) syrk(std::integral_constant<int, 1>());
        else if (wv == 2) syrk(std::integral_constant<int, 2>());
        else syrk(std::integral_constant<int, 3>());
    } else {
        // gradient of the tracking cost through G_s (wavefront 4: columns 0..63, wavefront 5: columns 64..), input cost.
        // e_r(s) = w_r (res_r + g_r) of every stage first (a table per wavefront: wave-level ordering only), then the sums over the
        // stages with nothing but loads and FMAs in the loop (dead columns read a zero instead of being branched round)
        if (wv >= CW_SYRK + 2) return;
        const int col = 64 * (wv - CW_SYRK) + lane;
        const bool colv = col < NVP;
        const int T = col >> 4, j = col >> 1, r0 = col & 1;
        double *sE = sEq + (wv - CW_SYRK) * ((NMAX + 1) * 4 + 2);
        for (int i = lane; i < 4 * N; i += 64) {
            const int s = 1 + (i >> 2), r = i & 3;
            const double sc = (s < N) ? dt : 1.0;
            const double gs = (SN && r == 3) ? sRec[s * PREC + PR_CV] * sG[s * GSTR + 3] + sRec[s * PREC + PR_CV + 1] * sG[s * GSTR + 4] : sG[s * GSTR + r];
            if constexpr (FULLW) {
                // row r of W (res + g) over the stage's outputs: the four state outputs and (s < N) the two inputs
                double a = 0.0;
#pragma unroll
                for (int r2 = 0; r2 < 4; r2++) a += sWf[s * 36 + r * 6 + r2] * (sRec[s * PREC + PR_RES + r2] + sG[s * GSTR + r2]);
                if (s < N) a += sWf[s * 36 + r * 6 + 4] * (sU0[2 * s] - gyref[s * 6 + 4]) + sWf[s * 36 + r * 6 + 5] * (sU0[2 * s + 1] - gyref[s * 6 + 5]);
                sE[4 * s + r] = a;
            } else
            sE[4 * s + r] = (sc * sWt[s * 6 + r]) * (sRec[s * PREC + PR_RES + r] + gs);
        }
        if (lane == 0) sE[0] = 0.0;
        wsync();
        double q = 0.0;
        static_for<1, NT>([&](auto tsc) {
            constexpr int Ts = decltype(tsc)::value;
            const bool live = colv && T < Ts;
            const int se = (N < 8 * Ts) ? N : 8 * Ts;
#pragma unroll 4
            for (int s = 8 * (Ts - 1) + 1; s <= se; s++) {
                const double *row = live ? sRows + CW::rowoff(s) + col : sE;      // (sE[0] = 0, and what lies 16 Ts, 32 Ts, 48 Ts doubles behind it is finite)
                double a = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double wr_ = live ? row[r * 16 * Ts] : 0.0;
                    a += sE[4 * s + r] * wr_;
                }
                q += a;
            }
        });
        if constexpr (FULLW) {
            if (col < nv) {      // the input output 4 + r0 of stage j: row 4 + r0 of W against all six residuals of that stage (stage 0: g = x0 - X_0)
                const double *wr_ = sWf + j * 36 + (4 + r0) * 6;
                double a = wr_[4] * (sU0[2 * j] - gyref[j * 6 + 4]) + wr_[5] * (sU0[2 * j + 1] - gyref[j * 6 + 5]);
#pragma unroll
                for (int r2 = 0; r2 < 4; r2++) a += wr_[r2] * (sRec[j * PREC + PR_RES + r2] + ((j == 0) ? gx0[r2] - gX[r2] : sG[j * GSTR + r2]));
                q += a;
            }
        } else
        if (col < nv) q += dt * sWt[j * 6 + 4 + r0] * (sU0[col] - gyref[j * 6 + 4 + r0]);
        if (colv) gvec[PV_Q + col] = q;
    }
}

// ---------------------------------------------------------------------------------------------------------------- K4
// The expansion of ONE instance by one wavefront: dx recursion, full step, cost at the new iterate. `lds` holds PD::E_LDS doubles;
// FUSED: the caller (the tail of ipm_kernel) has put the step of the inputs into lds[E_DV ..] and passes status / slack cost in
// registers; otherwise both come from the workspace the interior point kernel wrote.
template <int NT_, bool SN, bool FUSED>
__device__ __forceinline__ void expand_instance(const PArgs &pa, const int b, double *lds, int status, double slack_cost)
{
    PD_LOCALS
    constexpr int E_REC = D::E_REC, E_X = D::E_X, E_U = D::E_U, E_DV = D::E_DV;
    const KArgs &ka = pa.ka;
    const int lane = threadIdx.x;
    const int N = ka.N, nv = 2 * N;
    const double dt = ka.dt;
    double *sRec = lds + E_REC, *sX = lds + E_X, *sU1 = lds + E_U, *sDv = lds + E_DV;
    const double *grec = pa.rec + (size_t)b * (N + 1) * PREC;
    double *gX = ka.X + (size_t)b * (N + 1) * NX;
    double *gU = ka.U + (size_t)b * N * NU;
    const double *gx0 = ka.x0 + (size_t)b * NX;
    const double *gyref = ka.yref + (size_t)b * (N + 1) * 6;
    const double *gW = ka.W + (size_t)b * (N + 1) * 6;
    const double *gvec = pa.vec + (size_t)b * PVEC;
    if (!FUSED) status = ka.status[b];
    const int uph = SN ? ka.uph : 0;
    const int PP = SN ? sn_pro_pitch(uph) : 64, PSTAGE = 9 * PP;
    const double *gpro = SN ? ka.pro + (size_t)b * uph * PSTAGE : nullptr;
    for (int i = lane; i < (N + 1) * NX; i += 64) sX[i] = gX[i];
    for (int i = lane; i < NVP; i += 64) { sU1[i] = (i < nv) ? gU[i] : 0.0; if (!FUSED) sDv[i] = gvec[PV_DV + i]; }
    // The stage records come in EX_AHEAD stages ahead of their use, through a ring of registers (rounds 2-5: two ahead -- the 40-stage recursion
    // then waits half a load latency per stage, and the kernel was that wait: 36 us for 4096 instances, a quarter of a single vehicle's control step
    // as the tail of the interior point kernel)
    constexpr int EX_AHEAD = 8;
    double pre[EX_AHEAD];
    sRec[lane] = (lane < PR_RES) ? grec[lane] : 0.0;
#pragma unroll
    for (int j = 0; j < EX_AHEAD; j++) pre[j] = (lane < PR_RES && 1 + j <= N) ? grec[(size_t)(1 + j) * PREC + lane] : 0.0;          // records 1 .. EX_AHEAD
    const double gx0r = gx0[(lane < 8) ? lane : 0];
    wsync();
    if (status == 0) {
        const int ri = (lane < 8) ? lane : 0;
        const bool core = ri < 6;
        const double diag = (ri < 3 || ri >= 6) ? 1.0 : 0.0;
        double dxi = gx0r - sX[ri];
        wsync();
        if (lane < 8) sX[lane] += dxi;
        if (SN && uph > 0) {
            // stages 1..uph of the nominal copy have been stepped by the epilogue kernel (PCE mean of the stepped sample copies),
            // which runs in front of this one; the recursion continues from its step of stage uph
            dxi = gvec[PV_SC + 8 + ri];
        }
        for (int k0 = 0; k0 < N; k0 += EX_AHEAD)
#pragma unroll
        for (int j = 0; j < EX_AHEAD; j++) {
            const int k = k0 + j;
            if (k >= N) break;
            const double *rec = sRec + (k & 1) * PREC;
            if (k >= uph) {
            const double *Si = rec + 2 + (core ? ri : 0) * 7;
            const double du0 = sDv[2 * k], du1 = sDv[2 * k + 1];
            const double cpsi = (ri < 2) ? rec[ri] : 0.0;
            double c0 = Si[0], c1 = Si[1], c2 = Si[2], c3 = Si[3], c4 = Si[4], c5 = Si[5], c6 = Si[6];
            if (!core) { c0 = c1 = c2 = c3 = c4 = 0.0; c5 = (ri == 7) ? dt : 0.0; c6 = (ri == 6) ? dt : 0.0; }
            const double bi = rec[44 + ri];
            const double x2 = rl(dxi, 2), x3 = rl(dxi, 3), x4 = rl(dxi, 4), x5 = rl(dxi, 5), x6 = rl(dxi, 6), x7 = rl(dxi, 7);
            double acc0 = diag * dxi + cpsi * x2 + bi;
            double acc1 = c5 * du0 + c6 * du1;
            acc0 += c0 * x3; acc1 += c1 * x4;
            acc0 += c2 * x5; acc1 += c3 * x6;
            acc0 += c4 * x7;
            dxi = acc0 + acc1;
            if (lane < 8) sX[(k + 1) * NX + lane] += dxi;
            }
            sRec[((k + 1) & 1) * PREC + lane] = pre[j];          // record k + 1 (requested EX_AHEAD stages ago)
            if (k + 1 + EX_AHEAD < N) pre[j] = (lane < PR_RES) ? grec[(size_t)(k + 1 + EX_AHEAD) * PREC + lane] : 0.0;
            wsync();
        }
        sU1[lane] += sDv[lane];
        if (lane < NB1) sU1[64 + lane] += sDv[64 + lane];
    }
    wsync();
    double cl = (lane == 0) ? (FUSED ? slack_cost : gvec[PV_SC]) : 0.0;      // slack part of the cost (interior point kernel)
    if (lane <= N) {
        double Wd[6], We[4], yr[6];          // the weights of this lane's stage
#pragma unroll
        for (int i = 0; i < 6; i++) { Wd[i] = gW[lane * 6 + i]; yr[i] = gyref[lane * 6 + i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) We[i] = Wd[i];
        const int k = lane;
        const double sc = (k < N) ? dt : 1.0;
        double acc = 0.0, e;
        e = sX[k * NX + 0] - yr[0]; acc += ((k < N) ? Wd[0] : We[0]) * e * e;
        e = sX[k * NX + 1] - yr[1]; acc += ((k < N) ? Wd[1] : We[1]) * e * e;
        e = wrap_yaw(sX[k * NX + 2]) - yr[2]; acc += ((k < N) ? Wd[2] : We[2]) * e * e;
        e = (SN ? sqrt(sX[k * NX + 3] * sX[k * NX + 3] + sX[k * NX + 4] * sX[k * NX + 4]) : sX[k * NX + 3]) - yr[3];
        acc += ((k < N) ? Wd[3] : We[3]) * e * e;
        if (k < N) {
            e = sU1[2 * k] - yr[4]; acc += Wd[4] * e * e;
            e = sU1[2 * k + 1] - yr[5]; acc += Wd[5] * e * e;
        }
        if (!SN && ka.Wf) {      // a full W (cost_set 'W' with off-diagonal entries): r' W r over the stage's outputs instead of the diagonal sum
            const double *Wk = ka.Wf + ((size_t)b * (N + 1) + k) * 36;
            double rr[6];
            rr[0] = sX[k * NX + 0] - yr[0]; rr[1] = sX[k * NX + 1] - yr[1]; rr[2] = wrap_yaw(sX[k * NX + 2]) - yr[2]; rr[3] = sX[k * NX + 3] - yr[3];
            rr[4] = (k < N) ? sU1[2 * k] - yr[4] : 0.0; rr[5] = (k < N) ? sU1[2 * k + 1] - yr[5] : 0.0;
            const int ny = (k < N) ? 6 : 4;
            acc = 0.0;
            for (int i = 0; i < ny; i++)
                for (int j2 = 0; j2 < ny; j2++) acc += Wk[i * 6 + j2] * rr[i] * rr[j2];
        }
        cl += 0.5 * sc * acc;
    }
    const double cost = wave_sum(cl);
    for (int i = lane; i < (N + 1) * NX; i += 64) gX[i] = sX[i];
    for (int i = lane; i < nv; i += 64) gU[i] = sU1[i];
    if (lane == 0) ka.cost[b] = cost;
}

template <int NT_, bool SN>
__global__ void __launch_bounds__(64) expand_kernel(const PArgs pa)
{
    using D = PD<NT_>;
    __shared__ __attribute__((aligned(16))) double lds[D::E_LDS];
    const int b = blockIdx.x;
    if (b >= pa.ka.batch) return;
    expand_instance<NT_, SN, false>(pa, b, lds, 0, 0.0);
}

// ---------------------------------------------------------------------------------------------------------------- K3
// lane-derived quantities of the interior point kernel (instantiated from an opaque copy of the lane id inside the loop, see
// TUM_LANE_DEFS in nmpc_kernel.hpp). The gg rows are held in registers in MFMA operand layout: chv[cidx(c, T)] of lane
// (lq, lc) is entry (row 4c+lq+1, column 16T+lc), zero right of the row's end.
#define PIPE_LANE_DEFS \
    const bool boxlane = lane < N; \
    const bool gglane = (SLOTS == 2) && lane >= NMAX && lane < NMAX + 20; \
    const int gj = gglane ? lane - NMAX : 0; \
    /* per row slot: is there a row, its stage, its type (0 steering-rate box, 1 steering angle, 2 gg), penalty class / scale */ \
    bool on[SLOTS]; int stg[SLOTS], ty[SLOTS], pix[SLOTS]; double psc[SLOTS]; \
    if constexpr (SLOTS == 2) { \
        stg[0] = gglane ? 2 * gj + 1 : lane; stg[1] = gglane ? 2 * gj + 2 : lane + 1; \
        on[0] = gglane ? (stg[0] <= N) : boxlane; on[1] = gglane ? (stg[1] <= N) : boxlane; \
        ty[0] = gglane ? 2 : 0; ty[1] = gglane ? 2 : 1; \
        const int pc0 = gglane ? ((stg[0] < N) ? 1 : 2) : ((lane == 0) ? 0 : 1), pc1 = (stg[1] < N) ? 1 : 2; \
        psc[0] = (gglane && stg[0] >= N) ? 1.0 : dt; psc[1] = (stg[1] < N) ? dt : 1.0; \
        pix[0] = (pc0 * 3 + ty[0]) * 4; pix[1] = (pc1 * 3 + ty[1]) * 4; \
    } else { \
        stg[0] = lane; stg[1] = lane + 1; stg[SLOTS - 1] = lane + 1; \
        on[0] = boxlane; on[1] = boxlane; on[SLOTS - 1] = boxlane; \
        ty[0] = 0; ty[1] = 1; ty[SLOTS - 1] = 2; \
        const int pc0 = (lane == 0) ? 0 : 1, pc1 = (lane + 1 < N) ? 1 : 2; \
        psc[0] = dt; psc[1] = (lane + 1 < N) ? dt : 1.0; psc[SLOTS - 1] = psc[1]; \
        pix[0] = (pc0 * 3 + 0) * 4; pix[1] = (pc1 * 3 + 1) * 4; pix[SLOTS - 1] = (pc1 * 3 + 2) * 4; \
    } \
    auto pen = [&](int slot, int sd, int quad) -> double {   /* z from the workspace (start / end of the solve), Z from LDS */ \
        return psc[slot] * (quad ? sPZ[(pix[slot] >> 1) + sd] : gpen[pix[slot] + sd]); \
    }; \
    const bool v0on = lane < nv, v1on = (lane < NB1) && (64 + lane < nv); \
    const bool odd = lane & 1; \
    auto ctw = [&](double &o0, double &o1) {   /* C' w for this lane's columns */ \
        double a0 = (odd && v0on) ? sWb[lane >> 1] + dt * sSfx[(lane >> 1) + 1] : 0.0; \
        double a1 = (odd && v1on) ? sWb[32 + (lane >> 1)] + dt * sSfx[32 + (lane >> 1) + 1] : 0.0; \
        double wv[NC]; \
    _Pragma("unroll") \
        for (int c = 0; c < NC; c++) wv[c] = sWh[4 * c + lq]; \
        double t[NT]; \
    _Pragma("unroll") \
        for (int T = 0; T < NT; T++) { \
            double e0 = 0.0, e1 = 0.0; \
    _Pragma("unroll") \
            for (int c = 2 * T; c < NC; c++) { if (c & 1) e1 += chv[cidx(c, T)] * wv[c]; else e0 += chv[cidx(c, T)] * wv[c]; } \
            t[T] = quad_sum(e0 + e1); \
        } \
        a0 += (lq == 0) ? t[0] : (lq == 1) ? t[1] : (lq == 2) ? t[2] : t[3]; \
        a1 += (NT > 6 && lq == 2) ? t[NT - 1] : (NT > 5 && lq == 1) ? t[5 < NT ? 5 : 4] : t[4];          /* bank 1: tile 4 on its lanes 0..15, tile 5 on 16..31, tile 6 on 32..47 */ \
        o0 = v0on ? a0 : 0.0; o1 = v1on ? a1 : 0.0; \
    }; \
    auto publish = [&](const double *w_, double *dstH) {   /* slot values of this lane -> box scalars, steering suffix sums, gg weights */ \
        const double sfx = wave_suffix(boxlane ? w_[1] : 0.0, lane); \
        wsync(); \
        if (lane < NMAX) sWb[lane] = boxlane ? w_[0] : 0.0; \
        if (lane < NMAX + 1) sSfx[lane + 1] = (lane < N) ? sfx : 0.0; \
        if constexpr (SLOTS == 2) { \
            if (gglane) { dstH[2 * gj] = on[0] ? w_[0] : 0.0; dstH[2 * gj + 1] = on[1] ? w_[1] : 0.0; } \
        } else { \
            if (lane < NMAX) dstH[lane] = boxlane ? w_[SLOTS - 1] : 0.0; \
        } \
        wsync(); \
    }; \
    int rb[NT]; \
    _Pragma("unroll") \
    for (int I = 0; I < NT; I++) rb[I] = lpk_row(16 * I, lc, (int)(__umul24(lc, lc + 1) >> 1));

// FUSE: the expansion of the instance (K4) runs as the tail of this kernel, in the LDS the factor no longer needs: one launch
// less per solve, and neither the step of the inputs nor the status / slack cost make a round trip through the workspace. Used
// for batches of at most one round of resident wavefronts (the host decides, tum_nmpc.hip: larger batches lose 5 % to the
// expansion's loads running at this kernel's occupancy); the nominal OCP only -- the coupled SNMPC OCP keeps its own expansion
// kernel between this kernel and its epilogue
template <bool PROF, int NT_, bool FUSE = false>
__global__ void __launch_bounds__(64, 1) ipm_kernel(const PArgs pa)
{
    PD_LOCALS
    constexpr int I_WH = D::I_WH, I_WB = D::I_WB, I_SFX = D::I_SFX, I_DV = D::I_DV, I_DUMMY = D::I_DUMMY, I_PZ = D::I_PZ, I_M = D::I_M;
    constexpr int NS2 = 2 * SLOTS;                 // row sides of this lane
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const KArgs &ka = pa.ka;
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= ka.batch) return;
    const int b = ka.order ? ka.order[blockIdx.x] : (int)blockIdx.x;
    const int N = ka.N, nv = 2 * N;
    const double dt = ka.dt;
    double p_mu0 = ka.mu0, p_t0 = ka.t0, p_reg = ka.reg, p_ts = ka.tol_stat, p_ti = ka.tol_ineq, p_tc = ka.tol_comp;
    int p_itmax = ka.iter_max;
    asm volatile("" : "+v"(p_mu0), "+v"(p_t0), "+v"(p_reg), "+v"(p_ts), "+v"(p_ti), "+v"(p_tc), "+v"(p_itmax));

    double *sPZ = lds + I_PZ;
    double *sBk = lds + D::I_BK;
    double *sM = lds + I_M, *sWh = lds + I_WH, *sGamH = lds + I_WH, *sWb = lds + I_WB, *sSfx = lds + I_SFX, *sDv = lds + I_DV;
    const double *gU = ka.U + (size_t)b * N * NU;
    const double *gpen = ka.pen + (size_t)b * 36;
    const double *gbnd = ka.bnd + (size_t)b * 6 * (N + 1);
    double *gvec = pa.vec + (size_t)b * PVEC;
    const d4 *ghws = reinterpret_cast<const d4 *>(pa.hws) + (size_t)b * NTT * 64 + lane;

    long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
    for (int i = lane; i < I_PZ; i += 64) lds[i] = 0.0;
    if (lane == 0) lds[D::I_ZERO] = 0.0;
    if constexpr (D::DENSE_W)
        for (int i = lane; i < NT * D::W_TILE; i += 64) { const int row = (i >> 4) & 15; lds[D::I_W + i] = (((i & 15) ^ D::tile_swz(row)) == row) ? 1.0 : 0.0; }
    if (lane < 18) sPZ[lane] = gpen[(lane >> 1) * 4 + 2 + (lane & 1)];
    // the gg rows, MFMA operand layout (30 coalesced loads), resident in registers for the whole solve: the KKT assembly and the
    // row phases take their operands from them.
    double chv[NCH];
    const double *gcw = pa.cws + (size_t)b * NCH * 64 + lane;
#pragma unroll
    for (int i = 0; i < NCH; i++) chv[i] = gcw[i * 64];
    const double q0 = gvec[PV_Q + lane], q1 = (lane < NB1) ? gvec[PV_Q + 64 + lane] : 0.0;
    const int lq = lane >> 4, lc = lane & 15;
    wsync();

    double v0 = 0.0, v1 = 0.0, rv0, rv1, qn;
    double rowst[6][NS2];    // IPM row state of this lane: [s, t, lam, mu, rs, rt][slot*2+side]
    const double npairs = 12.0 * N;
    const double inv_npairs = 1.0 / npairs;
    int it = 0, qp_status = 1;
    double res_stat = 0.0, res_ineq = 0.0, res_comp = 0.0;
    {
        PIPE_LANE_DEFS
        {
            const int NB = N + 1;
            double dval[SLOTS], lo[SLOTS], hi[SLOTS];
#pragma unroll
            for (int rr = 0; rr < SLOTS; rr++) {
                // value of the row at dU = 0 and its bounds: steering rate of the iterate / steering angle / gg row (cond_kernel)
                const int t_ = ty[rr];
                const int i_ = on[rr] ? stg[rr] : ((t_ == 0) ? 0 : 1);
                dval[rr] = (t_ == 0) ? gU[2 * i_ + 1] : gvec[PV_D + 2 * (i_ - 1) + ((t_ == 2) ? 1 : 0)];
                lo[rr] = gbnd[(2 * t_) * NB + i_]; hi[rr] = gbnd[(2 * t_ + 1) * NB + i_];
            }
            // Warm start (acados: qp_solver_warm_start, SNMPC_acados_settings.py:307): when the previous QP of this instance converged, the method starts
            // at the smaller complementarity target warm_mu, a violation slack the last QP ended with stays, and the last multiplier of a
            // row is kept as far as the row is still near its bound in the new problem (within ten times the centred value mu / t, below
            // the slack-equation bound), the row pushed off the boundary to t lam = mu. -9 % interior point iterations over the logged
            // warm-started loops (profiles/r05_ipm_iterations.txt); a cold start is untouched.
            const double *gsl = ka.slack + (size_t)b * 6 * N, *glm = ka.qp_lam + (size_t)b * (6 * N + 2);
            bool warm = ka.warm_mu > 0.0 && glm[6 * N] != 0.0;          // (wave-uniform)
            double r0a[NS2], swa[NS2], lwa[NS2];
            {
                // how far is the previous QP's final point from the new problem: row sides whose activity changed (multiplier above 1e-3
                // against margin + old slack below 1e-3) and the largest violation beyond the old slack. A closed loop changes 0-2 row
                // sides per control step; a sequence whose initial state jumps changes dozens, and there stale multipliers cost iterations
                // and can stall the method: then it starts cold (scripts/study/warm_gate.py: 8.55 -> 7.91 iterations on such a sequence,
                // 99 % of the logged closed-loop solves still start warm)
                int flips = 0; double viol = -1e300;
#pragma unroll
                for (int rr = 0; rr < SLOTS; rr++)
#pragma unroll
                    for (int sd = 0; sd < 2; sd++) {
                        const int k = rr * 2 + sd;
                        const bool on_ = on[rr];
                        const double eps = sd ? -1.0 : 1.0, bnd = sd ? hi[rr] : lo[rr];
                        r0a[k] = eps * (dval[rr] - bnd);
                        const int widx = sd * 3 * N + (on_ ? ((ty[rr] == 0) ? stg[rr] : N + 2 * (stg[rr] - 1) + ((ty[rr] == 2) ? 1 : 0)) : 0);
                        swa[k] = warm ? gsl[widx] : 0.0; lwa[k] = warm ? glm[widx] : 0.0;
                        const bool mis = on_ && ((lwa[k] > 1e-3) != (r0a[k] + swa[k] < 1e-3));
                        flips += __popcll(__ballot(mis));
                        viol = fmax(viol, on_ ? -r0a[k] - swa[k] : -1e300);
                    }
                if (warm && ka.warm_flips >= 0) {
                    viol = wave_max(viol);
                    if (flips > ka.warm_flips || viol > ka.warm_viol) {
                        warm = false;
#pragma unroll
                        for (int k = 0; k < NS2; k++) { swa[k] = 0.0; lwa[k] = 0.0; }
                    }
                }
            }
            const double mu0_ = warm ? ka.warm_mu : p_mu0, t0_ = warm ? ka.warm_mu : p_t0;
#pragma unroll
            for (int rr = 0; rr < SLOTS; rr++)
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    const int k = rr * 2 + sd;
                    const bool on_ = on[rr];
                    const double r0v = r0a[k];
                    const double z = pen(rr, sd, 0), Z = pen(rr, sd, 1);
                    double s0 = mu0_ / (z > 1e-6 ? z : 1e-6);
                    const double sw = swa[k], lw0 = lwa[k];
                    if (sw > s0) s0 = sw;
                    double t = r0v + s0;
                    if (t < t0_) t = t0_;
                    double lam = mu0_ / t;
                    {
                        double lw = lw0;
                        if (lw > 10.0 * lam) lw = 10.0 * lam;
                        if (lw > lam) {
                            const double cap = 0.99 * (z + Z * s0);
                            if (lw > cap) lw = cap;
                            lam = lw;
                            const double tc = mu0_ / lam;
                            if (t < tc) t = tc;
                        }
                    }
                    double ms = z + Z * s0 - lam;
                    const double msf = 1e-2 * mu0_ / s0;
                    if (ms < msf) ms = msf;
                    ROWF(0, k) = on_ ? s0 : 1.0; ROWF(1, k) = on_ ? t : 1.0; ROWF(2, k) = on_ ? lam : 1.0; ROWF(3, k) = on_ ? ms : 1.0;
                    ROWF(4, k) = on_ ? z + Z * s0 - lam - ms : 0.0;
                    ROWF(5, k) = on_ ? t - r0v - s0 : 0.0;
                }
        }
        wsync();
        {
            double w_[SLOTS];
#pragma unroll
            for (int rr = 0; rr < SLOTS; rr++) w_[rr] = ROWF(2, 2 * rr) - ROWF(2, 2 * rr + 1);
            publish(w_, sWh);
        }
        {
            double c0, c1;
            ctw(c0, c1);
            rv0 = v0on ? q0 - c0 : 0.0; rv1 = v1on ? q1 - c1 : 0.0;
        }
        qn = wave_max(fmax(fabs(q0), (lane < NB1) ? fabs(q1) : 0.0));
        if (qn < 1.0) qn = 1.0;
    }
    TUM_TICK(1);
    const int lane_outer = lane;
    for (;; it++) {
        int lane_v = lane_outer;
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v;
        const int lq = lane >> 4, lc = lane & 15;
        const int trilq = (lq * (lq + 1)) >> 1;

        PIPE_LANE_DEFS
        // ---- row phase A: residual norms, gamma
        double gap;
        // D, G of every row side, fixed for both solves of an iteration: the five-tile build computes them with gamma and HOLDS them
        // across the factorisation (16 registers); the six-tile build, at the top of the register file, recomputes them behind it
        constexpr bool HOLD_DG = (NT_ == 5);
        double rD[NS2], rG[NS2];
        {
            double ls = fmax(fabs(rv0), fabs(rv1)), li = 0.0, lcmp = 0.0, lg = 0.0;
            double gsum[SLOTS];
#pragma unroll
            for (int rr = 0; rr < SLOTS; rr++) {
                gsum[rr] = 0.0;
                const bool on_ = on[rr];
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    const int k = rr * 2 + sd;
                    const double s_ = ROWF(0, k), t_ = ROWF(1, k), l_ = ROWF(2, k), m_ = ROWF(3, k);
                    // (a row side that does not exist keeps s = t = lam = mu = 1 and zero residuals from the initial point on -- its
                    //  steps are masked through the step length below --, so its residuals need no mask here)
                    ls = fmax(ls, fabs(ROWF(4, k)));
                    li = fmax(li, fabs(ROWF(5, k)));
                    const double c1 = t_ * l_, c2 = s_ * m_;
                    lcmp = fmax(lcmp, on_ ? fmax(c1, c2) : 0.0);
                    lg += on_ ? c1 + c2 : 0.0;
                    // D = 1/(Z s + mu), G = 1/(t + lam s D)  (rounds 2-3 recomputed them behind the factorisation everywhere -- eight
                    // quarter-rate reciprocals with their Newton steps -- when the register file had no room)
                    const double D_ = frcp(pen(rr, sd, 1) * s_ + m_), G_ = frcp(t_ + l_ * s_ * D_);
                    if constexpr (HOLD_DG) { rD[k] = D_; rG[k] = G_; }
                    gsum[rr] += l_ * G_;
                }
            }
            res_stat = ls; res_ineq = li; res_comp = lcmp;
            gap = wave_sum(lg) * inv_npairs;
            const bool lane_nan = !(ls == ls) || !(li == li) || !(lcmp == lcmp);
            if (__any(lane_nan) || !(gap == gap)) { qp_status = 3; break; }
            const bool lane_open = (ls > p_ts * qn) || (li > p_ti) || (lcmp > p_tc);
            if (!__any(lane_open)) { qp_status = 0; break; }
            if (it >= p_itmax) { qp_status = 1; break; }
            TUM_TICK(2);
            publish(gsum, sGamH);
            if constexpr (D::PRE_DIAG) {
                const double dt2_ = dt * dt;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int r = 64 * h + lane;
                    if (h == 0 || lane < NVP - 64) {
                        const bool has = (r & 1) && r < nv;
                        lds[D::I_XS + r] = dt2_ * lds[has ? I_SFX + (r >> 1) + 1 : D::I_ZERO];
                        lds[D::I_XB + r] = lds[has ? I_WB + (r >> 1) : D::I_ZERO];
                    }
                }
                wsync();
            }
        }
        // ---- M = H + C' Gamma C is assembled block column by block column INSIDE the factorisation below, straight into the
        // register tiles the factorisation works on (M itself never exists in LDS). H is streamed from the workspace HD tiles
        // ahead of its use, in the order the block columns need it: (0,0) (1,0) .. (NT-1,0) (1,1) (2,1) ...
        constexpr int HD = 4;
        auto hpos = [](int J, int I) { return J * NT - J * (J - 1) / 2 + I - J; };       // position of tile (row I, column J)
        auto hseq = [&](const int n) -> d4 {
            int J2 = 0;
            while (J2 + 1 < NT && (J2 + 1) * NT - (J2 + 1) * J2 / 2 <= n) J2++;
            return ghws[tidx(J2, J2 + n - (J2 * NT - J2 * (J2 - 1) / 2)) * 64];
        };
        d4 hq[HD];
#pragma unroll
        for (int n = 0; n < HD; n++) hq[n] = hseq(n);
        const double dt2 = dt * dt;
        // everything the block columns take from the small LDS tables is fetched up front (the loads cannot be moved across the
        // stores of the factorisation by the compiler, and at the start of a block column their latency would be exposed):
        // gamma per chunk, the structured (steering-angle) rows' term of the off-diagonal tiles (depends on the row only) and
        // the terms of the diagonal tiles
        constexpr bool PRE_DIAG = D::PRE_DIAG;
        double gch[NC], sfxo[NT], dadd[NT][4];
#pragma unroll
        for (int c = 0; c < NC; c++) gch[c] = sGamH[4 * c + lq];
#pragma unroll
        for (int I = 0; I < NT; I++) {
            const int r_ = 16 * I + lc;
            // (a column / row that carries no steering term -- even, or beyond nv -- reads a zero, and the smaller of two such values
            //  is the term of the entry)
            double sfc;
            if constexpr (PRE_DIAG) sfc = lds[D::I_XS + r_];
            else sfc = dt2 * lds[((lc & 1) && r_ < nv) ? I_SFX + (r_ >> 1) + 1 : D::I_ZERO];
            sfxo[I] = (lq & 1) ? sfc : 0.0;
            if constexpr (PRE_DIAG) {
                // entry (row, col) of a diagonal tile takes the suffix sum at max(row, col): the suffix sums of the (positive)
                // weights do not increase with the index, so that is the smaller of the row's and the column's; the box term
                // sits on the diagonal only and is read for the lane's column
                const double dterm = p_reg + lds[D::I_XB + r_];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) dadd[I][jj] = fmin(lds[D::I_XS + 16 * I + 4 * jj + lq], sfc) + ((lq + 4 * jj == lc) ? dterm : 0.0);
            }
        }
        // ---- blocked L D L' factorisation (row-panel register tiles as in the fused kernel; the 4-column micro-panels differ:
        //      no LDS round trips, see below)
        // smallest pivot, tracked through the HIGH WORDS of the pivots as signed integers (every lane its own: lane x of a block holds
        // pivot x; any lane below the threshold fails the factorisation): for positive doubles the order is the same, a negative
        // pivot has a negative high word; one integer minimum per micro-panel
        int dmin_hi = 0x3ff00000;
        // lane-distributed pivot blocks: x = lc & 3 is the lane's row of the 4 x 4 block; its column j is read from the strip of the
        // micro-panel at 16 j + 4 m + x, its diagonal at 17 x + 4 m; masks of the strictly lower part, the unit vector of the lane's row
        // of lanes (X = L^-1 e_lq) and (lq == 0)
        const int mpx = lc & 3;
        const double *mp_col = sBk + mpx, *mp_dg = sBk + 17 * mpx;
        double mpM0 = (mpx > 0) ? 1.0 : 0.0, mpM1 = (mpx > 1) ? 1.0 : 0.0, mpM2 = (mpx > 2) ? 1.0 : 0.0;
        double mp_ex = (mpx == lq) ? 1.0 : 0.0, mp_e0 = (lq == 0) ? 1.0 : 0.0;
        asm volatile("" : "+v"(mpM0), "+v"(mpM1), "+v"(mpM2), "+v"(mp_ex), "+v"(mp_e0));
        // tiled factor: element (row lc, column 4 q + lq) of a tile, q = 0..3 (operands of the left-looking update, column stores of
        // the micro-panels), and (row 4 q + lq, column lc) (stores of the inverse diagonal blocks)
        const int slc = D::tile_swz(lc);
        int e4[4], w4[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) { e4[q4] = 16 * lc + ((4 * q4 + lq) ^ slc); w4[q4] = 16 * (4 * q4 + lq) + (lc ^ D::tile_swz(4 * q4 + lq)); }
        double *sT = lds + I_M;
        (void)e4; (void)w4; (void)sT;
        static_for<0, NT - 1>([&](auto Jc) {      // (a template recursion: the body is far beyond the size up to which `#pragma unroll` is honoured)
            constexpr int J = decltype(Jc)::value;
            const int nd = NT - J;            // tiles below the diagonal one + the identity tile
            d4 T[NT + 1];       // T[NT]: the identity, taken through the same steps as the tiles below the diagonal one
#pragma unroll
            for (int jj = 0; jj < 4; jj++) T[NT][jj] = (lc == lq + 4 * jj) ? 1.0 : 0.0;
            {
                // block column J of M: the column operand of the gg rows is scaled by gamma once (chunks 2J..NC-1) and reused by
                // every tile below; tile (I, J) takes the chunks whose rows reach row tile I (c >= 2 I)
                double as[NC];
#pragma unroll
                for (int c = 2 * J; c < NC; c++) as[c] = chv[cidx(c, J)] * gch[c];
#pragma unroll
                for (int I = J; I < NT; I++) {
                    const int n = hpos(J, I);
                    d4 acc = hq[n % HD];
                    if (n + HD < NTT) hq[n % HD] = hseq(n + HD);
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        if constexpr (PRE_DIAG) acc[jj] += (I == J) ? dadd[J][jj] : sfxo[I];
                        else if (I > J) acc[jj] += sfxo[I];
                        else {
                            const int row = 16 * J + lq + 4 * jj, col = 16 * J + lc;
                            const int mx = (row > col) ? row : col;
                            const double sf = sSfx[(mx >> 1) + 1];
                            double add = ((row & 1) && (col & 1) && mx < nv) ? dt2 * sf : 0.0;
                            const double wb = sWb[(row >> 1) < NMAX ? (row >> 1) : 0];
                            if (row == col) add += p_reg + (((row & 1) && row < nv) ? wb : 0.0);
                            acc[jj] += add;
                        }
                    }
#pragma unroll
                    for (int c = 2 * I; c < NC; c++) acc = mfma(as[c], chv[cidx(c, I)], acc);
                    T[I] = acc;
                }
            }
            TUM_TICK(3);
#pragma unroll
            for (int K = 0; K < J; K++)
#pragma unroll
                for (int kc = 0; kc < 4; kc++) {
                    const int kk = 16 * K + 4 * kc + lq;
                    if constexpr (D::TILED) {
                        const double aJ = -sT[D::offt(J, K) * 256 + e4[kc]] * lds[D::I_D + kk];
#pragma unroll
                        for (int I = J; I < NT; I++) T[I] = mfma(aJ, sT[D::offt(I, K) * 256 + e4[kc]], T[I]);
                    } else {
                    const double aJ = -sM[rb[J] + kk] * sM[lpk_row(16 * K + 4 * kc, lq, trilq) + kk];          // (D of column kk: entry (kk, kk))
#pragma unroll
                    for (int I = J; I < NT; I++) T[I] = mfma(aJ, sM[rb[I] + kk], T[I]);
                    }
                }
            TUM_TICK(10);
            // four 4-column micro-panels. The 4x4 diagonal block sits in column m of the diagonal tile: entry (i, j) on lane
            // 16 j + 4 m + i. It is factorised lane-distributed (below); the panel below it is scaled by
            // ONE MFMA per tile with the 4x4 upper triangular P = L^-T D^-1 as the A operand (new column block = E P, straight
            // into the lanes that hold E), so there is no LDS round trip inside a micro-panel: finished columns are only
            // stored. The scaling is FOUR independent 4x4x4 products (16 rows x 4 columns, K = 4): one v_mfma_f64_4x4x4_4b
            // (measured 21 cycles; lane layout A[i][k] on lane 16 k + 4 blk + i, B[k][j] on 16 k + 4 blk + j, D[i][j] on
            // 16 i + 4 blk + j, scripts/probes/probe_mfma_4x4x4.cpp) instead of a quarter-used 16x16x4 (64 cycles): E sits in
            // the B layout as it is, the result lands where the 16x16x4 form put it.
            // Only the diagonal tile is on the dependency path (scale, rank-4 update, next strip); the two MFMAs
            // of every tile below it are owed to the NEXT micro-panel, which issues them in front of its pivot chain (`owed`, MPE / MPM
            // below; rounds 2-6a: between the chain's segments, which cost more than it hid). (An f64 MFMA occupies the same pipe as the f64
            // VALU instructions of its wavefront: measured, MFMA time and chain time add up.)
            // The identity tile T[NT] goes through the same two MFMAs: what comes out is W = L^-T D^-1 of the 16x16 diagonal
            // block, and L^-1 = (W D)^T is stored into the strict lower triangle of the block (the solves use the inverse
            // diagonal blocks; L of the diagonal block itself is never read again).
            double Lc[NT + 1], bd = 0.0, pop = 0.0, bprev = 0.0, LcW = 0.0;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int c0 = 16 * J + 4 * m;
                // k-th MFMA slot: what the previous micro-panel owes to the tiles
                // below the diagonal one and to the identity tile (their scaled columns, then their rank-4 updates)
                auto owed = [&](int k) {
                    if (m > 0 && k < nd) {
                        const int I = J + 1 + k;
                        Lc[I] = mfma4(pop, T[I][m - 1]);
                        if (I == NT) LcW = mfma4(bprev, T[NT][m - 1]);          // rows of L^-1 of the diagonal block: the unscaled X through the identity tile
                    } else if (m > 0 && k < 2 * nd) {
                        const int I = J + 1 + k - nd;
                        if (I < NT) { if constexpr (D::TILED) sT[D::offt(I, J) * 256 + e4[m > 0 ? m - 1 : 0]] = Lc[I]; else sM[rb[I] + c0 - 4 + lq] = Lc[I]; }
                        // (every lane stores: the unscaled X through the identity tile gives EXACTLY 1 on the diagonal of L^-1 and 0 above it --
                        //  the identity tile's entries above its diagonal only ever meet zero factors -- so the store needs no lane condition)
                        else if constexpr (D::DENSE_W) lds[D::I_W + J * D::W_TILE + w4[m > 0 ? m - 1 : 0]] = LcW;
                        else sM[(4 * (m - 1) + lq > lc) ? lpk_row(c0 - 4, lq, trilq) + 16 * J + lc : (I_DUMMY - I_M)] = LcW;
                        T[I] = mfma(bd, Lc[I], T[I]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                // The 4 x 4 block is factorised LANE-DISTRIBUTED (round 6; rounds 2-5 read its ten entries back from the strip at wave-uniform
                // addresses and ran the whole pivot arithmetic wave-uniform, with twenty v_cndmask per micro-panel to place P, the
                // reciprocals and the pivots on their lanes). The strip goes through LDS once and every lane reads ITS entries. Lane x (= lc & 3, the same in every 4-lane block of every row)
                // holds row x of the block: column j in register Aj (entries on and below the diagonal: the strip's upper triangle read
                // transposed), the diagonal in Dg. A pivot row's entry reaches the other lanes as the `row_newbcast` operand of the FMA
                // that uses it (fnmac_bc, mul_bc_fresh; the reciprocal of pivot j is taken on every lane and lane j's is the broadcast one): nothing of the block is
                // wave-uniform any more -- no ten broadcast reads, no lane selects of P, the reciprocals and the pivots (rounds 2-5: 40 FP64
                // instructions and 20 v_cndmask per micro-panel; now 28 and none). Dg ends as (d_0 .. d_3) on the lanes x = 0 .. 3, so
                // 1 / d_x for the scaling of P is one reciprocal of Dg; row lq runs the substitution L X = e_lq across its lanes, and
                // L D of the diagonal tile (the A operand of the rank-4 update) comes out of a second 4x4x4 product with the UNSCALED X
                // instead of a select of the pivots and a multiplication.
                // Where the matrix instructions owed to the tiles below go (slots 0 .. 2 nd - 1, `owed`): rounds 2-5 and the first form of round 6
                // spread them over the pivot chain (three behind the strip's reads, one between every two segments of the chain) "where their
                // result latency costs nothing". Measured in round 6 (profiles/r06_ab_mp_slots.txt): every matrix instruction inside the chain costs
                // MORE than its own issue time -- the chain's next instruction waits for it to drain whether it depends on it or not, and every
                // change between the vector and the matrix pipe has its own wait states -- and the kernel is 3.7 % faster with NONE inside: MPE slots in
                // front of the strip's store (they cover the rank-4 update of the diagonal tile the store waits for), MPM behind its reads (they
                // cover the LDS round trip), the chain of 28 FP64 instructions in one piece. (Only the seven-tile build has a 14th slot; it stays
                // behind the first segment.)
                constexpr int MPE = (NT == 6) ? 6 : 7, MPM = 13 - MPE;          // (measured per instantiation; six tiles: 6 + 7)
                static_for<0, MPE - 1>([&](auto kc) { owed(decltype(kc)::value); });
                sBk[lane] = T[J][m];
                double A0 = mp_col[4 * m], A1 = mp_col[16 + 4 * m], A2 = mp_col[32 + 4 * m], Dg = mp_dg[4 * m];
                static_for<MPE, MPE + MPM - 1>([&](auto kc) { owed(decltype(kc)::value); });
                // pivot 0
                const double r0 = frcp(Dg);                       // (lane 0: 1 / d_0)
                if constexpr (NT > 6) owed(13);          // (seven tiles: the first block column owes 2 x 7 slots)
                const double lm0 = mul_bc_fresh<0>(r0, A0, mpM0);          // column 0 of L below the diagonal (0 on the lanes x <= 0)
                Dg = fma(-lm0, A0, Dg);
                fnmac_bc<1>(A1, A0, lm0);
                fnmac_bc<2>(A2, A0, lm0);
                double bx = fma(-lm0, mp_e0, mp_ex);               // X = e_lq - l_0 X_0
                // pivot 1
                const double r1 = frcp(Dg);                       // (lane 1: 1 / d_1)
                const double lm1 = mul_bc_fresh<1>(r1, A1, mpM1);          // column 1 of L below the diagonal (0 on the lanes x <= 1)
                Dg = fma(-lm1, A1, Dg);
                fnmac_bc<2>(A2, A1, lm1);
                fnmac_bc_self<1, false, true>(bx, lm1);          // (bx: from the plain FMA above, wherever the compiler puts it)
                // pivot 2
                const double r2 = frcp(Dg);                       // (lane 2: 1 / d_2)
                const double lm2 = mul_bc_fresh<2>(r2, A2, mpM2);          // column 2 of L below the diagonal (0 on the lanes x <= 2)
                Dg = fma(-lm2, A2, Dg);
                fnmac_bc_self<2, true>(bx, lm2);          // (bx goes to matrix instructions: settled)
                // 1 / d_x on lane x, P = X / d, and the smallest pivot (per lane here: any lane may hold it)
                const double RR = frcp(Dg);
                dmin_hi = min(dmin_hi, __double2hiint(Dg));
                const double popn = bx * RR;
                const int rel = lc - (4 * m + lq);              // row - column inside the diagonal tile
                bprev = bx;
                pop = popn;
                Lc[J] = mfma4(pop, T[J][m]);
                bd = mfma4(-bx, T[J][m]);                       // - (L D) of the diagonal tile's columns
                // (the lane (lq, 4 m + lq) that stores pivot lq holds it in Dg: x = lq there)
                if constexpr (D::TILED) lds[(rel == 0) ? D::I_D + c0 + lq : I_DUMMY] = Dg;
                else sM[(rel == 0) ? rb[J] + c0 + lq : (I_DUMMY - I_M)] = Dg;
                // (the rank-4 update takes the scaled columns as they are: their rows on and above the 4x4 diagonal block --
                //  1 and 0 up to rounding -- only reach entries of the diagonal tile in rows or columns that are finished and
                //  never read again)
                if (m < 3) T[J] = mfma(bd, Lc[J], T[J]);
            }
            // what the last micro-panel owes: scaled columns
#pragma unroll
            for (int I = J + 1; I <= NT; I++) Lc[I] = mfma4(pop, T[I][3]);
            LcW = mfma4(bprev, T[NT][3]);
#pragma unroll
            for (int I = J + 1; I <= NT; I++) {
                if (I < NT) { if constexpr (D::TILED) sT[D::offt(I, J) * 256 + e4[3]] = Lc[I]; else sM[rb[I] + 16 * J + 12 + lq] = Lc[I]; }
                else if constexpr (D::DENSE_W) lds[D::I_W + J * D::W_TILE + w4[3]] = LcW;
                else sM[(12 + lq > lc) ? lpk_row(16 * J + 12, lq, trilq) + 16 * J + lc : (I_DUMMY - I_M)] = LcW;
            }
            wsync();
            TUM_TICK(11);
        });
        // (a failed factorisation -- a pivot that is not positive -- is acted upon after the parked registers are back:
        //  leaving the loop here would keep all of them alive across the factorisation for the code behind the loop)
        TUM_TICK(4);

        if (__any(dmin_hi < 0x01a56e1f)) { qp_status = 3; break; }          // a pivot below 1e-300 (or negative): the factorisation failed
        // ---- predictor / corrector
        if constexpr (!HOLD_DG) {
            int lane_r = lane_outer;
            asm volatile("" : "+v"(lane_r));
            const int lane = lane_r;
            const int lq = lane >> 4, lc = lane & 15;
            PIPE_LANE_DEFS
#pragma unroll
            for (int k = 0; k < NS2; k++) {
                rD[k] = frcp(pen(k >> 1, k & 1, 1) * ROWF(0, k) + ROWF(3, k));
                rG[k] = frcp(ROWF(1, k) + ROWF(2, k) * ROWF(0, k) * rD[k]);
            }
        }
        double cross1[NS2], cross2[NS2];
        double dv0 = 0.0, dv1 = 0.0, alpha = 1.0, sigma = 0.0;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            int lane_p = lane_outer;
            asm volatile("" : "+v"(lane_p));
            const int lane = lane_p;
            const int lq = lane >> 4, lc = lane & 15;
            PIPE_LANE_DEFS
            const double tau = (pass == 1) ? fmax(sigma * gap, 0.1 * p_tc) : 0.0;
            {
                double w[SLOTS];
#pragma unroll
                for (int rr = 0; rr < SLOTS; rr++) {
                    w[rr] = 0.0;
#pragma unroll
                    for (int sd = 0; sd < 2; sd++) {
                        const int k = rr * 2 + sd;
                        const double s_ = ROWF(0, k), t_ = ROWF(1, k), l_ = ROWF(2, k), m_ = ROWF(3, k);
                        const double D = rD[k], G = rG[k];
                        double rc1 = t_ * l_, rc2 = s_ * m_;
                        if (pass == 1) { rc1 += cross1[k] - tau; rc2 += cross2[k] - tau; }
                        const double gr = G * (rc1 - l_ * (ROWF(5, k) + (ROWF(4, k) * s_ + rc2) * D));
                        w[rr] += sd ? -gr : gr;
                    }
                }
                publish(w, sWh);
            }
            double b0, b1;
            ctw(b0, b1);
            b0 = v0on ? -rv0 - b0 : 0.0; b1 = v1on ? -rv1 - b1 : 0.0;
            TUM_TICK(5);
            {
                // The two substitutions on v_mfma_f64_4x4x4_4b (four independent 4x4x4 products per instruction, 21 cycles). A
                // vector block lives in "D form": lane (lq, lc) holds entry 4 (lc >> 2) + lq of the 16 -- what the instruction
                // returns (D[i][j] on lane 16 i + 4 blk + j, the same for every j when B does not depend on j) and what it
                // takes as the B operand (B[k][j] on lane 16 k + 4 blk + j) for the DIAGONAL 4x4 blocks of a 16x16 tile. The
                // other 4x4 blocks see the vector rotated by one, two, three quads inside a DPP row (row_ror: 4, 8, 12): a
                // 16x16 tile times a vector block is four accumulating 4x4x4 instructions, no row swaps, no lane gathers:
                // per block row the dependent chain is rotate -> 4 MFMAs -> rotate -> 4 MFMAs.
                const int blk = lc >> 2;
                double bj[NT], vr[NT][4];
#pragma unroll
                for (int J = 0; J < 4; J++) bj[J] = lane_gather(b0, (16 * J + 4 * blk + lq) << 2);
#pragma unroll
                for (int J = 4; J < NT; J++) bj[J] = lane_gather(b1, (16 * (J - 4) + 4 * blk + lq) << 2);
                int co[4];                                   // column (forward) / row (backward) offset of the block met at rotation d
#pragma unroll
                for (int d = 0; d < 4; d++) co[d] = 4 * ((blk - d) & 3) + lq;
                constexpr bool PRE = D::DENSE_W;          // (the five-tile build fetches the factor up front)
                double Lo[NTT][4], Ld[NT][4], Lp[NT];
                int tco[4];                                  // co (co + 1) / 2: the lane part of a packed row start (lpk_row)
#pragma unroll
                for (int d = 0; d < 4; d++) tco[d] = (int)(__umul24(co[d], co[d] + 1) >> 1);
                const int edg = 4 * blk + lq, tedg = (int)(__umul24(edg, edg + 1) >> 1) + edg;
                auto lo_f = [&](int J, int K, int d) { return sM[rb[J] + 16 * K + co[d]]; };
                // (the four loaded values of a tile are made opaque TOGETHER before the selects: otherwise the compiler turns
                //  "select of an LDS read" into an exec-masked read per element -- a branch, two exec saves and a wait of its own
                //  for each of the 40 operands; one opaque point per element would still wait for every read on its own)
                auto ld_f = [&](int J, double *o) {
                    double l0 = sM[rb[J] + 16 * J + co[0]], l1 = sM[rb[J] + 16 * J + co[1]], l2 = sM[rb[J] + 16 * J + co[2]],
                           l3 = sM[rb[J] + 16 * J + co[3]];
                    asm("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
                    o[0] = (co[0] < lc) ? l0 : ((co[0] == lc) ? 1.0 : 0.0); o[1] = (co[1] < lc) ? l1 : ((co[1] == lc) ? 1.0 : 0.0);
                    o[2] = (co[2] < lc) ? l2 : ((co[2] == lc) ? 1.0 : 0.0); o[3] = (co[3] < lc) ? l3 : ((co[3] == lc) ? 1.0 : 0.0);
                };
                auto lo_b = [&](int I, int J, int d) { return sM[lpk_row(16 * I, co[d], tco[d]) + 16 * J + lc]; };
                auto ld_b = [&](int J, double *o) {
                    double l0 = sM[lpk_row(16 * J, co[0], tco[0]) + 16 * J + lc], l1 = sM[lpk_row(16 * J, co[1], tco[1]) + 16 * J + lc],
                           l2 = sM[lpk_row(16 * J, co[2], tco[2]) + 16 * J + lc], l3 = sM[lpk_row(16 * J, co[3], tco[3]) + 16 * J + lc];
                    asm("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
                    o[0] = (co[0] > lc) ? l0 : ((co[0] == lc) ? 1.0 : 0.0); o[1] = (co[1] > lc) ? l1 : ((co[1] == lc) ? 1.0 : 0.0);
                    o[2] = (co[2] > lc) ? l2 : ((co[2] == lc) ? 1.0 : 0.0); o[3] = (co[3] > lc) ? l3 : ((co[3] == lc) ? 1.0 : 0.0);
                };
                auto l_piv = [&](int J) {       // D of entry 4 blk + lq of block J
                    if constexpr (D::TILED) return lds[D::I_D + 16 * J + edg];
                    else return sM[lpk_row(16 * J, edg, tedg) + 16 * J];
                };
                // The off-diagonal tile operands are only ever read by the matrix instructions, which take their A operand from an
                // ACCUMULATOR register as well: the reads are issued by hand with an accumulator register as the destination
                // (`ds_read_b64 a[..]`), so that the 80 registers they occupy during a substitution do not push the row state
                // (48 + 48 registers of s, t, lambda, mu, residuals, D, G, cross terms) out of the vector registers and back --
                // which the register allocator did with 200 v_accvgpr moves per pass. The compiler does not know these reads:
                // TUM_LDS_WAIT4 (s_waitcnt lgkmcnt(0), tied to the four values of a tile) stands between them and their use.
#define TUM_LDS_A64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "i"(off))
#define TUM_LDS_WAIT4(a) asm("s_waitcnt lgkmcnt(0)" : "+a"((a)[0]), "+a"((a)[1]), "+a"((a)[2]), "+a"((a)[3]))
                const unsigned sMb = (unsigned)(size_t)sM;         // LDS byte address of the factor
                const int slc = D::tile_swz(lc);
                if constexpr (PRE) {
                    // tile operands of the forward substitution: entry (row lc, column co[d]) of the tile -- one address per rotation
                    // for ALL tiles (the tile is an instruction offset), inverse diagonal blocks included
                    unsigned a[4];
#pragma unroll
                    for (int d = 0; d < 4; d++) a[d] = sMb + 8u * (unsigned)(16 * lc + (co[d] ^ slc));
#pragma unroll
                    for (int J = 1; J < NT; J++)
#pragma unroll
                        for (int K = 0; K < J; K++)
#pragma unroll
                            for (int d = 0; d < 4; d++) TUM_LDS_A64(Lo[D::tidx(K, J)][d], a[d], 2048 * D::offt(J, K));
#pragma unroll
                    for (int J = 0; J < NT; J++)
#pragma unroll
                        for (int d = 0; d < 4; d++) TUM_LDS_A64(Ld[J][d], a[d], 8 * (D::I_W - D::I_M) + 8 * D::W_TILE * J);
#pragma unroll
                    for (int J = 0; J < NT; J++) Lp[J] = l_piv(J);
#pragma unroll
                    for (int J = 1; J < NT; J++)
#pragma unroll
                        for (int K = 0; K < J; K++) TUM_LDS_WAIT4(Lo[D::tidx(K, J)]);
#pragma unroll
                    for (int J = 0; J < NT; J++) TUM_LDS_WAIT4(Ld[J]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int J = 0; J < NT; J++) {
                    double t = bj[J];
                    if (J > 0) {
                        double acc = 0.0, acc1 = 0.0;
#pragma unroll
                        for (int K = 0; K < J; K++)
#pragma unroll
                            for (int d = 0; d < 4; d++) {
                                const double lv = PRE ? Lo[tidx(K, J)][d] : lo_f(J, K, d);
                                if (d & 1) acc1 = mfma4a(lv, vr[K][d], acc1); else acc = mfma4a(lv, vr[K][d], acc);
                            }
                        t -= acc + acc1;
                    }
                    const double t1 = row_ror<4>(t), t2 = row_ror<8>(t), t3 = row_ror<12>(t);
                    if (!PRE) ld_f(J, Ld[J]);
                    double y = mfma4a(Ld[J][0], t, 0.0), y1 = mfma4a(Ld[J][1], t1, 0.0);
                    y = mfma4a(Ld[J][2], t2, y); y1 = mfma4a(Ld[J][3], t3, y1);
                    y += y1;
                    bj[J] = y * frcp(PRE ? Lp[J] : l_piv(J));
                    if (J < NT - 1) { vr[J][0] = y; vr[J][1] = row_ror<4>(y); vr[J][2] = row_ror<8>(y); vr[J][3] = row_ror<12>(y); }
                }
                if constexpr (PRE) {
                    // ... of the backward substitution: the transposes, entry (row co[d], column lc)
                    unsigned a[4];
#pragma unroll
                    for (int d = 0; d < 4; d++) a[d] = sMb + 8u * (unsigned)(16 * co[d] + (lc ^ D::tile_swz(co[d])));
#pragma unroll
                    for (int I = 1; I < NT; I++)
#pragma unroll
                        for (int J = 0; J < I; J++)
#pragma unroll
                            for (int d = 0; d < 4; d++) TUM_LDS_A64(Lo[D::tidx(J, I)][d], a[d], 2048 * D::offt(I, J));
#pragma unroll
                    for (int J = 0; J < NT; J++)
#pragma unroll
                        for (int d = 0; d < 4; d++) TUM_LDS_A64(Ld[J][d], a[d], 8 * (D::I_W - D::I_M) + 8 * D::W_TILE * J);
#pragma unroll
                    for (int I = 1; I < NT; I++)
#pragma unroll
                        for (int J = 0; J < I; J++) TUM_LDS_WAIT4(Lo[D::tidx(J, I)]);
#pragma unroll
                    for (int J = 0; J < NT; J++) TUM_LDS_WAIT4(Ld[J]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef TUM_LDS_A64
#undef TUM_LDS_WAIT4
#pragma unroll
                for (int J = NT - 1; J >= 0; J--) {
                    double t = bj[J];
                    if (J < NT - 1) {
                        double acc = 0.0, acc1 = 0.0;
#pragma unroll
                        for (int I = J + 1; I < NT; I++)
#pragma unroll
                            for (int d = 0; d < 4; d++) {
                                const double lv = PRE ? Lo[tidx(J, I)][d] : lo_b(I, J, d);
                                if (d & 1) acc1 = mfma4a(lv, vr[I][d], acc1); else acc = mfma4a(lv, vr[I][d], acc);
                            }
                        t -= acc + acc1;
                    }
                    const double t1 = row_ror<4>(t), t2 = row_ror<8>(t), t3 = row_ror<12>(t);
                    if (!PRE) ld_b(J, Ld[J]);
                    double x = mfma4a(Ld[J][0], t, 0.0), x1 = mfma4a(Ld[J][1], t1, 0.0);
                    x = mfma4a(Ld[J][2], t2, x); x1 = mfma4a(Ld[J][3], t3, x1);
                    x += x1;
                    bj[J] = x;
                    if (J > 0) { vr[J][0] = x; vr[J][1] = row_ror<4>(x); vr[J][2] = row_ror<8>(x); vr[J][3] = row_ror<12>(x); }
                }
                TUM_TICK(6);
                // back to lane = variable through the v-space buffer of the row phase (one representative lane per entry)
                wsync();
                if ((lc & 3) == 0) {
#pragma unroll
                    for (int J = 0; J < NT; J++) sDv[16 * J + 4 * blk + lq] = bj[J];
                }
                wsync();
                dv0 = sDv[lane]; dv1 = (lane < NB1) ? sDv[64 + lane] : 0.0;
                dv0 = v0on ? dv0 : 0.0; dv1 = v1on ? dv1 : 0.0;
            }
            // row phase B2: C*dv for this lane's rows, step in (s,t,lam,mu), step length
            double cdv[SLOTS];
            {
                const double xo = boxlane ? sDv[2 * lane + 1] : 0.0;
                const double pfx = dt * wave_prefix(xo, lane);
                // gg rows from the operand registers: this lane's partial sums over its column of every tile, then a 16-lane
                // row reduction per chunk (row_shr 8, 4, 2, 1: lane 15 of a DPP row ends up with the row total of rows 4c+lq+1)
                double dvT[NT];
#pragma unroll
                for (int T = 0; T < NT; T++) dvT[T] = sDv[16 * T + lc];
                // (a butterfly that folds the chunks into each other on the way down: at distance 8 a register keeps one chunk in
                //  the lower and another in the upper half of every DPP row, at distance 4 two such registers are folded again ...
                //  the last register holds a different row total in (almost) every lane: 9 merges of 7 instructions and 2 plain
                //  steps of 3 instead of 40 steps of 3, and ONE store instead of one masked store per chunk)
                static_assert(NC == 10 || NC == 12 || NC == 14, "the reduction tree below is written for 10, 12 or 14 row chunks");
                double pr[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    double a = 0.0;
#pragma unroll
                    for (int T = 0; 2 * T <= c; T++) a += chv[cidx(c, T)] * dvT[T];
                    pr[c] = a;
                }
                const bool h8 = lc & 8, h4 = lc & 4, h2 = lc & 2, h1 = lc & 1;
                // merge<CTRL>: lanes whose bit is clear keep `lo`'s partial sums, the others `hi`'s; the DPP pattern (mirror of
                // the row / the half row / the quad, swap of neighbours) brings the partner lane's share of the same chunk
#define TUM_MERGE(CTRL, BIT, lo, hi) (((BIT) ? (hi) : (lo)) + dpp0_f64<CTRL>((BIT) ? (lo) : (hi)))
#define TUM_FOLD(CTRL, v) ((v) + dpp0_f64<CTRL>(v))
                double R[NC / 2];
#pragma unroll
                for (int q = 0; q < NC / 2; q++) R[q] = TUM_MERGE(0x140, h8, pr[2 * q], pr[2 * q + 1]);
                const double S0 = TUM_MERGE(0x141, h4, R[0], R[1]), S1 = TUM_MERGE(0x141, h4, R[2], R[3]);
                double S2;
                if constexpr (NC >= 12) S2 = TUM_MERGE(0x141, h4, R[4], R[5]); else S2 = TUM_FOLD(0x141, R[4]);
                double U1;
                if constexpr (NC == 14) { const double S3 = TUM_FOLD(0x141, R[6]); U1 = TUM_MERGE(0x1b, h2, S2, S3); }      // (chunks 12, 13 in the h2 half of U1)
                else U1 = TUM_FOLD(0x1b, S2);
                const double U0 = TUM_MERGE(0x1b, h2, S0, S1);
                const double V = TUM_MERGE(0xb1, h1, U0, U1);
#undef TUM_MERGE
#undef TUM_FOLD
                // chunk whose row total this lane holds
                const int cl = h1 ? ((NC == 14 && h2) ? 12 : 8 + ((NC >= 12 && h4) ? 2 : 0)) + (h8 ? 1 : 0) : (h2 ? 4 : 0) + (h4 ? 2 : 0) + (h8 ? 1 : 0);
                wsync();
                sWh[4 * cl + lq] = V;
                wsync();
                if constexpr (SLOTS == 2) {
                    cdv[0] = gglane ? sWh[2 * gj] : xo;
                    cdv[1] = gglane ? sWh[2 * gj + 1] : pfx;
                } else {
                    cdv[0] = xo; cdv[1] = pfx; cdv[SLOTS - 1] = sWh[(lane < NMAX) ? lane : 0];
                }
            }
            double amax = 1.0, lmu = 0.0;
            double dcur[4][NS2];
#pragma unroll
            for (int rr = 0; rr < SLOTS; rr++)
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    const int k = rr * 2 + sd;
                    const bool on_ = on[rr];
                    const double eps = sd ? -1.0 : 1.0;
                    const double s_ = ROWF(0, k), t_ = ROWF(1, k), l_ = ROWF(2, k), m_ = ROWF(3, k);
                    double rc1 = t_ * l_, rc2 = s_ * m_;
                    // 1/s, 1/mu and 1/t, 1/lam from the reciprocals of the two complementarity products (two quarter-rate
                    // v_rcp_f64 per row side instead of four)
                    const double i1_ = frcp(rc1), i2_ = frcp(rc2);
                    const double is_ = m_ * i2_, im_ = s_ * i2_, il_ = t_ * i1_, it_ = l_ * i1_;
                    const double iDs = s_ * rD[k];
                    const double gam = l_ * rG[k];
                    if (pass == 1) { rc1 += cross1[k] - tau; rc2 += cross2[k] - tau; }
                    const double rsk = ROWF(4, k);
                    const double rho = -ROWF(5, k) + rc1 * il_ - (rsk + rc2 * is_) * iDs;
                    const double dl = -gam * (eps * cdv[rr] + rho);
                    const double dsl = (dl - rsk - rc2 * is_) * iDs;
                    const double dm = (-rc2 - m_ * dsl) * is_;
                    const double dtt = (-rc1 - t_ * dl) * il_;
                    // (the steps of a row side that does not exist are finite -- its state is all ones -- and never applied: its
                    //  step length is zero below; only what enters a wave reduction or the corrector is masked)
                    dcur[0][k] = dsl; dcur[1][k] = dtt; dcur[2][k] = dl; dcur[3][k] = dm;
                    double q = fmax(fmax(-dsl * is_, -dtt * it_), fmax(-dl * il_, -dm * im_));
                    q = on_ ? q : 0.0;
                    amax = fmax(amax, q);
                    if (pass == 0) { cross1[k] = on_ ? dtt * dl : 0.0; cross2[k] = on_ ? dsl * dm : 0.0; }
                }
            amax = frcp(wave_max(amax));
            if (pass == 0) {
#pragma unroll
                for (int k = 0; k < NS2; k++) {
                    const double pr_ = (ROWF(1, k) + amax * dcur[1][k]) * (ROWF(2, k) + amax * dcur[2][k])
                                     + (ROWF(0, k) + amax * dcur[0][k]) * (ROWF(3, k) + amax * dcur[3][k]);
                    lmu += on[k >> 1] ? pr_ : 0.0;
                }
                const double mu_aff = wave_sum(lmu) * inv_npairs;
                const double ratio = mu_aff * frcp(gap);
                sigma = ratio * ratio * ratio;
                if (amax < 0.1) {
#pragma unroll
                    for (int k = 0; k < NS2; k++) { cross1[k] = 0.0; cross2[k] = 0.0; }
                }
            } else {
                alpha = (amax >= 1.0) ? 1.0 : 0.995 * amax;
                if (alpha >= 1e-12) {
                    const double om_ = 1.0 - alpha;
#pragma unroll
                    for (int k = 0; k < NS2; k++) {
                        const double ak = on[k >> 1] ? alpha : 0.0;
#pragma unroll
                        for (int f = 0; f < 4; f++) ROWF(f, k) += ak * dcur[f][k];
                        ROWF(4, k) *= om_; ROWF(5, k) *= om_;
                    }
                }
            }
            TUM_TICK(7);
        }
        if (alpha < 1e-12) { qp_status = 2; break; }
        v0 += alpha * dv0; v1 += alpha * dv1;
        const double om = 1.0 - alpha;
        rv0 *= om; rv1 *= om;
        wsync();
    }
    const int status = acados_status(qp_status);
    res_stat = wave_max(res_stat); res_ineq = wave_max(res_ineq); res_comp = wave_max(res_comp);
    PIPE_LANE_DEFS
    TUM_TICK(8);
    // slack part of the cost and the slack outputs; the step of the inputs goes to the expansion kernel
    double cl = 0.0;
    {
#pragma unroll
        for (int rr = 0; rr < SLOTS; rr++)
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                const double sv = ROWF(0, rr * 2 + sd);
                const double c = pen(rr, sd, 0) * sv + 0.5 * pen(rr, sd, 1) * sv * sv;
                cl += on[rr] ? c : 0.0;
            }
        if (ka.slack) {
            double *sl = ka.slack + (size_t)b * 6 * N;
            // order: [lower: box_k (N) | (bx_k, h_k) k=1..N] then the same for upper
#pragma unroll
            for (int sd = 0; sd < 2; sd++)
#pragma unroll
                for (int rr = 0; rr < SLOTS; rr++) {
                    if (!on[rr]) continue;
                    const int idx = (ty[rr] == 0) ? stg[rr] : N + 2 * (stg[rr] - 1) + ((ty[rr] == 2) ? 1 : 0);
                    sl[sd * 3 * N + idx] = ROWF(0, 2 * rr + sd);
                    ka.qp_lam[(size_t)b * (6 * N + 2) + sd * 3 * N + idx] = ROWF(2, 2 * rr + sd);          // (the next solve's warm start)
                }
        }
    }
    const double scost = wave_sum(cl);
    if (!FUSE) {
        gvec[PV_DV + lane] = v0;
        if (lane < NB1) gvec[PV_DV + 64 + lane] = v1;
    }
    TUM_TICK(9);
    if (PROF && lane == 0)
        for (int i = 0; i < 12; i++) ka.prof[(size_t)b * 12 + i] = pacc[i];
    if (lane == 0) {
        if (!FUSE) gvec[PV_SC] = scost;
        ka.status[b] = status;
        ka.qp_iter[b] = it;
        ka.qp_status[b] = qp_status;
        ka.qp_lam[(size_t)b * (6 * N + 2) + 6 * N] = (qp_status == 0) ? 1.0 : 0.0;
        ka.res[b * 3 + 0] = res_stat; ka.res[b * 3 + 1] = res_ineq; ka.res[b * 3 + 2] = res_comp;
    }
    if constexpr (FUSE) {
        static_assert(D::E_LDS <= D::I_LDS - D::I_M, "the expansion works in the LDS of the factor");
        double *el = lds + D::I_M;
        wsync();
        el[D::E_DV + lane] = v0;
        if (lane < NB1) el[D::E_DV + 64 + lane] = v1;
        expand_instance<NT_, false, true>(pa, b, el, status, scost);
    }
}

}  // namespace tum
