// tum_nmpc.hip -- host side of libtumnmpc.so: the C-ABI declared in include/tum_nmpc.h.
// Owns device memory for `batch` OCP instances and launches the SQP-RTI pipeline (lin / cond / ipm / expand kernels).
// -DTUM_DEV_KERNELS (libtumnmpc_dev.so, tests and experiments only) adds the two other implementations of the solve the
// pipeline is held against: round 1's fused kernel and the four-wavefront interior point kernel.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tum_nmpc.h"
#include "pipe_kernels.hpp"
#ifdef TUM_DEV_KERNELS
#include "nmpc_kernel.hpp"
#include "ipm4_kernel.hpp"
#endif
#include "aux_kernels.hpp"
#include "loop_kernels.hpp"

using namespace tum;

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct tum_ocp {
    tum_ocp_desc d;
    int N, batch;
    hipStream_t stream; bool own_stream;
    hipEvent_t ev0, ev1;
    KArgs ka;
    double *dx0_own, *dyref_own;        // the capsule's own x0 / yref arrays; dx0 / dyref below are what is IN USE (tum_ocp_bind_device: the caller's)
    double *dX, *dU, *dx0, *dyref, *dW, *dpen, *dbnd, *dcost, *dres, *dslack, *dqpin, *ddbg, *dqplam;
    double *dWf;                       // full W per stage, [b][N+1][36] (allocated by the first cost_set 'W' with an off-diagonal entry; null: diagonal W)
    int *dstatus, *dqpiter, *dqpstatus, *dorder;
    bool lpt, order_valid;
    long long *dprof;
    double *dws, *dhws;
    int kmode;                     // 0 auto (= the pipeline), 1 fused, 2 pipeline, 3 pipeline with the four-wavefront interior point kernel
    unsigned epoch;                // bumped by everything a captured launch bakes into its kernel arguments (kernel variant, schedule,
                                   // SNMPC horizon / risk parameter / work buffers, R2 attachment): tum_sim_run re-captures its graph
    bool pipe;                     // this solve runs the four-kernel pipeline (resolved from kmode at launch)
    bool solved_pipe;              // the LAST solve ran the pipeline: its linearisation is in the stage records, not in qpin
    double *drec, *dcws, *dvec;    // pipeline workspace: stage records, gg rows in operand layout, q | d | dv
    hipEvent_t evi0, evi1;         // around the interior point kernel of the pipeline
    bool skip_ipm_events, ipm_timed;   // a step leaves them out (tum_ocp_step_async); was the LAST solve timed with them
    float last_ms;
    bool solved;
    std::vector<double> stage;     // host staging
    // coupled SNMPC OCP (tum_ocp_snmpc_attach)
    bool sn;
    SnArgs sa;
    double *dXS, *dxs0, *dApce, *dws2, *dpro, *ddv, *doffs;
    int *dxs_dirty; bool xs_lazy;          // sample copies of the stages > uph not yet frozen (snmpc_freeze_kernel)
    bool have_offs, fanout;        // sample initial conditions derived from the nominal x0 at every solve
    int sim_fork; bool lin_ahead;      // device closed loop: linearisation beside the planner (-1 the library's choice, 0 never, 1 where possible); state of one step
    int cond_wide;                     // condensing with six wavefronts per OCP: -1 the library's choice (at most one workgroup per CU), 0 never, 1 always
    int lin_cols;                      // linearisation with eight lanes per (instance, stage): -1 the library's choice (small batches), 0 never, 1 always (tum_ocp_set_kernel)
    int sn_prologue;                   // prologue of the SNMPC OCP: -1 the library's choice, 2 the matrix-core kernel, 0 column slots and passes (tum_ocp_set_kernel)
    // R2NMPC tightening after every solve (tum_ocp_r2_attach)
    bool r2; int r2_uph; double r2_dmin, r2_dmax, r2_uh; double *dr2S, *dr2B;
    // per-stage parameter vector of the SNMPC OCP as the caller last set it (tum_ocp_set "p")
    std::vector<double> hApce, p_gamma, p_stop; bool p_dirty; int uph_cap; size_t pro_cap; double gamma;
    // PCE matrix of the scenario fan-out (tum_pce_attach), snapshot of the bounds (tum_ocp_bounds_snapshot)
    double *dpceA; int pce_L, pce_S; double *dbnd_snap;
    // results on the host without a stream stall (tum_ocp_results_async / _wait): device slab of the packed summary, pinned
    // host slabs, the event behind the copies
    // two sets, used in turn: the request for the NEXT batch can be enqueued before the previous batch's results have been read
    double *dsum, *hsum[2], *hX[2], *hU[2]; hipEvent_t evres[2]; bool res_iter[2]; int res_head, res_count;
    double *hin[2];                    // pinned staging of x0 | yref of a step (tum_ocp_step_async), one per result slot
    unsigned long long *hts[2];        // device wall clock at the start / end of a step timed without events (pinned, one pair per result slot)
    int ts_slot; double ts_khz;        // slot whose clock pair times the LAST solve (-1: the events ev0 / ev1 do; 2: the synchronous slot below)
    // The reference's LITERAL call pattern on a small capsule (NMPC_class.py:169-206: N+1 x set yref, solve, 1 + N x get, get_cost,
    // 3 x get_stats -- every one a synchronous call): the per-step setters land in a pinned shadow of x0 | yref with a dirty map and go
    // up in ONE kernel in front of the next solve; a synchronous solve ends with ONE kernel that writes summary, X and U into pinned
    // slabs, and the getters that follow are served from those slabs until something changes the iterate. 83 calls of ~21 us each
    // (a stream synchronisation per call) become one solve and 82 host copies.
    double *hin_s; unsigned long long in_mask; bool in_x0, in_inflight; hipEvent_t ev_in;      // shadow of x0 | yref, dirty stages, upload in flight
    double *hsum_s, *hX_s, *hU_s; unsigned long long *hts_s; bool cache_valid;                 // results of the last synchronous solve
    double *hXS_s; bool xs_cached;                                                             // ... and the sample copies of an SNMPC capsule, read back on first use
    bool time_ipm;                     // keep the events around the interior point kernel also where the library leaves them out (tum_ocp_set_kernel "time-ipm")
};

static const int DBG_STRIDE = 20480;
static const int DBG_INST = 4;

extern "C" const char *tum_ocp_last_error(void) { return g_err.c_str(); }
extern "C" int tum_ocp_batch(const tum_ocp *c) { return c->batch; }
extern "C" int tum_ocp_horizon(const tum_ocp *c) { return c->N; }

// Every entry point that touches the device runs under the capsule's device and leaves the caller's (torch's) current
// device as it found it: two capsules on different GPUs may live in one process.
struct DevGuard {
    int prev = -1, dev; bool changed = false, ok = true;
    explicit DevGuard(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != d) { changed = hipSetDevice(d) == hipSuccess; ok = changed; }
    }
    ~DevGuard() { if (changed) (void)hipSetDevice(prev); }
};
// entry points that allocate or launch refuse to run on the caller's device when the switch failed
#define GUARD_OK(g) do { if (!(g).ok) return fail("hipSetDevice failed"); } while (0)

// temporary device buffer that cannot leak on an early error return
struct DevTmp {
    void *p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <typename T> T *as() const { return (T *)p; }
};

template <typename T>
static hipError_t dalloc(T **p, size_t n)
{
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(T));
    // hipMemset on device memory returns before the fill has run, and the fill runs on the NULL stream -- which the capsule's
    // non-blocking stream does not wait for. Without this wait the first solve after a large allocation races the fill: at
    // 65 536 instances the fill of the pipeline's 4.4 GB of workspace was still zeroing what the first eighth of the instances
    // had already handed from kernel to kernel (wrong first solves at batches >= 32 768; scripts/probes/large_batch_variants.py).
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e;
}

extern "C" tum_ocp *tum_ocp_create(const tum_ocp_desc *desc)
{
    if (!desc) { fail("null desc"); return nullptr; }
    if (desc->N < 1 || desc->N > TUM_N_MAX) { fail("N out of range (1..56)"); return nullptr; }
    if (desc->batch < 1) { fail("batch < 1"); return nullptr; }
    if (desc->nsub < 1 || !(desc->dt > 0)) { fail("bad nsub/dt"); return nullptr; }
    if (desc->n_ggv < 2 || desc->n_ggv > 16) { fail("n_ggv out of range (2..16)"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fail("no HIP device: libtumnmpc has no CPU fallback"); return nullptr; }
    if (desc->device < 0 || desc->device >= ndev) { fail("device ordinal out of range"); return nullptr; }
    DevGuard guard(desc->device); if (!guard.ok) { fail("hipSetDevice failed"); return nullptr; }
    tum_ocp *c = new tum_ocp();
    c->d = *desc; c->N = desc->N; c->batch = desc->batch; c->last_ms = 0; c->solved = false; c->epoch = 0;
    c->sn = false; c->dXS = c->dxs0 = c->dApce = c->dws2 = c->dpro = c->ddv = c->doffs = nullptr; c->dxs_dirty = nullptr; c->xs_lazy = false;
    c->have_offs = c->fanout = false; c->sn_prologue = -1; c->lin_cols = -1; c->cond_wide = -1; c->sim_fork = -1; c->lin_ahead = false;
    c->r2 = false; c->dr2S = c->dr2B = nullptr;
    c->p_dirty = false; c->uph_cap = 0; c->gamma = 0.0; c->dpceA = nullptr; c->pce_L = c->pce_S = 0; c->dbnd_snap = nullptr;
    c->dsum = nullptr; c->res_head = c->res_count = 0;
    c->hin_s = c->hsum_s = c->hX_s = c->hU_s = nullptr; c->hts_s = nullptr; c->in_mask = 0; c->in_x0 = c->in_inflight = false; c->ev_in = nullptr;
    c->cache_valid = false; c->time_ipm = false; c->hXS_s = nullptr; c->xs_cached = false;
    for (int i = 0; i < 2; i++) { c->hsum[i] = c->hX[i] = c->hU[i] = c->hin[i] = nullptr; c->hts[i] = nullptr; c->evres[i] = nullptr; c->res_iter[i] = false; }
    const int N = c->N; const size_t B = c->batch;
    bool ok = true;
    ok &= hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess; c->own_stream = true;
    ok &= hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
    ok &= dalloc(&c->dX, B * (N + 1) * NX) == hipSuccess;
    ok &= dalloc(&c->dU, B * N * NU) == hipSuccess;
    ok &= dalloc(&c->dx0, B * NX) == hipSuccess;
    ok &= dalloc(&c->dyref, B * (N + 1) * 6) == hipSuccess;
    c->dx0_own = c->dx0; c->dyref_own = c->dyref;
    ok &= dalloc(&c->dW, B * (size_t)(N + 1) * 6) == hipSuccess;          // diagonal of W per stage (stage N: the first 4 = W_e)
    ok &= dalloc(&c->dpen, B * 36) == hipSuccess;
    ok &= dalloc(&c->dbnd, B * 6 * (N + 1)) == hipSuccess;
    ok &= dalloc(&c->dcost, B) == hipSuccess;
    ok &= dalloc(&c->dres, B * 3) == hipSuccess;
    ok &= dalloc(&c->dslack, B * 6 * N) == hipSuccess;
    ok &= dalloc(&c->dstatus, B) == hipSuccess;
    ok &= dalloc(&c->dqpiter, B) == hipSuccess;
    ok &= dalloc(&c->dqpstatus, B) == hipSuccess;
    ok &= dalloc(&c->dorder, B) == hipSuccess;
    ok &= dalloc(&c->dqplam, B * (6 * (size_t)N + 2)) == hipSuccess;      // warm start of the interior point method: multipliers | converged flag
    { const char *e = getenv("TUM_NMPC_SCHEDULE"); c->lpt = !(e && std::string(e) == "natural"); }
    if (ok) {   // start from the identity map: the schedule is always a valid permutation
        std::vector<int> id(B);
        for (size_t i = 0; i < B; i++) id[i] = (int)i;
        ok &= hipMemcpy(c->dorder, id.data(), sizeof(int) * B, hipMemcpyHostToDevice) == hipSuccess;
        c->order_valid = true;
    }
    c->dqpin = nullptr;
#ifdef TUM_DEV_KERNELS      // (only the fused kernel writes A | B | b per stage; the pipeline's stage records hold them)
    if (desc->store_qp_in) ok &= dalloc(&c->dqpin, B * N * 88) == hipSuccess;
#endif
    ok &= dalloc(&c->ddbg, (size_t)DBG_STRIDE * DBG_INST) == hipSuccess;
    ok &= dalloc(&c->dprof, B * 12) == hipSuccess;
    c->dws = nullptr;
#ifdef TUM_DEV_KERNELS      // (linearisation records parked by the fused kernel during its interior point loop)
    ok &= dalloc(&c->dws, B * WS_DOUBLES) == hipSuccess;
#endif
    c->dhws = nullptr;
    { const char *e = getenv("TUM_NMPC_KERNEL"); const std::string k(e ? e : "auto"); c->kmode = (k == "pipeline") ? 2 : 0;
#ifdef TUM_DEV_KERNELS
      if (k == "fused") c->kmode = 1; else if (k == "pipeline4") c->kmode = 3;
#endif
      c->pipe = false; c->solved_pipe = false; }
    c->drec = c->dcws = c->dvec = nullptr; c->evi0 = c->evi1 = nullptr; c->skip_ipm_events = c->ipm_timed = false; c->ts_slot = -1; c->ts_khz = 0.0;
    ok &= hipEventCreate(&c->evi0) == hipSuccess && hipEventCreate(&c->evi1) == hipSuccess;
    if (!ok) { fail("device allocation failed"); tum_ocp_free(c); return nullptr; }

    KArgs &ka = c->ka;
    memset(&ka, 0, sizeof(ka));
    ka.N = N; ka.nsub = desc->nsub; ka.batch = c->batch; ka.flags = desc->store_qp_in ? 1 : 0; ka.dt = desc->dt;
    ka.iter_max = desc->qp_iter_max > 0 ? desc->qp_iter_max : 50;
    ka.tol_stat = desc->qp_tol_stat > 0 ? desc->qp_tol_stat : 1e-8;
    ka.tol_ineq = desc->qp_tol_ineq > 0 ? desc->qp_tol_ineq : 1e-8;
    ka.tol_comp = desc->qp_tol_comp > 0 ? desc->qp_tol_comp : 1e-8;
    ka.mu0 = desc->qp_mu0 > 0 ? desc->qp_mu0 : 0.05;
    ka.t0 = desc->qp_t0 > 0 ? desc->qp_t0 : 0.05;
    ka.reg = 0.0;
    // qp_solver_warm_start (SNMPC_acados_settings.py:307): the interior point method starts from the previous QP's multipliers
    ka.warm_mu = desc->qp_warm_start ? (desc->qp_warm_mu > 0 ? desc->qp_warm_mu : 1e-2) : 0.0;
    ka.qp_lam = c->dqplam;
    // (the gate of the warm start: constants of the oracle study, scripts/study/warm_gate.py; TUM_WARM_GATE=0 removes it -- development aid)
    // tum_ocp_desc.qp_warm_flips / qp_warm_viol override them (flips < 0: no gate)
    { const char *e = getenv("TUM_WARM_GATE");
      ka.warm_flips = (e && e[0] == '0') ? -1 : (desc->qp_warm_flips != 0 ? desc->qp_warm_flips : 16);
      ka.warm_viol = desc->qp_warm_viol > 0 ? desc->qp_warm_viol : 0.1; }
    Model &m = ka.mp;
    m.lf = desc->lf; m.lr = desc->lr; m.m = desc->m; m.inv_m = 1.0 / desc->m; m.inv_Iz = 1.0 / desc->Iz;
    m.ka = 0.5 * desc->ro * desc->S * desc->Cd;
    m.Bf = desc->Bf; m.Cf = desc->Cf; m.Df = desc->Df; m.Ef = desc->Ef;
    m.Br = desc->Br; m.Cr = desc->Cr; m.Dr = desc->Dr; m.Er = desc->Er;
    m.Fz_f = desc->m * desc->lr * desc->g / (desc->lf + desc->lr);
    m.Fz_r = desc->m * desc->lf * desc->g / (desc->lf + desc->lr);
    m.invFmax_f = 1.0 / std::sqrt(m.Fz_f * m.Fz_f + (desc->Cf * m.Fz_f) * (desc->Cf * m.Fz_f));
    m.invFmax_r = 1.0 / std::sqrt(m.Fz_r * m.Fz_r + (desc->Cr * m.Fz_r) * (desc->Cr * m.Fz_r));
    m.fr0 = desc->fr0; m.fr1 = desc->fr1; m.fr4 = desc->fr4;
    m.ax_brake = -desc->acc_min;
    m.n_ggv = desc->n_ggv;
    for (int i = 0; i < 16; i++) { m.ggv_v[i] = desc->ggv_v[i]; m.ggv_ax[i] = desc->ggv_ax[i]; m.ggv_ay[i] = desc->ggv_ay[i]; }
    ka.X = c->dX; ka.U = c->dU; ka.x0 = c->dx0; ka.yref = c->dyref; ka.W = c->dW; ka.pen = c->dpen; ka.bnd = c->dbnd;
    ka.cost = c->dcost; ka.res = c->dres; ka.slack = c->dslack;
    ka.status = c->dstatus; ka.qp_iter = c->dqpiter; ka.qp_status = c->dqpstatus;
    ka.qpin = c->dqpin; ka.dbg = c->ddbg; ka.dbg_stride = DBG_STRIDE; ka.prof = c->dprof; ka.ws = c->dws;

    if (hipFuncSetAttribute((const void *)ipm_kernel<false, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)ipm_kernel<true, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)ipm_kernel<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)ipm_kernel<false, 5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)ipm_kernel<false, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess
#ifdef TUM_DEV_KERNELS
        || hipFuncSetAttribute((const void *)ipm4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)ipm4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)nmpc_rti_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void *)nmpc_rti_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void *)nmpc_rti_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void *)nmpc_rti_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess
#endif
        ) {
        fail("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); tum_ocp_free(c); return nullptr;
    }
    return c;
}

extern "C" void tum_ocp_free(tum_ocp *c)
{
    if (!c) return;
    DevGuard guard(c->d.device);
    (void)hipFree(c->dX); (void)hipFree(c->dU); (void)hipFree(c->dx0_own); (void)hipFree(c->dyref_own); (void)hipFree(c->dW); if (c->dWf) (void)hipFree(c->dWf); (void)hipFree(c->dpen); (void)hipFree(c->dbnd);
    (void)hipFree(c->dqplam);
    (void)hipFree(c->dcost); (void)hipFree(c->dres); (void)hipFree(c->dslack); (void)hipFree(c->dstatus); (void)hipFree(c->dqpiter); (void)hipFree(c->dqpstatus); (void)hipFree(c->dorder);
    if (c->dqpin) (void)hipFree(c->dqpin);
    (void)hipFree(c->ddbg); (void)hipFree(c->dprof); (void)hipFree(c->dws); (void)hipFree(c->dhws); (void)hipFree(c->drec); (void)hipFree(c->dcws); (void)hipFree(c->dvec);
    if (c->evi0) (void)hipEventDestroy(c->evi0);
    if (c->evi1) (void)hipEventDestroy(c->evi1);
    (void)hipFree(c->dXS); (void)hipFree(c->dxs0); (void)hipFree(c->dApce); (void)hipFree(c->dws2); (void)hipFree(c->dpro); (void)hipFree(c->ddv); (void)hipFree(c->doffs); (void)hipFree(c->dxs_dirty);
    (void)hipFree(c->dr2S); (void)hipFree(c->dr2B); (void)hipFree(c->dpceA); (void)hipFree(c->dbnd_snap);
    (void)hipFree(c->dsum);
    if (c->hin_s) (void)hipHostFree(c->hin_s);
    if (c->hsum_s) (void)hipHostFree(c->hsum_s);
    if (c->hX_s) (void)hipHostFree(c->hX_s);
    if (c->hU_s) (void)hipHostFree(c->hU_s);
    if (c->hts_s) (void)hipHostFree(c->hts_s);
    if (c->hXS_s) (void)hipHostFree(c->hXS_s);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    for (int i = 0; i < 2; i++) {
        if (c->hsum[i]) (void)hipHostFree(c->hsum[i]);
        if (c->hin[i]) (void)hipHostFree(c->hin[i]);
        if (c->hts[i]) (void)hipHostFree(c->hts[i]);
        if (c->hX[i]) (void)hipHostFree(c->hX[i]);
        if (c->hU[i]) (void)hipHostFree(c->hU[i]);
        if (c->evres[i]) (void)hipEventDestroy(c->evres[i]);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// dynamic LDS (doubles) of the prologue kernel that `want` (a capsule's choice, < 0: the library's) selects for (uph, ns)
// (nsb: the compile-time bound of the kernels' sample / PCE-term sums: 16, or 32 where either count exceeds 16)
static int sn_nsb(int ns, int L) { return (ns > SN_B16 || L > SN_B16) ? SN_NSMAX : SN_B16; }
static size_t sn_prologue_lds(int uph, int ns, int want, int nsb = SN_B16)
{
    const int kind = sn_prologue_kind(uph, ns, want);
    if (kind == 2 && nsb == SN_B16) return sn_mfma_lds_doubles(uph, ns);
    return sn_prologue_lds_doubles(uph, ns, nsb == SN_B16 ? sn_prologue_variant(uph, ns) : 0, nsb);
}

// Turn the capsule into the coupled SNMPC OCP: the stacked state is the nominal copy followed by `ns`
// sample copies; A_pce (L x ns, row-major) and the uncertainty propagation horizon replace the per-stage parameter
// vector p = [A_pce.flatten(), risk_parameter, stop_flag] of the reference (SNMPC_class.py:103-104,124).
extern "C" int tum_ocp_snmpc_attach(tum_ocp *c, int ns, int L, const double *Apce, int uph, double gamma)
{
    if (!c || !Apce) return fail("null argument");
    if (c->sn) return fail("snmpc_attach: already attached");
    if (c->dWf) return fail("snmpc_attach: the capsule holds a full W; the coupled SNMPC OCP takes a diagonal one");
    if (ns < 1 || ns > SN_NSMAX) return fail("snmpc_attach: n_samples out of range (1..32)");
    if (L < 1 || L > SN_LMAX) return fail("snmpc_attach: number of PCE terms out of range (1..32)");
    if (uph < 0 || uph > c->N || uph > SN_UPHMAX) return fail("snmpc_attach: uncertainty propagation horizon out of range (0..N)");
    if (!(gamma > 0.0 && gamma <= 1.0)) return fail("snmpc_attach: gamma out of range (0,1]");
    if (c->d.nsub != 1) return fail("snmpc_attach: the SNMPC model is DISCRETE with one RK4 step per stage: create the capsule with nsub = 1");
    if (c->d.store_qp_in) return fail("snmpc_attach: store_qp_in is not available for the stacked state");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    {
        const size_t lds = sizeof(double) * (sn_prologue_lds(uph, ns, c->sn_prologue, sn_nsb(ns, L)));
        if (lds > 128 * 1024) return fail("snmpc_attach: n_samples x uph too large for the prologue kernel's LDS");
        HIPCHK(hipFuncSetAttribute((const void *)snmpc_prologue_mfma_kernel<SN_MFMA_NSW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)snmpc_prologue_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)snmpc_prologue_kernel<0, SN_NSMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)snmpc_prologue_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)snmpc_prologue_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)snmpc_prologue_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)snmpc_prologue_kernel<17>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    }
    const size_t B = c->batch; const int N = c->N;
    bool ok = true;
    ok &= dalloc(&c->dXS, B * (N + 1) * ns * NX) == hipSuccess;
    ok &= dalloc(&c->dxs0, B * ns * NX) == hipSuccess;
    ok &= dalloc(&c->dApce, (size_t)L * ns) == hipSuccess;
    ok &= dalloc(&c->dws2, B * (size_t)(uph > 0 ? uph : 1) * ns * (ABS + 5)) == hipSuccess;      // records, then the gg values / gradients
    ok &= dalloc(&c->dpro, B * (size_t)(uph > 0 ? uph : 1) * sn_pro_stage(uph)) == hipSuccess;
    ok &= dalloc(&c->ddv, B * NVP) == hipSuccess;
    ok &= dalloc(&c->doffs, (size_t)ns * NX) == hipSuccess;
    if (ok) { (void)hipFree(c->dxs_dirty); c->dxs_dirty = nullptr; ok &= hipMalloc((void **)&c->dxs_dirty, sizeof(int) * B) == hipSuccess; }
    if (!ok) return fail("snmpc_attach: device allocation failed");
    HIPCHK(hipMemset(c->dxs_dirty, 0, sizeof(int) * B)); HIPCHK(hipDeviceSynchronize()); c->xs_lazy = false;
    HIPCHK(hipMemcpy(c->dApce, Apce, sizeof(double) * L * ns, hipMemcpyHostToDevice));
    SnArgs &sa = c->sa;
    memset(&sa, 0, sizeof(sa));
    sa.N = N; sa.batch = c->batch; sa.ns = ns; sa.L = L; sa.uph = uph; sa.dt = c->ka.dt;
    sa.kappa = std::sqrt((1.0 - gamma) / gamma);      // SNMPC_acados_settings.py:187
    sa.mp = c->ka.mp;
    sa.X = c->dX; sa.U = c->dU; sa.XS = c->dXS; sa.xs0 = c->dxs0; sa.Apce = c->dApce; sa.ws2 = c->dws2; sa.pro = c->dpro;
    sa.gh = c->dws2 + B * (size_t)(uph > 0 ? uph : 1) * ns * ABS;
    sa.dv = c->ddv; sa.dv_stride = NVP; sa.status = c->dstatus; sa.xs_dirty = c->dxs_dirty;
    sa.dbg = c->ddbg + 20000;                            // tail of instance 0's dump area (tum_ocp_debug_dump), unused by the fused kernel
    c->ka.uph = uph; c->ka.pro = c->dpro; c->ka.dv = c->ddv;
    c->sn = true;
    c->hApce.assign(Apce, Apce + (size_t)L * ns);
    c->gamma = gamma; c->uph_cap = uph > 0 ? uph : 1; c->pro_cap = (size_t)(uph > 0 ? uph : 1) * sn_pro_stage(uph);
    c->p_gamma.assign(N + 1, gamma); c->p_stop.assign(N + 1, 0.0);
    for (int k = uph; k <= N; k++) c->p_stop[k] = 1.0;
    c->p_dirty = false;
    c->epoch++;
    return 0;
}

// acados_solver.set(stage, "p", [A_pce.flatten(), risk_parameter, stop_flag])   SNMPC_class.py:124,185,193.
// A_pce and the risk parameter are shared by all stages of the stacked model (the reference sends the same values to every
// stage); the stop flags must form the pattern the model is built for -- 0 on the stages < uph, 1 from stage uph on
// (SNMPC_class.py:103-104) -- and define the uncertainty propagation horizon. Applied at the next solve.
static int sn_set_p(tum_ocp *c, int stage, const double *v, int len, int nb, int stride)
{
    if (!c->sn) return fail("set p: not an SNMPC capsule (tum_ocp_snmpc_attach)");
    const int L = c->sa.L, ns = c->sa.ns, N = c->N;
    if (stage < 0 || stage > N) return fail("set p: stage out of range");
    if (len != L * ns + 2) return fail("set p: mismatching dimension for field \"p\" with dimension " + std::to_string(L * ns + 2) +
                                       " (you have " + std::to_string(len) + ")");
    if (stride != 0)
        for (int i = 1; i < nb; i++)
            if (memcmp(v, v + (size_t)i * stride, sizeof(double) * len) != 0) return fail("set p: the parameter vector is shared by all instances of the batch");
    const double g = v[L * ns], sf = v[L * ns + 1];
    if (!(g > 0.0 && g <= 1.0)) return fail("set p: risk parameter out of range (0,1]");
    if (sf != 0.0 && sf != 1.0) return fail("set p: stop_flag must be 0 or 1");
    if (memcmp(v, c->hApce.data(), sizeof(double) * L * ns) != 0) {
        DevGuard guard(c->d.device); GUARD_OK(guard);
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(c->dApce, v, sizeof(double) * L * ns, hipMemcpyHostToDevice));
        c->hApce.assign(v, v + (size_t)L * ns);
    }
    c->p_gamma[stage] = g; c->p_stop[stage] = sf;
    c->p_dirty = true;
    return 0;
}
// linearisation of the sample stages (K-S1), in front of the prologue kernel
static void sn_launch_lin(tum_ocp *c)
{
    const long long items = (long long)c->batch * c->sa.uph * c->sa.ns;
    if (items <= 0) return;
    // eight lanes per item while that still is one round of wavefronts on the chip (the same rule as launch_pipeline's for lin_cols_kernel)
    static const int cols_env = [] { const char *e = getenv("TUM_LIN_COLS"); return e ? atoi(e) : -1; }();
    const int want = (c->lin_cols >= 0) ? c->lin_cols : cols_env;
    if (want > 0 || (want < 0 && items * SLC_LANES <= 64LL * 1024))
        hipLaunchKernelGGL(snmpc_lin_cols_kernel, dim3((unsigned)((items + SLC_ITEMS - 1) / SLC_ITEMS)), dim3(64), 0, c->stream, c->sa);
    else hipLaunchKernelGGL(snmpc_lin_kernel, dim3((unsigned)((items + 63) / 64)), dim3(64), 0, c->stream, c->sa);
}

// the epilogue leaves the sample copies of the stages > uph for later (snmpc_epilogue_kernel); this brings them up to date
static int sn_materialise(tum_ocp *c)
{
    if (!c->sn || !c->xs_lazy) return 0;
    DevGuard guard(c->d.device); GUARD_OK(guard);
    hipLaunchKernelGGL(snmpc_freeze_kernel, dim3(c->batch), dim3(256), 0, c->stream, c->dXS, c->dxs_dirty, c->N, c->sa.ns, c->sa.uph, c->batch);
    HIPCHK(hipGetLastError());
    c->xs_lazy = false;
    return 0;
}

// the stacked state behind a synchronous solve of a small SNMPC capsule: the reference reads get(j, "x")[0:8] on every stage
// (SNMPC_class.py:205-209) -- the sample copies of all stages come back in ONE copy on the first such read
static int sn_cache_samples(tum_ocp *c)
{
    if (!c->sn || !c->cache_valid || c->xs_cached) return 0;
    if (sn_materialise(c)) return 1;
    DevGuard guard(c->d.device); GUARD_OK(guard);
    const size_t n = (size_t)c->batch * (c->N + 1) * c->sa.ns * NX;
    if (!c->hXS_s) HIPCHK(hipHostMalloc((void **)&c->hXS_s, sizeof(double) * n, hipHostMallocDefault));
    HIPCHK(hipMemcpyAsync(c->hXS_s, c->dXS, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->xs_cached = true;
    return 0;
}

// resolve the per-stage parameters into (uph, kappa) before a solve
static int sn_apply_p(tum_ocp *c)
{
    if (!c->p_dirty) return 0;
    const int N = c->N, ns = c->sa.ns;
    for (int k = 1; k <= N; k++)
        if (c->p_gamma[k] != c->p_gamma[0]) return fail("solve: the risk parameter p[-2] differs between stages (stage " + std::to_string(k) + ")");
    int uph = N + 1;
    for (int k = 0; k <= N; k++) if (c->p_stop[k] == 1.0) { uph = k; break; }
    for (int k = uph; k <= N; k++)
        if (c->p_stop[k] != 1.0) return fail("solve: stop_flag pattern not supported: it must be 0 on the stages < uph and 1 from stage uph on (stage " + std::to_string(k) + " is 0 after a 1)");
    if (uph > N) uph = N;                      // no stop flag at all: the samples are propagated over the whole horizon
    if (uph > SN_UPHMAX) return fail("solve: uncertainty propagation horizon from the stop flags exceeds " + std::to_string(SN_UPHMAX));
    if (sizeof(double) * (sn_prologue_lds(uph, ns, c->sn_prologue, sn_nsb(ns, c->sa.L))) > 128 * 1024)      // (the variant sn_launch_prologue will pick)
        return fail("solve: n_samples x uph too large for the prologue kernel's LDS");
    // (the hand-over buffer of the prologue is sized uph x sn_pro_stage(uph): its row pitch doubles beyond uph = 31)
    if (uph > c->uph_cap || (size_t)uph * sn_pro_stage(uph) > c->pro_cap) {
        DevGuard guard(c->d.device); GUARD_OK(guard);
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipFree(c->dws2); (void)hipFree(c->dpro); c->dws2 = c->dpro = nullptr;
        const size_t B = c->batch;
        c->pro_cap = (size_t)uph * sn_pro_stage(uph);
        if (dalloc(&c->dws2, B * (size_t)uph * ns * (ABS + 5)) != hipSuccess || dalloc(&c->dpro, B * c->pro_cap) != hipSuccess)
            return fail("solve: device allocation failed for the longer uncertainty propagation horizon");
        c->uph_cap = uph; c->sa.ws2 = c->dws2; c->sa.gh = c->dws2 + B * (size_t)uph * ns * ABS; c->sa.pro = c->dpro; c->ka.pro = c->dpro;
    }
    if (uph != c->sa.uph) {
        if (sn_materialise(c)) return 1;          // (the frozen copies belong to the horizon they were solved with)
        // the prologue only writes the live columns of a stage (2k+3 of them) and relies on the rest of its hand-over
        // buffers being zero; the per-instance stride of both buffers depends on uph, so a new horizon starts from zeros
        DevGuard guard(c->d.device); GUARD_OK(guard);
        HIPCHK(hipMemsetAsync(c->dws2, 0, sizeof(double) * (size_t)c->batch * c->uph_cap * ns * ABS, c->stream));
        HIPCHK(hipMemsetAsync(c->dpro, 0, sizeof(double) * (size_t)c->batch * c->pro_cap, c->stream));
    }
    c->gamma = c->p_gamma[0];
    c->sa.kappa = std::sqrt((1.0 - c->gamma) / c->gamma);
    c->sa.uph = uph; c->ka.uph = uph;
    c->p_dirty = false;
    c->epoch++;                                   // (uph, kappa and the work buffers are kernel arguments of a captured launch)
    return 0;
}
extern "C" int tum_ocp_snmpc_samples(const tum_ocp *c) { return (c && c->sn) ? c->sa.ns : 0; }

// Offsets of the sample initial conditions from the nominal one (compute_x0dist, stochastic_mpc_utils.py:78-91:
// row s = x0 + stds * w_s), ns x 8 (host). Once registered, an 8-value lbx_0 / ubx_0 (and the device closed loop, whose
// state estimator writes the nominal x0) fans out to the sample copies on the device at every solve.
extern "C" int tum_ocp_snmpc_set_offsets(tum_ocp *c, const double *offs)
{
    if (!c || !offs) return fail("null argument");
    if (!c->sn) return fail("snmpc_set_offsets: not an SNMPC capsule (tum_ocp_snmpc_attach)");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    HIPCHK(hipMemcpy(c->doffs, offs, sizeof(double) * c->sa.ns * NX, hipMemcpyHostToDevice));
    c->have_offs = true;
    return 0;
}
static int sn_fanout(tum_ocp *c)
{
    const int n = c->batch * c->sa.ns * NX;
    hipLaunchKernelGGL(snmpc_fanout_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->dxs0, c->dx0, c->doffs, c->sa.ns, c->batch);
    HIPCHK(hipGetLastError());
    return 0;
}

static int chk_range(tum_ocp *c, int b0, int nb)
{
    if (!c) return fail("null capsule");
    if (b0 < 0 || nb < 1 || b0 + nb > c->batch) return fail("instance range out of bounds");
    return 0;
}

// the summary (and, for small batches, the iterate) is written by the packing kernel STRAIGHT into the pinned slab; TUM_RESULTS_ZERO_COPY=0
// (development aid) goes through a device slab and copy commands instead
static bool results_zero_copy()
{
    static const bool zc = [] { const char *e = getenv("TUM_RESULTS_ZERO_COPY"); return !(e && e[0] == '0'); }();
    return zc;
}
// does a results request of this capsule run as ONE kernel that writes summary, X and U into the pinned slabs (pack_results_kernel:
// the kernel that also reads the device clock at the end of a clocked step, tum_ocp_step_async)
static bool results_pack_all(const tum_ocp *c, int with_iterate)
{
    return results_zero_copy() && with_iterate && (size_t)c->batch * (size_t)(c->N + 1) * NX <= 32768;
}

// ---- the pinned shadow of the per-step inputs (x0 | yref) of a SMALL capsule
static bool small_inputs(const tum_ocp *c) { return (size_t)c->batch * (NX + (size_t)(c->N + 1) * 6) <= 32768 && c->N + 1 <= 64; }
static int shadow_ready(tum_ocp *c)
{
    if (!c->hin_s) {
        HIPCHK(hipHostMalloc((void **)&c->hin_s, sizeof(double) * (size_t)c->batch * (NX + (size_t)(c->N + 1) * 6), hipHostMallocDefault));
        // (a stage-N reference has four entries: the two behind it in its six-double record are uploaded with it and must not be garbage)
        memset(c->hin_s, 0, sizeof(double) * (size_t)c->batch * (NX + (size_t)(c->N + 1) * 6));
        HIPCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    }
    // the upload kernel of an asynchronous solve may still be reading the shadow
    if (c->in_inflight) { HIPCHK(hipEventSynchronize(c->ev_in)); c->in_inflight = false; }
    return 0;
}
// host records (batch x len, `stride` apart; 0 = one record for all) into the shadow: x0 (stage < 0) or the yref record of a stage
static int shadow_write(tum_ocp *c, int stage, const double *v, int len, int stride)
{
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (shadow_ready(c)) return 1;
    const size_t B = c->batch, rec = (size_t)(c->N + 1) * 6;
    for (size_t b = 0; b < B; b++) {
        const double *src = v + b * (size_t)stride;
        double *dst = (stage < 0) ? c->hin_s + b * NX : c->hin_s + B * NX + b * rec + (size_t)stage * 6;
        memcpy(dst, src, sizeof(double) * len);
    }
    if (stage < 0) c->in_x0 = true; else c->in_mask |= 1ull << stage;
    return 0;
}
// pending setters -> device, one kernel on the capsule's stream (ts: the device clock at its start goes to ts[0])
static int flush_inputs(tum_ocp *c, unsigned long long *ts = nullptr, bool force = false)
{
    if (!c->in_x0 && !c->in_mask && !force) return 0;
    const size_t B = c->batch;
    hipLaunchKernelGGL(stage_in_masked_kernel, dim3(4), dim3(256), 0, c->stream, c->hin_s, c->dx0, (int)(B * NX), c->dyref, (int)(B * (c->N + 1) * 6),
                       c->N + 1, c->in_x0 ? 1 : 0, c->in_mask, ts);
    HIPCHK(hipGetLastError());
    if (c->in_x0 || c->in_mask) { HIPCHK(hipEventRecord(c->ev_in, c->stream)); c->in_inflight = true; }
    c->in_x0 = false; c->in_mask = 0;
    return 0;
}

// strided scatter: host records (nb x len, `stride` apart; stride 0 = broadcast) -> device rows
static int put(tum_ocp *c, double *dbase, size_t rec, size_t off, const double *v, int len, int b0, int nb, int stride)
{
    if (stride != 0 && stride < len) return fail("stride < len");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if ((dbase == c->dx0 || dbase == c->dyref) && flush_inputs(c)) return 1;          // (older setters still in the shadow go first)
    if (dbase == c->dX || dbase == c->dU) c->cache_valid = false;
    const double *src = v;
    size_t spitch = (size_t)stride * sizeof(double);
    if (stride == 0) {
        c->stage.assign((size_t)nb * len, 0.0);
        for (int i = 0; i < nb; i++) memcpy(&c->stage[(size_t)i * len], v, sizeof(double) * len);
        src = c->stage.data(); spitch = (size_t)len * sizeof(double);
    }
    HIPCHK(hipMemcpy2DAsync(dbase + (size_t)b0 * rec + off, rec * sizeof(double), src, spitch,
                            (size_t)len * sizeof(double), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
static int fetch(tum_ocp *c, const double *dbase, size_t rec, size_t off, double *v, int len, int b0, int nb, int stride)
{
    if (stride < len) return fail("stride < len");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    HIPCHK(hipMemcpy2DAsync(v, (size_t)stride * sizeof(double), dbase + (size_t)b0 * rec + off, rec * sizeof(double),
                            (size_t)len * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int tum_ocp_set(tum_ocp *c, int stage, const char *field, const double *v, int len, int b0, int nb, int stride)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!field || !v) return fail("null argument");
    const int N = c->N;
    const std::string f(field);
    if (f == "x") {
        if (stage == TUM_ALL_STAGES) { if (len != (N + 1) * NX) return fail("set x: len != (N+1)*8"); return put(c, c->dX, (N + 1) * NX, 0, v, len, b0, nb, stride); }
        if (stage < 0 || stage > N) return fail("set x: stage out of range");
        if (c->sn && len == NX * (c->sa.ns + 1)) {   // stacked state: nominal copy, then the sample copies (SNMPC_class.py:126-127)
            if (stride != 0 && stride < len) return fail("stride < len");
            const int ns = c->sa.ns;
            // (a write to stage uph would change what the deferred freeze copies into the stages behind it: freeze first)
            if (stage >= c->sa.uph && sn_materialise(c)) return 1;
            if (put(c, c->dX, (N + 1) * NX, (size_t)stage * NX, v, NX, b0, nb, stride)) return 1;
            return put(c, c->dXS, (size_t)(N + 1) * ns * NX, (size_t)stage * ns * NX, v + NX, ns * NX, b0, nb, stride);
        }
        if (len != NX) return fail("set x: mismatching dimension, expected 8");
        return put(c, c->dX, (N + 1) * NX, (size_t)stage * NX, v, len, b0, nb, stride);
    }
    if (f == "u") {
        if (stage == TUM_ALL_STAGES) { if (len != N * NU) return fail("set u: len != N*2"); return put(c, c->dU, N * NU, 0, v, len, b0, nb, stride); }
        if (stage < 0 || stage >= N) return fail("set u: stage out of range");
        if (len != NU) return fail("set u: mismatching dimension, expected 2");
        return put(c, c->dU, N * NU, (size_t)stage * NU, v, len, b0, nb, stride);
    }
    if (f == "yref") {
        if (stage == TUM_ALL_STAGES) { if (len != (N + 1) * 6) return fail("set yref: len != (N+1)*6"); return put(c, c->dyref, (N + 1) * 6, 0, v, len, b0, nb, stride); }
        if (stage < 0 || stage > N) return fail("set yref: stage out of range");
        const int want = (stage < N) ? TUM_NY : TUM_NYE;
        if (len != want) return fail("set yref: mismatching dimension for this stage");
        // the reference sets one stage per call, N + 1 calls per control step (NMPC_class.py:169-180): into the shadow, up with the solve
        if (b0 == 0 && nb == c->batch && small_inputs(c) && (stride == 0 || stride >= len)) return shadow_write(c, stage, v, len, stride);
        return put(c, c->dyref, (N + 1) * 6, (size_t)stage * 6, v, len, b0, nb, stride);
    }
    if (f == "p") {
        if (b0 != 0 || nb != c->batch) return fail("set p: the parameter vector is shared by all instances (b0 = 0, nb = batch)");
        return sn_set_p(c, stage, v, len, nb, stride);
    }
    return fail("set: unknown field '" + f + "'");
}

extern "C" int tum_ocp_get(tum_ocp *c, int stage, const char *field, double *v, int len, int b0, int nb, int stride)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!field || !v) return fail("null argument");
    const int N = c->N;
    const std::string f(field);
    // after a synchronous solve of a small capsule the iterate is in the capsule's pinned slabs (tum_ocp_solve): the N + 1 getters the
    // reference issues per control step (NMPC_class.py:193-198) are host copies
    auto cached = [&](const double *slab, size_t rec, size_t off) {
        if (stride < len) return fail("stride < len");
        for (int i = 0; i < nb; i++) memcpy(v + (size_t)i * stride, slab + (size_t)(b0 + i) * rec + off, sizeof(double) * len);
        return 0;
    };
    if (f == "x") {
        if (stage == TUM_ALL_STAGES) {
            if (len != (N + 1) * NX) return fail("get x: len");
            if (c->cache_valid) return cached(c->hX_s, (size_t)(N + 1) * NX, 0);
            return fetch(c, c->dX, (N + 1) * NX, 0, v, len, b0, nb, stride);
        }
        if (c->sn && stage >= 0 && stage <= N && len == NX * (c->sa.ns + 1)) {
            const int ns = c->sa.ns;
            if (stride < len) return fail("stride < len");
            if (sn_cache_samples(c)) return 1;
            if (c->cache_valid && c->hXS_s) {          // the stacked state of every stage, read back ONCE after the solve
                for (int i = 0; i < nb; i++) {
                    memcpy(v + (size_t)i * stride, c->hX_s + ((size_t)(b0 + i) * (N + 1) + stage) * NX, sizeof(double) * NX);
                    memcpy(v + (size_t)i * stride + NX, c->hXS_s + ((size_t)(b0 + i) * (N + 1) + stage) * ns * NX, sizeof(double) * ns * NX);
                }
                return 0;
            }
            if (stage > c->sa.uph && sn_materialise(c)) return 1;
            if (fetch(c, c->dX, (N + 1) * NX, (size_t)stage * NX, v, NX, b0, nb, stride)) return 1;
            return fetch(c, c->dXS, (size_t)(N + 1) * ns * NX, (size_t)stage * ns * NX, v + NX, ns * NX, b0, nb, stride);
        }
        if (stage < 0 || stage > N || len != NX) return fail("get x: bad stage/len");
        if (c->cache_valid) return cached(c->hX_s, (size_t)(N + 1) * NX, (size_t)stage * NX);
        return fetch(c, c->dX, (N + 1) * NX, (size_t)stage * NX, v, len, b0, nb, stride);
    }
    if (f == "u") {
        if (stage == TUM_ALL_STAGES) {
            if (len != N * NU) return fail("get u: len");
            if (c->cache_valid) return cached(c->hU_s, (size_t)N * NU, 0);
            return fetch(c, c->dU, N * NU, 0, v, len, b0, nb, stride);
        }
        if (stage < 0 || stage >= N || len != NU) return fail("get u: bad stage/len");
        if (c->cache_valid) return cached(c->hU_s, (size_t)N * NU, (size_t)stage * NU);
        return fetch(c, c->dU, N * NU, (size_t)stage * NU, v, len, b0, nb, stride);
    }
    if (f == "sl" || f == "su") {
        // acados order per stage: stage 0 [sbu], stages 1..N-1 [sbu,sbx,sh], stage N [sbx,sh]
        const int want = (stage == 0) ? 1 : (stage == N ? 2 : 3);
        if (stage < 0 || stage > N || len != want) return fail("get sl/su: bad stage/len");
        std::vector<double> all((size_t)nb * 6 * N);
        if (fetch(c, c->dslack, 6 * N, 0, all.data(), 6 * N, b0, nb, 6 * N)) return 1;
        const int side = (f == "su") ? 3 * N : 0;
        for (int i = 0; i < nb; i++) {
            const double *s = &all[(size_t)i * 6 * N + side];
            double *o = v + (size_t)i * stride; int n = 0;
            if (stage < N) o[n++] = s[stage];
            if (stage >= 1) { o[n++] = s[N + 2 * (stage - 1)]; o[n++] = s[N + 2 * (stage - 1) + 1]; }
        }
        return 0;
    }
    return fail("get: unknown field '" + f + "'");
}

extern "C" int tum_ocp_constraints_set(tum_ocp *c, int stage, const char *field, const double *v, int len, int b0, int nb, int stride)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!field || !v) return fail("null argument");
    const int N = c->N, NB = N + 1;
    const std::string f(field);
    if (stage < 0 || stage > N) return fail("constraints_set: stage out of range");
    if (f == "lbx" || f == "ubx") {
        if (stage == 0) {   // x0 equality: lbx_0 = ubx_0 = x0 (NMPC_class.py:243-246)
            if (c->sn && len == NX * (c->sa.ns + 1)) {   // x0 of all copies (SNMPC_class.py:262-264)
                if (stride != 0 && stride < len) return fail("stride < len");
                c->fanout = false;
                if (put(c, c->dx0, NX, 0, v, NX, b0, nb, stride)) return 1;
                return put(c, c->dxs0, (size_t)c->sa.ns * NX, 0, v + NX, c->sa.ns * NX, b0, nb, stride);
            }
            if (len != NX) return fail("constraints_set lbx/ubx at stage 0: expected 8 values (x0)");
            if (c->sn) {
                if (!c->have_offs) return fail("constraints_set lbx/ubx at stage 0: an SNMPC capsule takes 8 (n_samples+1) values, or 8 after tum_ocp_snmpc_set_offsets");
                c->fanout = true;
            }
            // (set twice per control step, lbx and ubx: NMPC_class.py:243-246)
            if (b0 == 0 && nb == c->batch && small_inputs(c) && (stride == 0 || stride >= len)) return shadow_write(c, -1, v, len, stride);
            return put(c, c->dx0, NX, 0, v, len, b0, nb, stride);
        }
        if (len != 1) return fail("constraints_set lbx/ubx: expected 1 value (steering angle)");
        return put(c, c->dbnd, 6 * NB, (size_t)(f == "lbx" ? 2 : 3) * NB + stage, v, 1, b0, nb, stride);
    }
    if (f == "lbu" || f == "ubu") {
        if (stage >= N) return fail("constraints_set lbu/ubu: stage out of range");
        if (len != 1) return fail("constraints_set lbu/ubu: expected 1 value (steering rate)");
        return put(c, c->dbnd, 6 * NB, (size_t)(f == "lbu" ? 0 : 1) * NB + stage, v, 1, b0, nb, stride);
    }
    if (f == "lh" || f == "uh") {
        if (stage == 0) return fail("constraints_set lh/uh: no nonlinear constraint at stage 0 (nh_0 = 0)");
        if (len != 1) return fail("constraints_set lh/uh: expected 1 value");
        return put(c, c->dbnd, 6 * NB, (size_t)(f == "lh" ? 4 : 5) * NB + stage, v, 1, b0, nb, stride);
    }
    return fail("constraints_set: unknown field '" + f + "'");
}

extern "C" int tum_ocp_cost_set(tum_ocp *c, int stage, const char *field, const double *v, int len, int b0, int nb, int stride)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!field || !v) return fail("null argument");
    const int N = c->N;
    const std::string f(field);
    if ((stage < 0 || stage > N) && !(stage == TUM_ALL_STAGES && f == "W")) return fail("cost_set: stage out of range");
    if (f == "W") {
        // per stage, like acados (NMPC_class.py:294-296 sets every stage in a loop); stage == TUM_ALL_STAGES: one 6 x 6 W for all
        // the stages 0..N-1 in one call
        const bool all = stage == TUM_ALL_STAGES;
        const int ny = (all || stage < N) ? TUM_NY : TUM_NYE;
        if (len != ny * ny) return fail("cost_set W: mismatching dimension");
        const int cnt = stride == 0 ? 1 : nb;
        const int rep = all ? N : 1;
        std::vector<double> diag((size_t)cnt * rep * ny);
        bool offdiag = false;
        for (int i = 0; i < cnt; i++) {
            const double *Wm = v + (size_t)i * stride;
            for (int r = 0; r < ny; r++)
                for (int q = 0; q < ny; q++) {
                    if (r == q) { for (int k = 0; k < rep; k++) diag[((size_t)i * rep + k) * ny + r] = Wm[r * ny + r]; }
                    else if (Wm[q * ny + r] != 0.0) offdiag = true;
                }
        }
        // A W with off-diagonal entries (acados takes any matrix, NMPC_class.py:290-296; the reference installs diagonal ones): from the first such
        // call on the capsule keeps a full symmetric 6 x 6 per stage (its symmetric part: the cost only sees that) beside the diagonal, and the
        // pipeline condenses with the full-W instantiation of the six-wavefront condensing kernel. Nominal / R2 OCP through the pipeline only.
        if (offdiag && !c->dWf) {
            if (c->N > 48) return fail("cost_set W: horizons beyond 48 take a diagonal W (the full-W condensing kernel's row store does not fit there)");
            if (c->sn) return fail("cost_set W: the coupled SNMPC OCP takes a diagonal W (its cost rows come from the prologue kernel)");
            DevGuard guard(c->d.device); GUARD_OK(guard);
            HIPCHK(hipStreamSynchronize(c->stream));
            if (dalloc(&c->dWf, (size_t)c->batch * (N + 1) * 36) != hipSuccess) return fail("cost_set W: device allocation failed");
            const size_t n = (size_t)c->batch * (N + 1);
            hipLaunchKernelGGL(wf_from_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->dW, c->dWf, (long long)n);
            HIPCHK(hipGetLastError());
            c->ka.Wf = c->dWf; c->epoch++;
        }
        if (c->dWf) {
            std::vector<double> full((size_t)cnt * rep * 36, 0.0);
            for (int i = 0; i < cnt; i++) {
                const double *Wm = v + (size_t)i * stride;
                for (int k = 0; k < rep; k++)
                    for (int r = 0; r < ny; r++)
                        for (int q = 0; q < ny; q++) full[((size_t)i * rep + k) * 36 + r * 6 + q] = 0.5 * (Wm[q * ny + r] + Wm[r * ny + q]);
            }
            if (put(c, c->dWf, (size_t)(N + 1) * 36, all ? 0 : (size_t)stage * 36, full.data(), rep * 36, b0, nb, stride == 0 ? 0 : rep * 36)) return 1;
        }
        return put(c, c->dW, (size_t)(N + 1) * 6, all ? 0 : (size_t)stage * 6, diag.data(), rep * ny, b0, nb, stride == 0 ? 0 : rep * ny);
    }
    int which = -1;
    if (f == "zl") which = 0; else if (f == "zu") which = 1; else if (f == "Zl") which = 2; else if (f == "Zu") which = 3;
    if (which < 0) return fail("cost_set: unknown field '" + f + "'");
    // penalty classes: 0 = stage 0 [sbu], 1 = stages 1..N-1 [sbu,sbx,sh], 2 = stage N [sbx,sh]
    const int cls = (stage == 0) ? 0 : (stage < N ? 1 : 2);
    const int want = (cls == 0) ? 1 : (cls == 1 ? 3 : 2);
    if (len != want) return fail("cost_set " + f + ": mismatching dimension for this stage");
    const int slot0 = (cls == 2) ? 1 : 0;
    for (int j = 0; j < want; j++) {
        const int slot = slot0 + j;
        // element (cls, slot, which) of pen[b][3][3][4]; one strided put per slot
        const int cnt = stride == 0 ? 1 : nb;
        std::vector<double> col(cnt);
        for (int i = 0; i < cnt; i++) col[i] = v[(size_t)i * stride + j];
        if (put(c, c->dpen, 36, (size_t)(cls * 3 + slot) * 4 + which, col.data(), 1, b0, nb, stride == 0 ? 0 : 1)) return 1;
    }
    return 0;
}

// MFMA tiles of condensed variables this capsule's pipeline runs with: five (N <= 40), six (41..48), seven (49..56).
// (TUM_FORCE_TILES: development aid -- a larger instantiation at a horizon a smaller one covers: the padding variables must not change the answer.)
static int tiles_of(const tum_ocp *c)
{
    static const int force = [] { const char *e = getenv("TUM_FORCE_TILES"); return e ? atoi(e) : 0; }();
    const int need = c->N > 48 ? 7 : (c->N > NMAX ? 6 : 5);
    return (force > need && force <= 7 && !c->sn) ? force : need;
}
// workspaces of the kernel variants, allocated when a variant is first used
static int ensure_workspace(tum_ocp *c)
{
    const size_t B = c->batch;
    const int nt = tiles_of(c);          // horizons beyond 40: six MFMA tiles (PD<6>), beyond 48: seven (PD<7>)
    const size_t ntt = nt == 7 ? PD<7>::NTT : nt == 6 ? PD<6>::NTT : PD<5>::NTT, nch = nt == 7 ? PD<7>::NCH : nt == 6 ? PD<6>::NCH : PD<5>::NCH,
                 pvec = nt == 7 ? PD<7>::PVEC : nt == 6 ? PD<6>::PVEC : PD<5>::PVEC;
    if (c->pipe && !c->dhws) {
        if (dalloc(&c->dhws, B * ntt * 256) != hipSuccess) return fail("workspace allocation failed (H tiles)");
    }
    if (c->pipe && !c->drec) {
        if (dalloc(&c->drec, B * (size_t)(c->N + 1) * PREC) != hipSuccess || dalloc(&c->dcws, B * nch * 64) != hipSuccess ||
            dalloc(&c->dvec, B * pvec) != hipSuccess) return fail("workspace allocation failed (pipeline)");
    }
    return 0;
}

// "fused": one kernel per solve; "pipeline": linearise / condense / interior point / expand as four kernels, each at its own
// occupancy, handing over through the L2-resident workspace. Since the pipeline's interior point kernel has the factorisation
// without LDS round trips it is the faster one at every batch size (device time per solve 0.334 against 0.346 ms at one
// instance, 1.51 against 1.71 ms at 4096; wall time of a solve() call likewise): "auto" (default) runs the pipeline; the fused
// kernel stays selectable and is what the condensed-QP debug dump runs. The coupled SNMPC OCP follows the same rule
// (prologue / epilogue around either).
extern "C" int tum_ocp_set_kernel(tum_ocp *c, const char *name)
{
    if (!c || !name) return fail("null argument");
    const std::string n(name);
    if (n == "time-ipm") { c->time_ipm = true; return 0; }            // (timing options, not kernels: the events around the interior point kernel on EVERY solve)
    if (n == "no-time-ipm") { c->time_ipm = false; return 0; }
    if (n == "auto") { c->kmode = 0; c->lin_cols = -1; c->cond_wide = -1; c->sim_fork = -1; }        // (the wide kernels of the latency path: the library decides by batch size again)
    else if (n == "pipeline") c->kmode = 2;
    // the prologue of the coupled SNMPC OCP: the matrix-core kernel (default where n_samples <= 10) or the column-slot / pass variants
    // the linearisation: one lane per (instance, stage) or eight (default: eight while the batch is one round of wavefronts)
    else if (n == "loop-serial") c->sim_fork = 0;
    else if (n == "loop-fork") c->sim_fork = 1;
    else if (n == "cond-one-wavefront") c->cond_wide = 0;
    else if (n == "cond-six-wavefronts") c->cond_wide = 1;
    else if (n == "lin-lane-per-stage") c->lin_cols = 0;
    else if (n == "lin-eight-lanes") c->lin_cols = 1;
    else if (n == "prologue-passes" || n == "prologue-mfma") {
        if (c->sn) {
            const int want = (n == "prologue-passes") ? 0 : 2;
            const size_t lds = sizeof(double) * sn_prologue_lds(c->sa.uph, c->sa.ns, want, sn_nsb(c->sa.ns, c->sa.L));
            if (lds > 128 * 1024) return fail("set_kernel: n_samples x uph too large for that prologue kernel's LDS");
        }
        c->sn_prologue = (n == "prologue-passes") ? 0 : 2;
    }
#ifdef TUM_DEV_KERNELS
    else if (n == "fused") c->kmode = 1;
    else if (n == "pipeline4") c->kmode = 3;
#else
    else if (n == "fused" || n == "pipeline4")
        return fail("set_kernel: kernel '" + n + "' exists in the development build only (libtumnmpc_dev.so); this library is the pipeline");
#endif
    else return fail("set_kernel: unknown kernel '" + n + "' (auto | pipeline | lin-lane-per-stage | lin-eight-lanes | cond-one-wavefront | cond-six-wavefronts | loop-fork | loop-serial | prologue-mfma | prologue-passes; development build: fused | pipeline4)");
    c->epoch++;
    return 0;
}

// which kernel variant this solve runs, and its workspace (allocated on first use; never inside a stream capture)
static int resolve_kernel(tum_ocp *c)
{
#ifdef TUM_DEV_KERNELS
    c->pipe = !(c->ka.flags & 2) && c->kmode != 1;
    if (c->dWf && c->kmode == 1) return fail("solve: a full W (cost_set 'W' with off-diagonal entries) runs on the pipeline only, not on the development kernel 'fused'");
    if (c->ka.warm_mu > 0.0 && (c->kmode == 1 || c->kmode == 3))
        return fail("solve: the development kernels 'fused' / 'pipeline4' always cold-start the interior point method: create the capsule with qp_warm_start = 0");
    if (c->N > NMAX) {      // the fused kernel covers N <= 40; longer horizons exist as a pipeline instantiation only
        if (c->ka.flags & 2) return fail("debug_dump: the condensed-QP dump is built for N <= 40");
        if (c->kmode == 1) return fail("solve: kernel 'fused' is built for N <= 40 (use 'auto' or 'pipeline')");
        c->pipe = true;
    }
#else
    c->pipe = true;
#endif
    return ensure_workspace(c);
}

// the prologue of the coupled SNMPC OCP: column state of its recursions in registers (instantiation by the number of passes of the
// last stage) or, for short propagation horizons and beyond the largest instantiation, in LDS
static void sn_launch_prologue(tum_ocp *c)
{
    if (sn_nsb(c->sa.ns, c->sa.L) != SN_B16) {      // more than 16 samples or PCE terms: the column-slot kernel, column state in LDS, 32-wide bounds
        const size_t lds32 = sizeof(double) * sn_prologue_lds_doubles(c->sa.uph, c->sa.ns, 0, SN_NSMAX);
        hipLaunchKernelGGL((snmpc_prologue_kernel<0, SN_NSMAX>), dim3(c->batch), dim3(64), lds32, c->stream, c->sa);
        return;
    }
    const int kind = sn_prologue_kind(c->sa.uph, c->sa.ns, c->sn_prologue);
    if (kind == 2) {      // the column recursions on the matrix cores, the samples split over the two wavefronts of a workgroup
        const size_t lds = sizeof(double) * sn_mfma_lds_doubles(c->sa.uph, c->sa.ns);
        // (one wavefront with all ten samples instead: 178 spilled registers, 3.6 against 2.55 ms per solve at UPH = Tp)
        hipLaunchKernelGGL((snmpc_prologue_mfma_kernel<SN_MFMA_NSW, 2>), dim3(c->batch), dim3(128), lds, c->stream, c->sa);
        return;
    }
    const int v = sn_prologue_variant(c->sa.uph, c->sa.ns);
    const size_t lds = sizeof(double) * sn_prologue_lds_doubles(c->sa.uph, c->sa.ns, v);
    const dim3 g(c->batch), blk(64);
    switch (v) {
    case 6: hipLaunchKernelGGL(snmpc_prologue_kernel<6>, g, blk, lds, c->stream, c->sa); break;
    case 9: hipLaunchKernelGGL(snmpc_prologue_kernel<9>, g, blk, lds, c->stream, c->sa); break;
    case 13: hipLaunchKernelGGL(snmpc_prologue_kernel<13>, g, blk, lds, c->stream, c->sa); break;
    case 17: hipLaunchKernelGGL(snmpc_prologue_kernel<17>, g, blk, lds, c->stream, c->sa); break;
    default: hipLaunchKernelGGL(snmpc_prologue_kernel<0>, g, blk, lds, c->stream, c->sa); break;
    }
}

// do the wide kernels of the latency path run for this capsule (host's choice by batch size, tum_ocp_set_kernel, environment)
static bool use_lin_cols(const tum_ocp *c)
{
    static const int cols_env = [] { const char *e = getenv("TUM_LIN_COLS"); return e ? atoi(e) : -1; }();
    const int want = (c->lin_cols >= 0) ? c->lin_cols : cols_env;
    return want > 0 || (want < 0 && (long long)c->batch * (c->N + 1) * LC_LANES <= 64LL * 1024);
}
static bool use_cond_wide(const tum_ocp *c)
{
    static const int wide_env = [] { const char *e = getenv("TUM_COND_WIDE"); return e ? atoi(e) : -1; }();
    const int want = (c->cond_wide >= 0) ? c->cond_wide : wide_env;
    if (tiles_of(c) == 7) return false;          // (seven tiles: the six-wavefront kernel's row store does not fit a CU's LDS beside a full W; one wavefront per OCP at every batch size)
    return want > 0 || (want < 0 && c->batch <= 256) || c->dWf != nullptr;          // (a full W exists as an instantiation of this kernel only)
}
// The device closed loop (tum_sim_run) can run the linearisation of a solve BESIDE the planner of the same control step: the
// Runge-Kutta pass needs the iterate, not the reference -- only the four residuals of the cost do, and cond_wide_kernel forms those
// while it loads the records (flags & 8). Nominal OCP on the latency path only (lin_cols_kernel + cond_wide_kernel).
static bool lin_ahead_ok(const tum_ocp *c)
{
    static const int fork_env = [] { const char *e = getenv("TUM_SIM_FORK"); return e ? atoi(e) : -1; }();
    const int want = (c->sim_fork >= 0) ? c->sim_fork : fork_env;
    // (off unless asked for: measured SLOWER -- 0.169 against 0.160 ms per control step at 26 vehicles, 0.158-0.162 against 0.156 at one:
    //  the two cross-stream dependencies of a step cost more than the 15 us of planner the linearisation hides behind; HISTORY.md (round-4 document, section 7))
    return want > 0 && c->pipe && !c->sn && !(c->ka.flags & 6) && use_lin_cols(c) && use_cond_wide(c);
}
static void launch_lin_ahead(tum_ocp *c, hipStream_t st)
{
    PArgs pa;
    pa.ka = c->ka; pa.rec = c->drec; pa.hws = c->dhws; pa.cws = c->dcws; pa.vec = c->dvec;
    pa.ka.flags |= 8;
    const long long items = (long long)c->batch * (c->N + 1);
    hipLaunchKernelGGL(lin_cols_kernel<false>, dim3((unsigned)((items + LC_ITEMS - 1) / LC_ITEMS)), dim3(64), 0, st, pa);
    c->lin_ahead = true;          // the next launch_pipeline skips its linearisation and tells the condensing kernel (consumed there)
}

static int launch_pipeline(tum_ocp *c, bool events)
{
    PArgs pa;
    pa.ka = c->ka; pa.rec = c->drec; pa.hws = c->dhws; pa.cws = c->dcws; pa.vec = c->dvec;
    const bool prof = (c->ka.flags & 4) != 0;
    const long long items = (long long)c->batch * (c->N + 1);
    // linearisation: eight lanes per item while that still is one round of wavefronts on the chip (256 CUs x 4 SIMDs), see lin_cols_kernel
    const bool cols = use_lin_cols(c);
    const dim3 g_cols((unsigned)((items + LC_ITEMS - 1) / LC_ITEMS)), g_lane((unsigned)((items + 63) / 64));
    const bool lin_done = c->lin_ahead;
    c->lin_ahead = false;
    if (lin_done) pa.ka.flags |= 8;          // (launch_lin_ahead ran it on another stream; the caller has joined that stream)
    else if (c->sn) {   // coupled SNMPC OCP: sample fan-out and prologue first, the QP solution goes to the epilogue through the workspace
        if (c->fanout && sn_fanout(c)) return 1;
        sn_launch_lin(c);
        sn_launch_prologue(c);
        if (cols) hipLaunchKernelGGL(lin_cols_kernel<true>, g_cols, dim3(64), 0, c->stream, pa);
        else hipLaunchKernelGGL(lin_kernel<true>, g_lane, dim3(64), 0, c->stream, pa);
    } else {
        if (cols) hipLaunchKernelGGL(lin_cols_kernel<false>, g_cols, dim3(64), 0, c->stream, pa);
        else hipLaunchKernelGGL(lin_kernel<false>, g_lane, dim3(64), 0, c->stream, pa);
    }
    // (development aid: a larger LDS request lowers the number of OCPs that share a CU)
    // The expansion as the tail of the interior point kernel pays where a batch is at most one round of resident wavefronts (one
    // launch less: 0.424 against 0.432 ms per solve() call at 26 instances, 0.457 against 0.469 at 1024); beyond that its
    // loads run at the interior point kernel's occupancy -- one wavefront per SIMD, four OCPs per CU -- and hold that slot:
    // 3.72 against 3.96 M solves/s on config 2 (three streams). TUM_FUSED_EXPAND=0 / 1 forces it off / on (development aid).
    static const int fuse_env = [] { const char *e = getenv("TUM_FUSED_EXPAND"); return e ? atoi(e) : -1; }();
    const bool no_fuse = fuse_env == 0 || (fuse_env < 0 && c->batch > 1024);
    static const int lds_req = [] { const char *e = getenv("TUM_IPM_LDS"); const int v = e ? atoi(e) : 0; return (v > 0 && v <= 64 * 1024) ? v : 0; }();
    auto rest = [&](auto ntc) {
        constexpr int NTv = decltype(ntc)::value;
        const int ipm_lds = lds_req > PD<NTv>::I_LDS_BYTES ? lds_req : PD<NTv>::I_LDS_BYTES;
        {
            // six wavefronts per OCP while every OCP can have a CU's LDS to itself (cond_wide_kernel)
            const bool wide = use_cond_wide(c);
            if constexpr (NTv == 7) {      // N = 49..56: a diagonal W, one wavefront per OCP at every batch size
                if (c->sn && 2 * c->sa.uph <= c->N) hipLaunchKernelGGL((cond_kernel<NTv, true, true>), dim3(c->batch), dim3(64), 0, c->stream, pa);
                else if (c->sn) hipLaunchKernelGGL((cond_kernel<NTv, true, false>), dim3(c->batch), dim3(64), 0, c->stream, pa);
                else hipLaunchKernelGGL((cond_kernel<NTv, false>), dim3(c->batch), dim3(64), 0, c->stream, pa);
            } else {
            if (wide && c->sn) hipLaunchKernelGGL((cond_wide_kernel<NTv, true>), dim3(c->batch), dim3(64 * cw_waves<NTv>()), 0, c->stream, pa);
            else if (wide && c->dWf) hipLaunchKernelGGL((cond_wide_kernel<NTv, false, true>), dim3(c->batch), dim3(64 * cw_waves<NTv>()), 0, c->stream, pa);
            else if (wide) hipLaunchKernelGGL((cond_wide_kernel<NTv, false>), dim3(c->batch), dim3(64 * cw_waves<NTv>()), 0, c->stream, pa);
            // (coupled SNMPC: the register form of the stage record pays behind stage uph and costs in front of it, pipe_kernels.hpp)
            else if (c->sn && 2 * c->sa.uph <= c->N) hipLaunchKernelGGL((cond_kernel<NTv, true, true>), dim3(c->batch), dim3(64), 0, c->stream, pa);
            else if (c->sn) hipLaunchKernelGGL((cond_kernel<NTv, true, false>), dim3(c->batch), dim3(64), 0, c->stream, pa);
            else hipLaunchKernelGGL((cond_kernel<NTv, false>), dim3(c->batch), dim3(64), 0, c->stream, pa);
            }
        }
        if ((events && !c->skip_ipm_events) || c->time_ipm) (void)hipEventRecord(c->evi0, c->stream);
        bool expanded = false;
#ifdef TUM_DEV_KERNELS
        if (prof && c->kmode == 3 && NTv == 5) hipLaunchKernelGGL((ipm4_kernel<true>), dim3(c->batch), dim3(256), I4::BYTES, c->stream, pa);
        else if (!prof && c->kmode == 3 && NTv == 5) hipLaunchKernelGGL((ipm4_kernel<false>), dim3(c->batch), dim3(256), I4::BYTES, c->stream, pa);
        else
#endif
        // the nominal OCP: the expansion runs as the tail of the interior point kernel (ipm_kernel<., ., true>); the instrumented
        // instantiation and the coupled SNMPC OCP keep the expansion kernel
        if constexpr (NTv == 5) {      // (the instrumented instantiation exists for the five-tile build only)
            if (prof) hipLaunchKernelGGL((ipm_kernel<true, 5>), dim3(c->batch), dim3(64), ipm_lds, c->stream, pa);
            else if (c->sn || no_fuse) hipLaunchKernelGGL((ipm_kernel<false, 5>), dim3(c->batch), dim3(64), ipm_lds, c->stream, pa);
            else { hipLaunchKernelGGL((ipm_kernel<false, 5, true>), dim3(c->batch), dim3(64), ipm_lds, c->stream, pa); expanded = true; }
        } else if constexpr (NTv == 7) {      // (seven tiles: the expansion stays a kernel of its own)
            hipLaunchKernelGGL((ipm_kernel<false, NTv>), dim3(c->batch), dim3(64), ipm_lds, c->stream, pa);
        } else {
            if (c->sn || no_fuse) hipLaunchKernelGGL((ipm_kernel<false, NTv>), dim3(c->batch), dim3(64), ipm_lds, c->stream, pa);
            else { hipLaunchKernelGGL((ipm_kernel<false, NTv, true>), dim3(c->batch), dim3(64), ipm_lds, c->stream, pa); expanded = true; }
        }
        if ((events && !c->skip_ipm_events) || c->time_ipm) (void)hipEventRecord(c->evi1, c->stream);
        if (c->sn) {
            // the epilogue steps the sample copies AND the nominal copy of the stages 1..uph (their PCE mean); the expansion
            // kernel behind it takes the nominal recursion from stage uph to the end of the horizon and evaluates the cost
            SnArgs sa = c->sa;
            sa.dv = c->dvec + PD<NTv>::PV_DV; sa.dv_stride = PD<NTv>::PVEC;
            sa.Xn = c->dX; sa.dxu = c->dvec + PD<NTv>::PV_SC + 8;
            if (sn_nsb(sa.ns, sa.L) != SN_B16) hipLaunchKernelGGL((snmpc_epilogue_kernel<SN_NSMAX>), dim3(c->batch), dim3(64), 0, c->stream, sa);
            else hipLaunchKernelGGL((snmpc_epilogue_kernel<>), dim3(c->batch), dim3(64), 0, c->stream, sa);
            c->xs_lazy = true;
            hipLaunchKernelGGL((expand_kernel<NTv, true>), dim3(c->batch), dim3(64), 0, c->stream, pa);
        } else if (!expanded) hipLaunchKernelGGL((expand_kernel<NTv, false>), dim3(c->batch), dim3(64), 0, c->stream, pa);
    };
    { const int nt = tiles_of(c); if (nt == 7) rest(std::integral_constant<int, 7>()); else if (nt == 6) rest(std::integral_constant<int, 6>()); else rest(std::integral_constant<int, 5>()); }
    return 0;
}

static int launch(tum_ocp *c, bool events = true)
{
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (resolve_kernel(c)) return 1;
    c->cache_valid = false; c->xs_cached = false;
    if (flush_inputs(c)) return 1;          // setters still in the pinned shadow (small capsules)
    if (events) HIPCHK(hipEventRecord(c->ev0, c->stream));
    // longest-first schedule from the previous solve's iteration counts (only matters when the batch is more than one
    // round of resident wavefronts)
    c->ka.order = (c->lpt && c->order_valid && c->batch > 1024) ? c->dorder : nullptr;
    // the instrumented instantiation carries the phase timers (flag 4) and the debug dump (flag 2)
    if (c->sn && sn_apply_p(c)) return 1;
#ifdef TUM_DEV_KERNELS
    const bool prof = (c->ka.flags & 6) != 0;
    auto fused = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(c->batch), dim3(64), LDS_BYTES, c->stream, c->ka); };
    if (c->sn && !c->pipe && sn_nsb(c->sa.ns, c->sa.L) != SN_B16)
        return fail("solve: kernel 'fused' takes at most 16 samples / PCE terms (use 'auto' or 'pipeline')");
    if (c->sn && !c->pipe && c->sa.uph > SN_UPHMAX_FUSED)
        return fail("solve: kernel 'fused' reads the sample columns of one wavefront: uncertainty propagation horizon <= 31 (use 'auto' or 'pipeline')");
    if (c->pipe) { if (launch_pipeline(c, events)) return 1; }
    else if (c->sn) {
        if (c->fanout && sn_fanout(c)) return 1;
        sn_launch_lin(c);
        sn_launch_prologue(c);
        if (prof) fused(nmpc_rti_kernel<true, true>); else fused(nmpc_rti_kernel<false, true>);
        hipLaunchKernelGGL((snmpc_epilogue_kernel<>), dim3(c->batch), dim3(64), 0, c->stream, c->sa); c->xs_lazy = true;
    }
    else { if (prof) fused(nmpc_rti_kernel<true>); else fused(nmpc_rti_kernel<false>); }
    if (!c->pipe) HIPCHK(hipMemsetAsync(c->dqplam, 0, sizeof(double) * (size_t)c->batch * (6 * (size_t)c->N + 2), c->stream));      // (the fused kernel leaves no multipliers behind)
#else
    if (launch_pipeline(c, events)) return 1;
#endif
    HIPCHK(hipGetLastError());
    if (c->r2) {   // constraint tightening for the NEXT solve from this one's linearisation (skipped per instance on failure)
        hipLaunchKernelGGL(r2_backoff_kernel, dim3((c->batch + 3) / 4), dim3(256), 0, c->stream, c->dqpin, c->dX, c->dbnd, c->ka.mp,
                           c->dr2S, c->dr2B, c->N, c->r2_uph, c->batch, c->r2_dmin, c->r2_dmax, c->r2_uh, (double *)nullptr, c->dstatus,
                           c->pipe ? c->drec : (const double *)nullptr, PREC);
        HIPCHK(hipGetLastError());
    }
    if (events) HIPCHK(hipEventRecord(c->ev1, c->stream));
    if (c->lpt && c->batch > 1024) {
        hipLaunchKernelGGL(lpt_order_kernel, dim3(1), dim3(1024), 0, c->stream, c->dqpiter, c->dorder, c->batch);
        HIPCHK(hipGetLastError());
        c->order_valid = true;
    }
    c->solved = true;
    c->solved_pipe = c->pipe;
    c->ipm_timed = (events && !c->skip_ipm_events) || c->time_ipm;
    c->ts_slot = -1;          // (tum_ocp_step_async sets it behind this call)
    return 0;
}

// 1: dispatch instances longest-first using the previous solve's iteration counts (default), 0: natural order
extern "C" int tum_ocp_set_schedule(tum_ocp *c, int longest_first)
{
    if (!c) return fail("null capsule");
    if (c->lpt != (longest_first != 0)) c->epoch++;
    c->lpt = longest_first != 0;
    return 0;
}

extern "C" int tum_ocp_solve_async(tum_ocp *c)
{
    if (!c) return fail("null capsule");
    return launch(c);
}
extern "C" int tum_ocp_synchronize(tum_ocp *c)
{
    if (!c) return fail("null capsule");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int tum_ocp_solve(tum_ocp *c)
{
    if (!c) { fail("null capsule"); return -1; }
    DevGuard guard(c->d.device); if (!guard.ok) { fail("hipSetDevice failed"); return -1; }
    if (small_inputs(c) && results_pack_all(c, 1)) {
        // small capsule: [pending setters up + device clock] -> solve -> [summary, X, U into pinned slabs + device clock], ONE wait. No
        // event and no copy command on the stream; the getters that follow read the slabs (cache_valid).
        const size_t B = c->batch; const int N = c->N;
        auto hm = [&](auto **pp, size_t bytes) { return *pp || hipHostMalloc((void **)pp, bytes, hipHostMallocDefault) == hipSuccess; };
        if (shadow_ready(c)) return -1;
        if (!hm(&c->hsum_s, sizeof(double) * B * 5) || !hm(&c->hX_s, sizeof(double) * B * (N + 1) * NX) || !hm(&c->hU_s, sizeof(double) * B * N * NU) ||
            !hm(&c->hts_s, 2 * sizeof(unsigned long long))) { fail("solve: pinned allocation failed"); return -1; }
        if (c->ts_khz == 0.0) { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->d.device) != hipSuccess) khz = 0; c->ts_khz = khz > 0 ? (double)khz : 1e5; }
        if (flush_inputs(c, c->hts_s, true)) return -1;
        c->skip_ipm_events = true;
        const int rc = launch(c, false);
        c->skip_ipm_events = false;
        if (rc) return -1;
        hipLaunchKernelGGL(pack_results_kernel, dim3(8), dim3(256), 0, c->stream, c->dX, c->dU, c->dcost, c->dstatus, c->dqpiter, N, (int)B,
                           c->hsum_s, c->hX_s, c->hU_s, c->hts_s);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { fail("kernel execution failed"); return -1; }
        c->in_inflight = false;
        c->ts_slot = 2; c->cache_valid = true; c->xs_cached = false;
        int mx = 0;
        for (size_t b = 0; b < B; b++) { const int st = (int)c->hsum_s[b * 5 + 3]; if (st > mx) mx = st; }
        return mx;
    }
    if (launch(c)) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) { fail("kernel execution failed"); return -1; }
    std::vector<int> st(c->batch);
    if (hipMemcpy(st.data(), c->dstatus, sizeof(int) * c->batch, hipMemcpyDeviceToHost) != hipSuccess) { fail("status copy failed"); return -1; }
    int mx = 0;
    for (int s : st) if (s > mx) mx = s;
    return mx;
}

extern "C" double tum_ocp_last_kernel_ms(tum_ocp *c)
{
    if (!c || !c->solved) return 0.0;
    DevGuard guard(c->d.device); if (!guard.ok) return 0.0;
    if (c->ts_slot == 2) {      // the synchronous solve of a small capsule: clocked by its first and last kernel, already waited for
        c->last_ms = (float)((double)(c->hts_s[1] - c->hts_s[0]) / c->ts_khz);
        return c->last_ms;
    }
    if (c->ts_slot >= 0) {      // a step timed by the device's wall clock (tum_ocp_step_async): valid once its results have been waited for
        const int w = c->ts_slot;
        if (c->evres[w] && hipEventSynchronize(c->evres[w]) != hipSuccess) return 0.0;
        const unsigned long long t0 = c->hts[w][0], t1 = c->hts[w][1];
        c->last_ms = (float)((double)(t1 - t0) / c->ts_khz);
        return c->last_ms;
    }
    if (hipEventSynchronize(c->ev1) != hipSuccess) return 0.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return 0.0;
    c->last_ms = ms;
    return ms;
}

extern "C" int tum_ocp_get_cost(tum_ocp *c, double *out, int b0, int nb)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!out) return fail("null argument");
    if (c->cache_valid) { for (int i = 0; i < nb; i++) out[i] = c->hsum_s[(size_t)(b0 + i) * 5 + 2]; return 0; }
    DevGuard guard(c->d.device); GUARD_OK(guard);
    HIPCHK(hipStreamSynchronize(c->stream));        // (the copy below runs on the NULL stream, which the capsule's non-blocking stream is not ordered with)
    HIPCHK(hipMemcpy(out, c->dcost + b0, sizeof(double) * nb, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int tum_ocp_get_stats(tum_ocp *c, const char *field, void *out, int b0, int nb)
{
    if (!c || !field || !out) return fail("null argument");
    const std::string f(field);
    if (f == "time_tot") { *(double *)out = tum_ocp_last_kernel_ms(c) * 1e-3; return 0; }
    if (f == "time_ipm") {    // pipeline only: device seconds of the interior point kernel of the last solve
        if (!c->solved || !c->solved_pipe || !c->ipm_timed)
            return fail("get_stats time_ipm: pipeline kernel only, after a solve() / solve_async() (a step, and the synchronous solve of a small capsule, leave these "
                        "events out unless tum_ocp_set_kernel(c, \"time-ipm\") asked for them)");
        DevGuard guard(c->d.device); GUARD_OK(guard);
        float ms = 0;
        if (hipEventSynchronize(c->evi1) != hipSuccess || hipEventElapsedTime(&ms, c->evi0, c->evi1) != hipSuccess) return fail("get_stats time_ipm: no timing");
        *(double *)out = ms * 1e-3; return 0;
    }
    if (chk_range(c, b0, nb)) return 1;
    if (f == "sqp_iter") { int *o = (int *)out; for (int i = 0; i < nb; i++) o[i] = 1; return 0; }
    if (c->cache_valid && (f == "qp_iter" || f == "status")) {
        int *o = (int *)out; const int col = (f == "status") ? 3 : 4;
        for (int i = 0; i < nb; i++) o[i] = (int)c->hsum_s[(size_t)(b0 + i) * 5 + col];
        return 0;
    }
    DevGuard guard(c->d.device); GUARD_OK(guard);
    HIPCHK(hipStreamSynchronize(c->stream));        // (after tum_ocp_solve_async: the copies below are on the NULL stream)
    if (f == "qp_iter") { HIPCHK(hipMemcpy(out, c->dqpiter + b0, sizeof(int) * nb, hipMemcpyDeviceToHost)); return 0; }
    if (f == "status") { HIPCHK(hipMemcpy(out, c->dstatus + b0, sizeof(int) * nb, hipMemcpyDeviceToHost)); return 0; }
    if (f == "qp_status") { HIPCHK(hipMemcpy(out, c->dqpstatus + b0, sizeof(int) * nb, hipMemcpyDeviceToHost)); return 0; }
    if (f == "res") { HIPCHK(hipMemcpy(out, c->dres + (size_t)b0 * 3, sizeof(double) * 3 * nb, hipMemcpyDeviceToHost)); return 0; }
    return fail("get_stats: unknown field '" + f + "'");
}

extern "C" int tum_ocp_reset(tum_ocp *c)
{
    if (!c) return fail("null capsule");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    c->cache_valid = false;
    HIPCHK(hipMemsetAsync(c->dqplam, 0, sizeof(double) * (size_t)c->batch * (6 * (size_t)c->N + 2), c->stream));
    HIPCHK(hipMemsetAsync(c->dX, 0, sizeof(double) * (size_t)c->batch * (c->N + 1) * NX, c->stream));
    HIPCHK(hipMemsetAsync(c->dU, 0, sizeof(double) * (size_t)c->batch * c->N * NU, c->stream));
    if (c->sn) HIPCHK(hipMemsetAsync(c->dXS, 0, sizeof(double) * (size_t)c->batch * (c->N + 1) * c->sa.ns * NX, c->stream));
    if (c->sn) { HIPCHK(hipMemsetAsync(c->dxs_dirty, 0, sizeof(int) * (size_t)c->batch, c->stream)); c->xs_lazy = false; }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int tum_ocp_cold_start(tum_ocp *c)
{
    if (!c) return fail("null capsule");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    c->cache_valid = false;
    if (flush_inputs(c)) return 1;          // (the x0 it copies may still be in the pinned shadow)
    hipLaunchKernelGGL(cold_start_kernel, dim3(c->batch), dim3(64), 0, c->stream, c->dX, c->dU, c->dx0, c->N, c->batch, c->dqplam);
    if (c->sn && c->fanout && sn_fanout(c)) return 1;
    if (c->sn) hipLaunchKernelGGL(snmpc_cold_start_kernel, dim3(c->batch), dim3(256), 0, c->stream, c->dXS, c->dxs0, c->N, c->sa.ns, c->batch);
    if (c->sn) { HIPCHK(hipMemsetAsync(c->dxs_dirty, 0, sizeof(int) * (size_t)c->batch, c->stream)); c->xs_lazy = false; }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int tum_ocp_get_from_qp_in(tum_ocp *c, int stage, const char *field, double *out, int len, int b0, int nb, int stride)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!c->d.store_qp_in) return fail("get_from_qp_in: capsule created without store_qp_in");
    if (!c->solved) return fail("get_from_qp_in: no solve yet");
    if (stage < 0 || stage >= c->N) return fail("get_from_qp_in: stage out of range");
    const std::string f(field ? field : "");
    int off, n, rows, cols;
    if (f == "A") { off = 0; n = 64; rows = 8; cols = 8; }
    else if (f == "B") { off = 64; n = 16; rows = 8; cols = 2; }
    else if (f == "b") { off = 80; n = 8; rows = 8; cols = 1; }
    else return fail("get_from_qp_in: unknown field '" + f + "'");
    if (len != n || stride < n) return fail("get_from_qp_in: bad len/stride");
    if (c->solved_pipe) {   // the pipeline keeps the linearisation as compact stage records (pipe_kernels.hpp): expand on the host
        std::vector<double> r((size_t)nb * PREC);
        if (fetch(c, c->drec, (size_t)(c->N + 1) * PREC, (size_t)stage * PREC, r.data(), PREC, b0, nb, PREC)) return 1;
        const double dt = c->ka.dt;
        for (int i = 0; i < nb; i++) {
            const double *q = &r[(size_t)i * PREC];
            double *o = out + (size_t)i * stride;
            for (int e = 0; e < n; e++) o[e] = 0.0;
            if (f == "A") {           // column-major 8 x 8
                for (int d : {0, 1, 2, 6, 7}) o[d * 8 + d] = 1.0;
                o[2 * 8 + 0] = q[0]; o[2 * 8 + 1] = q[1];
                for (int ri = 0; ri < 6; ri++) for (int cc = 0; cc < 5; cc++) o[(3 + cc) * 8 + ri] = q[2 + ri * 7 + cc];
            } else if (f == "B") {    // column-major 8 x 2
                for (int ri = 0; ri < 6; ri++) { o[0 * 8 + ri] = q[2 + ri * 7 + 5]; o[1 * 8 + ri] = q[2 + ri * 7 + 6]; }
                o[1 * 8 + 6] = dt; o[0 * 8 + 7] = dt;
            } else for (int ri = 0; ri < 8; ri++) o[ri] = q[44 + ri];
        }
        return 0;
    }
    std::vector<double> tmp((size_t)nb * n);
    if (fetch(c, c->dqpin, (size_t)c->N * 88, (size_t)stage * 88 + off, tmp.data(), n, b0, nb, n)) return 1;
    for (int i = 0; i < nb; i++)          // device keeps row-major, acados hands out column-major
        for (int r = 0; r < rows; r++)
            for (int q = 0; q < cols; q++) out[(size_t)i * stride + q * rows + r] = tmp[(size_t)i * n + r * cols + q];
    return 0;
}

extern "C" int tum_ocp_set_stream(tum_ocp *c, void *hip_stream)
{
    if (!c) return fail("null capsule");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    // whatever is still pending on the old stream (asynchronous uploads, a bounds snapshot, a solve) must not be overtaken
    // by work enqueued on the new one
    if (c->stream) HIPCHK(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream; c->own_stream = false;
    return 0;
}

extern "C" int tum_ocp_get_device(tum_ocp *c, const char *field, void *dst, int b0, int nb)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!field || !dst) return fail("null argument");
    const int N = c->N;
    const std::string f(field);
    DevGuard guard(c->d.device); GUARD_OK(guard);
    hipStream_t s = c->stream;
    if (f == "summary") {
        hipLaunchKernelGGL(pack_summary_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, c->dU, c->dcost, c->dstatus, c->dqpiter, N, b0, nb, (double *)dst);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (f == "u0") { HIPCHK(hipMemcpy2DAsync(dst, 2 * 8, c->dU + (size_t)b0 * N * NU, (size_t)N * NU * 8, 2 * 8, nb, hipMemcpyDeviceToDevice, s)); return 0; }
    if (f == "x1") { HIPCHK(hipMemcpy2DAsync(dst, 8 * 8, c->dX + (size_t)b0 * (N + 1) * NX + NX, (size_t)(N + 1) * NX * 8, 8 * 8, nb, hipMemcpyDeviceToDevice, s)); return 0; }
    if (f == "cost") { HIPCHK(hipMemcpyAsync(dst, c->dcost + b0, 8 * (size_t)nb, hipMemcpyDeviceToDevice, s)); return 0; }
    if (f == "X") { HIPCHK(hipMemcpyAsync(dst, c->dX + (size_t)b0 * (N + 1) * NX, 8 * (size_t)nb * (N + 1) * NX, hipMemcpyDeviceToDevice, s)); return 0; }
    if (f == "U") { HIPCHK(hipMemcpyAsync(dst, c->dU + (size_t)b0 * N * NU, 8 * (size_t)nb * N * NU, hipMemcpyDeviceToDevice, s)); return 0; }
    if (f == "status") { HIPCHK(hipMemcpyAsync(dst, c->dstatus + b0, 4 * (size_t)nb, hipMemcpyDeviceToDevice, s)); return 0; }
    if (f == "qp_iter") { HIPCHK(hipMemcpyAsync(dst, c->dqpiter + b0, 4 * (size_t)nb, hipMemcpyDeviceToDevice, s)); return 0; }
    return fail("get_device: unknown field '" + f + "'");
}

// Results on the host behind an event instead of a stream synchronisation (include/tum_nmpc.h). The slabs are allocated on first
// use: pinned memory is a scarce host resource and most capsules (closed loops on the device, the RCCL gather) never ask.
// from_step: the request closes a tum_ocp_step_async whose clock pair hts[w] the packing kernel completes; a plain request never
// touches a clock pair (it would change the time_tot of the last step after the fact)
static int results_enqueue(tum_ocp *c, int with_iterate, bool from_step)
{
    const size_t B = c->batch; const int N = c->N;
    // Two sets of slabs / events used in turn: a caller keeps one request outstanding per capsule while it enqueues the next batch
    // and its request, and only then reads the older one -- the stream never runs dry between two batches waiting for the host.
    if (c->res_count == 2) return fail("results_async: two requests outstanding on this capsule (call tum_ocp_results_wait first)");
    const int w = (c->res_head + c->res_count) & 1;
    if (!c->evres[w]) HIPCHK(hipEventCreateWithFlags(&c->evres[w], hipEventDisableTiming));
    if (!c->dsum) { if (dalloc(&c->dsum, B * 5) != hipSuccess) return fail("results_async: device allocation failed"); }
    if (!c->hsum[w]) HIPCHK(hipHostMalloc((void **)&c->hsum[w], sizeof(double) * B * 5, hipHostMallocDefault));
    if (with_iterate && !c->hX[w]) {
        HIPCHK(hipHostMalloc((void **)&c->hX[w], sizeof(double) * B * (N + 1) * NX, hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&c->hU[w], sizeof(double) * B * N * NU, hipHostMallocDefault));
    }
    // The copies stay on the capsule's own stream. (Tried: a copy stream per capsule behind an event of the solve. Three capsules
    // then own six streams, more than the four hardware queues the runtime multiplexes streams onto: the capsules' compute
    // streams start to share queues and their batches no longer overlap -- 2.6 instead of 3.8 M solves/s on config 2.)
    hipStream_t s = c->stream;
    // the summary (40 B per instance) is written by the packing kernel STRAIGHT into the pinned slab (host memory mapped into the
    // device's address space): no copy command, no DMA engine and no signal round trip between the kernels of two batches
    const bool zero_copy = results_zero_copy();
    // small batches with the iterate: ONE kernel writes summary, X and U into the pinned slabs (a copy command costs 4-5 us on the
    // stream whatever its size; the two of the iterate were a tenth of a host-driven control step of one instance)
    const bool pack_all = results_pack_all(c, with_iterate);
    if (pack_all) {
        hipLaunchKernelGGL(pack_results_kernel, dim3(8), dim3(256), 0, s, c->dX, c->dU, c->dcost, c->dstatus, c->dqpiter, N, (int)B,
                           c->hsum[w], c->hX[w], c->hU[w], (from_step && c->ts_slot == w) ? c->hts[w] : (unsigned long long *)nullptr);
        HIPCHK(hipGetLastError());
    } else {
    hipLaunchKernelGGL(pack_summary_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, c->dU, c->dcost, c->dstatus, c->dqpiter, N, 0, (int)B,
                       zero_copy ? c->hsum[w] : c->dsum);
    HIPCHK(hipGetLastError());
    }
    if (!zero_copy) HIPCHK(hipMemcpyAsync(c->hsum[w], c->dsum, sizeof(double) * B * 5, hipMemcpyDeviceToHost, s));
    if (with_iterate && !pack_all) {
        HIPCHK(hipMemcpyAsync(c->hX[w], c->dX, sizeof(double) * B * (N + 1) * NX, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(c->hU[w], c->dU, sizeof(double) * B * N * NU, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipEventRecord(c->evres[w], s));
    c->res_iter[w] = with_iterate != 0;
    c->res_count++;
    return 0;
}

extern "C" int tum_ocp_results_async(tum_ocp *c, int with_iterate)
{
    if (!c) return fail("null capsule");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    return results_enqueue(c, with_iterate, false);
}

extern "C" int tum_ocp_results_outstanding(const tum_ocp *c) { return c ? c->res_count : -1; }

extern "C" int tum_ocp_results_wait(tum_ocp *c, const double **summary, const double **X, const double **U)
{
    if (!c) return fail("null capsule");
    if (c->res_count == 0) return fail("results_wait: no tum_ocp_results_async before it");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    const int r = c->res_head;          // the OLDEST outstanding request
    HIPCHK(hipEventSynchronize(c->evres[r]));
    if (summary) *summary = c->hsum[r];
    if (X) *X = c->res_iter[r] ? c->hX[r] : nullptr;
    if (U) *U = c->res_iter[r] ? c->hU[r] : nullptr;
    c->res_head ^= 1; c->res_count--;
    return 0;
}

extern "C" int tum_ocp_step_async(tum_ocp *c, const double *x0, const double *yref, int with_iterate)
{
    if (!c) return fail("null capsule");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (c->res_count == 2) return fail("step_async: two requests outstanding on this capsule (call tum_ocp_results_wait first)");
    if (flush_inputs(c)) return 1;          // (setters older than this step's inputs)
    const size_t B = c->batch, nx0 = B * NX, nyr = B * (size_t)(c->N + 1) * 6;
    // the staging area of the result slot this step will use: its previous step has been waited for, so its uploads are done
    const int w = (c->res_head + c->res_count) & 1;
    if ((x0 || yref) && !c->hin[w]) HIPCHK(hipHostMalloc((void **)&c->hin[w], sizeof(double) * (nx0 + nyr), hipHostMallocDefault));
    if (x0) {
        if (c->sn) { if (!c->have_offs) return fail("step_async x0: an SNMPC capsule needs its sample offsets (tum_ocp_snmpc_set_offsets)"); c->fanout = true; }
        memcpy(c->hin[w], x0, sizeof(double) * nx0);
    }
    if (yref) memcpy(c->hin[w] + nx0, yref, sizeof(double) * nyr);
    // small batches with inputs and the iterate: the step is timed by the device's wall clock, read by its first and its last kernel
    // (stage_in_kernel, pack_results_kernel), instead of by events around the solve -- every event on the stream is a gap of 5 us
    const bool small_in = (x0 || yref) && nx0 + nyr <= 32768;
    // (the packing kernel is what reads the clock at the end: the same predicate as its launch, results_pack_all)
    const bool clocked = small_in && results_pack_all(c, with_iterate);
    if (clocked) {
        if (!c->hts[w]) HIPCHK(hipHostMalloc((void **)&c->hts[w], 2 * sizeof(unsigned long long), hipHostMallocDefault));
        if (c->ts_khz == 0.0) { int khz = 0; HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->d.device)); c->ts_khz = khz > 0 ? (double)khz : 1e5; }
    }
    if (small_in) {      // one kernel reads the staging area (host memory mapped into the device's address space)
        hipLaunchKernelGGL(stage_in_kernel, dim3(4), dim3(256), 0, c->stream, c->hin[w], c->dx0, (int)nx0, c->dyref, (int)nyr, x0 ? 1 : 0, yref ? 1 : 0,
                           clocked ? c->hts[w] : (unsigned long long *)nullptr);
        HIPCHK(hipGetLastError());
    } else {
        if (x0) HIPCHK(hipMemcpyAsync(c->dx0, c->hin[w], sizeof(double) * nx0, hipMemcpyHostToDevice, c->stream));
        if (yref) HIPCHK(hipMemcpyAsync(c->dyref, c->hin[w] + nx0, sizeof(double) * nyr, hipMemcpyHostToDevice, c->stream));
    }
    // (the two events around the interior point kernel -- get_stats "time_ipm" -- are left out of a step: every event on the stream
    //  is a gap of a few microseconds between two kernels; "time_tot" keeps its events)
    c->skip_ipm_events = true;
    const int rc = launch(c, !clocked);
    c->skip_ipm_events = false;
    if (rc) return 1;
    c->ipm_timed = c->time_ipm;
    c->ts_slot = clocked ? w : -1;
    const int rr = results_enqueue(c, with_iterate, true);
    if (rr) c->ts_slot = -1;
    return rr;
}

// device-to-device upload of per-instance inputs from caller-owned HBM (asynchronous, on the capsule's stream)
extern "C" int tum_ocp_put_device(tum_ocp *c, const char *field, const void *src, int b0, int nb)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!field || !src) return fail("null argument");
    const int N = c->N;
    const std::string f(field);
    DevGuard guard(c->d.device); GUARD_OK(guard);
    hipStream_t s = c->stream;
    c->cache_valid = false;
    if (flush_inputs(c)) return 1;          // (setters older than this upload)
    if (f == "x0") {
        if (c->sn) { if (!c->have_offs) return fail("put_device x0: an SNMPC capsule needs its sample offsets (tum_ocp_snmpc_set_offsets)"); c->fanout = true; }
        HIPCHK(hipMemcpyAsync(c->dx0 + (size_t)b0 * NX, src, 8 * (size_t)nb * NX, hipMemcpyDeviceToDevice, s)); return 0;
    }
    if (f == "yref") { HIPCHK(hipMemcpyAsync(c->dyref + (size_t)b0 * (N + 1) * 6, src, 8 * (size_t)nb * (N + 1) * 6, hipMemcpyDeviceToDevice, s)); return 0; }
    if (f == "X") { HIPCHK(hipMemcpyAsync(c->dX + (size_t)b0 * (N + 1) * NX, src, 8 * (size_t)nb * (N + 1) * NX, hipMemcpyDeviceToDevice, s)); return 0; }
    if (f == "U") { HIPCHK(hipMemcpyAsync(c->dU + (size_t)b0 * N * NU, src, 8 * (size_t)nb * N * NU, hipMemcpyDeviceToDevice, s)); return 0; }
    return fail("put_device: unknown field '" + f + "'");
}

// Zero-copy inputs for callers whose batches are resident in HBM already (scenario fan-outs, sweeps: bench.py rotates resident batches):
// the capsule USES the caller's array as its x0 / yref array -- kernels read it in place, setters and the device closed loop write
// through to it -- instead of copying it into its own (tum_ocp_put_device: a blit kernel per field and step, 8 MB for yref at 16384 x
// N = 40; measured 1.2 % of a step on one stream, nothing with three capsules in flight, profiles/r05_bind_inputs.txt). Whole batch
// only; the memory must stay valid and unchanged while solves that use it are in flight. dev_ptr = NULL hands the capsule's own array back (its contents are what they were before the binding).
extern "C" int tum_ocp_bind_device(tum_ocp *c, const char *field, void *dev_ptr)
{
    if (!c || !field) return fail("null argument");
    const std::string f(field);
    if (f != "x0" && f != "yref") return fail("bind_device: field must be 'x0' or 'yref'");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (flush_inputs(c)) return 1;          // (setters parked in the pinned shadow belong to the array in use until now)
    if (f == "x0") {
        if (c->sn && dev_ptr) { if (!c->have_offs) return fail("bind_device x0: an SNMPC capsule needs its sample offsets (tum_ocp_snmpc_set_offsets)"); c->fanout = true; }
        c->dx0 = dev_ptr ? (double *)dev_ptr : c->dx0_own; c->ka.x0 = c->dx0;
    } else { c->dyref = dev_ptr ? (double *)dev_ptr : c->dyref_own; c->ka.yref = c->dyref; }
    c->epoch++;          // (a captured closed-loop chunk holds the pointers by value)
    return 0;
}

// One solve, then the condensed QP of instance b as the solve built it. Layout of `out` (doubles, N <= 40):
//   [0, 6400) H (80 x 80, symmetric, incl. the input cost and the unit padding) | [6400, 6480) q |
//   [6480, 6480 + 2 N 80) the steering-angle row and the gg row of every stage 1..N | [12880, 12880 + 2 N) their constants.
// The shipped library reads it back from the hand-over buffers of the pipeline (H tiles, gg rows in MFMA operand layout,
// q | d) -- so the dump shows what the interior point kernel is given; the development build keeps the fused kernel's own
// dump, which continues with g, the KKT matrix and the first right-hand side / step of the first iteration.
extern "C" int tum_ocp_debug_dump(tum_ocp *c, int b, double *out, int len)
{
    if (!c) return fail("null capsule");
    if (b < 0 || b >= DBG_INST || b >= c->batch) return fail("debug_dump: instance out of range");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (len > DBG_STRIDE) len = DBG_STRIDE;
#ifdef TUM_DEV_KERNELS
    if (c->kmode != 2 && c->kmode != 3) {
        c->ka.flags |= 2;
        const int rc = launch(c);
        c->ka.flags &= ~2;
        if (rc) return 1;
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(out, c->ddbg + (size_t)b * DBG_STRIDE, sizeof(double) * len, hipMemcpyDeviceToHost));
        return 0;
    }
#endif
    if (c->N > NMAX) return fail("debug_dump: the condensed-QP dump is built for N <= 40");
    if (launch(c)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    using D = PD<5>;
    const int N = c->N;
    std::vector<double> hw((size_t)D::NTT * 256), cw((size_t)D::NCH * 64), vv(D::PVEC), o(DBG_STRIDE, 0.0);
    HIPCHK(hipMemcpy(hw.data(), c->dhws + (size_t)b * D::NTT * 256, sizeof(double) * hw.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cw.data(), c->dcws + (size_t)b * D::NCH * 64, sizeof(double) * cw.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(vv.data(), c->dvec + (size_t)b * D::PVEC, sizeof(double) * vv.size(), hipMemcpyDeviceToHost));
    for (int K = 0; K < D::NT; K++)
        for (int I = K; I < D::NT; I++)
            for (int l = 0; l < 64; l++)
                for (int jj = 0; jj < 4; jj++) {         // accumulator layout: row (l >> 4) + 4 jj, column l & 15
                    const int row = 16 * K + (l >> 4) + 4 * jj, col = 16 * I + (l & 15);
                    const double v = hw[((size_t)D::tidx(K, I) * 64 + l) * 4 + jj];
                    o[row * 80 + col] = v; o[col * 80 + row] = v;
                }
    for (int i = 0; i < 80; i++) o[6400 + i] = vv[D::PV_Q + i];
    for (int s = 1; s <= N; s++)
        for (int col = 0; col < 80; col++) {
            o[6480 + (2 * (s - 1)) * 80 + col] = ((col & 1) && col < 2 * s) ? c->ka.dt : 0.0;
            const int cc = (s - 1) >> 2, lq = (s - 1) & 3, T = col >> 4;       // operand layout: chunk cc, DPP row lq, tile column T
            o[6480 + (2 * (s - 1) + 1) * 80 + col] = (cc >= 2 * T) ? cw[(size_t)D::cidx(cc, T) * 64 + 16 * lq + (col & 15)] : 0.0;
        }
    for (int i = 0; i < 2 * N; i++) o[12880 + i] = vv[D::PV_D + i];
    // (development aid: the tail of the dump area carries the phase cycle counters of the SNMPC prologue kernels, SnArgs::dbg)
    HIPCHK(hipMemcpy(o.data() + 20000, c->ddbg + (size_t)b * DBG_STRIDE + 20000, sizeof(double) * (DBG_STRIDE - 20000), hipMemcpyDeviceToHost));
    memcpy(out, o.data(), sizeof(double) * len);
    return 0;
}

// development aid: one solve with the in-kernel phase timers on; out = batch x 12 cycle counters
extern "C" int tum_ocp_profile_phases(tum_ocp *c, long long *out)
{
    if (!c || !out) return fail("null argument");
    if (c->N > NMAX) return fail("profile_phases: the instrumented kernels are built for N <= 40");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    c->ka.flags |= 4;
    int rc = launch(c);
    c->ka.flags &= ~4;
    if (rc) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->dprof, sizeof(long long) * 12 * (size_t)c->batch, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------------- K6 / K7
extern "C" int tum_ocp_set_x0_fanout(tum_ocp *c, const double *pose, const double *offs, int P, int S)
{
    if (!c || !pose || (S > 0 && !offs)) return fail("null argument");
    const int S1 = S + 1;
    if (P < 1 || S < 0 || (long long)P * S1 != c->batch) return fail("set_x0_fanout: P*(S+1) must equal the batch size");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (flush_inputs(c)) return 1;
    DevTmp tp, to;
    HIPCHK(tp.alloc(sizeof(double) * P * NX));
    HIPCHK(to.alloc(sizeof(double) * (S > 0 ? S : 1) * NX));
    double *dp = tp.as<double>(), *dofs = to.as<double>();
    HIPCHK(hipMemcpyAsync(dp, pose, sizeof(double) * P * NX, hipMemcpyHostToDevice, c->stream));
    if (S > 0) HIPCHK(hipMemcpyAsync(dofs, offs, sizeof(double) * S * NX, hipMemcpyHostToDevice, c->stream));
    const int n = P * S1 * NX;
    hipLaunchKernelGGL(sigma_fanout_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->dx0, dp, dofs, P, S1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// PCE matrix of the scenario fan-out kept on the device (L x S, row-major, host) for tum_pce_moments_device
extern "C" int tum_pce_attach(tum_ocp *c, const double *A, int L, int S)
{
    if (!c || !A) return fail("null argument");
    if (S < 1 || L < 1 || c->batch % (S + 1) != 0) return fail("pce_attach: batch must be a multiple of S+1");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(c->dpceA); c->dpceA = nullptr;
    if (dalloc(&c->dpceA, (size_t)L * S) != hipSuccess) return fail("pce_attach: device allocation failed");
    HIPCHK(hipMemcpy(c->dpceA, A, sizeof(double) * L * S, hipMemcpyHostToDevice));
    c->pce_L = L; c->pce_S = S;
    return 0;
}

static int pce_launch(tum_ocp *c, const std::string &f, int stage, const double *dA, int L, int S, double *dmean, double *dvar)
{
    const int S1 = S + 1, N = c->N, P = c->batch / S1;
    int m; const double *src; size_t rec, off;
    if (f == "x") { if (stage < 0 || stage > N) return fail("pce_moments: stage"); m = NX; src = c->dX; rec = (size_t)(N + 1) * NX; off = (size_t)stage * NX; }
    else if (f == "u") { if (stage < 0 || stage >= N) return fail("pce_moments: stage"); m = NU; src = c->dU; rec = (size_t)N * NU; off = (size_t)stage * NU; }
    else return fail("pce_moments: unknown field '" + f + "'");
    hipLaunchKernelGGL(pce_moments_kernel, dim3((P * m + 127) / 128), dim3(128), 0, c->stream, src + off, rec, dA, P, S1, m, L, dmean, dvar);
    HIPCHK(hipGetLastError());
    return 0;
}

// Asynchronous, device-to-device flavour (the reduction of BASELINE config 3 inside the timed step): mean_dev / var_dev are
// caller-owned device buffers of P x m doubles; the matrix is the one registered with tum_pce_attach.
extern "C" int tum_pce_moments_device(tum_ocp *c, const char *field, int stage, double *mean_dev, double *var_dev)
{
    if (!c || !field || !mean_dev || !var_dev) return fail("null argument");
    if (!c->dpceA) return fail("pce_moments_device: no PCE matrix registered (tum_pce_attach)");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    return pce_launch(c, field, stage, c->dpceA, c->pce_L, c->pce_S, mean_dev, var_dev);
}

extern "C" int tum_pce_moments(tum_ocp *c, const char *field, int stage, const double *A, int L, int S, double *mean, double *var)
{
    if (!c || !field || !A || !mean || !var) return fail("null argument");
    const int S1 = S + 1;
    if (S < 1 || L < 1 || c->batch % S1 != 0) return fail("pce_moments: batch must be a multiple of S+1");
    const int P = c->batch / S1;
    const int m = (std::string(field) == "u") ? NU : NX;
    DevGuard guard(c->d.device); GUARD_OK(guard);
    DevTmp tA, tm, tv;
    HIPCHK(tA.alloc(sizeof(double) * L * S));
    HIPCHK(tm.alloc(sizeof(double) * P * m)); HIPCHK(tv.alloc(sizeof(double) * P * m));
    double *dA = tA.as<double>(), *dm = tm.as<double>(), *dv = tv.as<double>();
    HIPCHK(hipMemcpyAsync(dA, A, sizeof(double) * L * S, hipMemcpyHostToDevice, c->stream));
    if (pce_launch(c, field, stage, dA, L, S, dm, dv)) return 1;
    HIPCHK(hipMemcpyAsync(mean, dm, sizeof(double) * P * m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(var, dv, sizeof(double) * P * m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Snapshot / restore of all per-stage bounds (lbu, ubu, lbx, ubx, lh, uh) on the device. The R2NMPC tightening rewrites the
// bounds after every solve; a benchmark or a sweep that restarts from the nominal problem restores them without a host copy.
extern "C" int tum_ocp_bounds_snapshot(tum_ocp *c)
{
    if (!c) return fail("null capsule");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    const size_t n = (size_t)c->batch * 6 * (c->N + 1);
    if (!c->dbnd_snap && dalloc(&c->dbnd_snap, n) != hipSuccess) return fail("bounds_snapshot: device allocation failed");
    HIPCHK(hipMemcpyAsync(c->dbnd_snap, c->dbnd, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int tum_ocp_bounds_restore(tum_ocp *c)
{
    if (!c) return fail("null capsule");
    if (!c->dbnd_snap) return fail("bounds_restore: no snapshot taken");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    HIPCHK(hipMemcpyAsync(c->dbnd, c->dbnd_snap, sizeof(double) * (size_t)c->batch * 6 * (c->N + 1), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

extern "C" int tum_ocp_r2_backoff(tum_ocp *c, const double *Sigma0, const double *BWB, int uph,
                                  double delta_min, double delta_max, double uh_nom, double *backoff)
{
    if (!c || !Sigma0 || !BWB) return fail("null argument");
    if (!c->d.store_qp_in) return fail("r2_backoff: capsule created without store_qp_in");
    if (!c->solved) return fail("r2_backoff: no solve yet");
    if (uph < 1) return fail("r2_backoff: uncertainty propagation horizon < 1");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    const int N = c->N;
    DevTmp tS, tB, tbo;
    HIPCHK(tS.alloc(64 * 8)); HIPCHK(tB.alloc(64 * 8));
    double *dS = tS.as<double>(), *dB = tB.as<double>(), *dbo = nullptr;
    if (backoff) {
        HIPCHK(tbo.alloc(sizeof(double) * (size_t)c->batch * N * 2)); dbo = tbo.as<double>();
        HIPCHK(hipMemsetAsync(dbo, 0, sizeof(double) * (size_t)c->batch * N * 2, c->stream));
    }
    HIPCHK(hipMemcpyAsync(dS, Sigma0, 64 * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dB, BWB, 64 * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(r2_backoff_kernel, dim3((c->batch + 3) / 4), dim3(256), 0, c->stream, c->dqpin, c->dX, c->dbnd, c->ka.mp,
                       dS, dB, N, uph, c->batch, delta_min, delta_max, uh_nom, dbo, c->dstatus,
                       c->solved_pipe ? c->drec : (const double *)nullptr, PREC);
    HIPCHK(hipGetLastError());
    if (backoff) HIPCHK(hipMemcpyAsync(backoff, dbo, sizeof(double) * (size_t)c->batch * N * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Make the R2NMPC tightening part of every solve (what Reduced_Robustified_NMPC_class.py:276-378 does after each successful
// acados call): after the SQP-RTI kernel the back-off kernel rewrites lbx / ubx / uh of the stages 1..N-1 for the NEXT solve,
// on the capsule's stream, inside the time `get_stats("time_tot")` reports (:379-381). uph = 0 detaches.
extern "C" int tum_ocp_r2_attach(tum_ocp *c, const double *Sigma0, const double *BWB, int uph,
                                 double delta_min, double delta_max, double uh_nom)
{
    if (!c) return fail("null capsule");
    if (uph == 0) { if (c->r2) c->epoch++; c->r2 = false; return 0; }
    if (!Sigma0 || !BWB) return fail("null argument");
    if (!c->d.store_qp_in) return fail("r2_attach: capsule created without store_qp_in");
    if (c->sn) return fail("r2_attach: not available for an SNMPC capsule");
    if (uph < 1) return fail("r2_attach: uncertainty propagation horizon < 1");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (!c->dr2S && (dalloc(&c->dr2S, 64) != hipSuccess || dalloc(&c->dr2B, 64) != hipSuccess)) return fail("r2_attach: device allocation failed");
    HIPCHK(hipMemcpy(c->dr2S, Sigma0, 64 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->dr2B, BWB, 64 * 8, hipMemcpyHostToDevice));
    c->r2_uph = uph; c->r2_dmin = delta_min; c->r2_dmax = delta_max; c->r2_uh = uh_nom;
    c->r2 = true;
    c->epoch++;
    return 0;
}

extern "C" int tum_ocp_constraints_get(tum_ocp *c, int stage, const char *field, double *v, int b0, int nb)
{
    if (chk_range(c, b0, nb)) return 1;
    if (!field || !v) return fail("null argument");
    const int N = c->N, NB = N + 1;
    const std::string f(field);
    int row;
    if (f == "lbu") row = 0; else if (f == "ubu") row = 1; else if (f == "lbx") row = 2; else if (f == "ubx") row = 3;
    else if (f == "lh") row = 4; else if (f == "uh") row = 5; else return fail("constraints_get: unknown field '" + f + "'");
    if (stage < 0 || stage > N) return fail("constraints_get: stage out of range");
    return fetch(c, c->dbnd, (size_t)6 * NB, (size_t)row * NB + stage, v, 1, b0, nb, 1);
}

// ---------------------------------------------------------------------------------------------- K8 / K9: device closed loop
struct tum_sim {
    tum_ocp *c;
    int n_track, loop_circuit, n_elem, step, log_cap;
    double Tp, Ts;
    int win[8];
    double *dtrack, *dxsim, *dpose, *dhist, *dref0;
    int *dclosest, *derr, *dstep;
    hipGraphExec_t graph; int graph_steps;            // captured chunk of control steps (tum_sim_run)
    unsigned graph_epoch; bool graph_fanout;          // configuration of the capsule the chunk was captured with
    double *lCiLX, *lSimX, *lU, *lREF, *lDBG;
    hipStream_t s2; hipEvent_t evF, evJ;              // side stream of the linearisation that runs beside the planner, fork / join events
    double *ddw, *dde; int dist_len;                  // disturbance realisation played back by the loop (tum_sim_set_disturbances)
};

extern "C" int tum_planner_emulate(const double *track, int n_track, const double *pose, int P, int n_points, double Tp,
                                   int loop_circuit, double *ref_out, int *closest_out, int device)
{
    if (!track || !pose || !ref_out) return fail("null argument");
    if (n_track < 2 || P < 1 || n_points < 2) return fail("planner_emulate: bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device: libtumnmpc has no CPU fallback");
    if (device < 0 || device >= ndev) return fail("planner_emulate: device ordinal out of range");
    DevGuard guard(device); GUARD_OK(guard);
    DevTmp tt, tp, tout, tcl, terr;
    HIPCHK(tt.alloc(sizeof(double) * 4 * n_track)); HIPCHK(tp.alloc(sizeof(double) * 2 * P));
    HIPCHK(tout.alloc(sizeof(double) * 4 * (size_t)P * n_points)); HIPCHK(tcl.alloc(sizeof(int) * P)); HIPCHK(terr.alloc(sizeof(int)));
    double *dt = tt.as<double>(), *dp = tp.as<double>(), *dout = tout.as<double>(); int *dcl = tcl.as<int>(), *derr = terr.as<int>();
    HIPCHK(hipMemset(derr, 0, sizeof(int)));
    HIPCHK(hipMemcpy(dt, track, sizeof(double) * 4 * n_track, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dp, pose, sizeof(double) * 2 * P, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(planner_kernel, dim3(P), dim3(64), 0, 0, dt, n_track, dp, 2, n_points, Tp, loop_circuit, dout, 4,
                       (double *)nullptr, dcl, derr, P, (int *)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    int err = 0;
    HIPCHK(hipMemcpy(&err, derr, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ref_out, dout, sizeof(double) * 4 * (size_t)P * n_points, hipMemcpyDeviceToHost));
    if (closest_out) HIPCHK(hipMemcpy(closest_out, dcl, sizeof(int) * P, hipMemcpyDeviceToHost));
    if (err) return fail("planner_emulate: extracted segment longer than PLAN_MAXM points");
    return 0;
}

extern "C" void tum_sim_free(tum_sim *s)
{
    if (!s) return;
    DevGuard guard(s->c->d.device);
    (void)hipFree(s->dtrack); (void)hipFree(s->dxsim); (void)hipFree(s->dpose); (void)hipFree(s->dhist); (void)hipFree(s->dref0);
    (void)hipFree(s->dclosest); (void)hipFree(s->derr); (void)hipFree(s->dstep);
    if (s->graph) (void)hipGraphExecDestroy(s->graph);
    if (s->s2) { (void)hipStreamSynchronize(s->s2); (void)hipStreamDestroy(s->s2); (void)hipEventDestroy(s->evF); (void)hipEventDestroy(s->evJ); }
    (void)hipFree(s->lCiLX); (void)hipFree(s->lSimX); (void)hipFree(s->lU); (void)hipFree(s->lREF); (void)hipFree(s->lDBG);
    (void)hipFree(s->ddw); (void)hipFree(s->dde);
    delete s;
}

extern "C" tum_sim *tum_sim_create(tum_ocp *c, const double *track, int n_track, double Tp, int loop_circuit, double Ts, int n_elem,
                                   const int *windows, int log_capacity)
{
    if (!c || !track || !windows) { fail("null argument"); return nullptr; }
    if (n_track < 2 || !(Tp > 0) || !(Ts > 0) || n_elem < 1 || log_capacity < 0) { fail("sim_create: bad arguments"); return nullptr; }
    for (int i = 0; i < 8; i++) if (windows[i] < 1 || windows[i] > 4) { fail("sim_create: estimator windows must be 1..4"); return nullptr; }
    DevGuard guard(c->d.device); if (!guard.ok) { fail("hipSetDevice failed"); return nullptr; }
    if (c->sn) {   // the state estimator writes the nominal x0 only: the samples follow by fan-out
        if (!c->have_offs) { fail("sim_create: an SNMPC capsule needs its sample offsets (tum_ocp_snmpc_set_offsets)"); return nullptr; }
        c->fanout = true;
    }
    tum_sim *s = new tum_sim();
    memset(s, 0, sizeof(*s));
    s->c = c; s->n_track = n_track; s->loop_circuit = loop_circuit; s->n_elem = n_elem; s->Tp = Tp; s->Ts = Ts; s->log_cap = log_capacity;
    for (int i = 0; i < 8; i++) s->win[i] = windows[i];
    const size_t B = c->batch, L = log_capacity;
    bool ok = true;
    ok &= dalloc(&s->dtrack, (size_t)4 * n_track) == hipSuccess;
    ok &= dalloc(&s->dxsim, B * 7) == hipSuccess && dalloc(&s->dpose, B * 2) == hipSuccess && dalloc(&s->dhist, B * 32) == hipSuccess;
    ok &= dalloc(&s->dref0, B * 4) == hipSuccess && dalloc(&s->dclosest, B) == hipSuccess && dalloc(&s->derr, (size_t)1) == hipSuccess && dalloc(&s->dstep, (size_t)2) == hipSuccess;
    if (L > 0) {
        ok &= dalloc(&s->lCiLX, (L + 1) * B * 7) == hipSuccess && dalloc(&s->lSimX, (L + 1) * B * 8) == hipSuccess;
        ok &= dalloc(&s->lU, L * B * 2) == hipSuccess && dalloc(&s->lREF, L * B * 4) == hipSuccess && dalloc(&s->lDBG, L * B * 5) == hipSuccess;
    }
    ok = ok && hipMemcpy(s->dtrack, track, sizeof(double) * 4 * n_track, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { fail("sim_create: device allocation failed"); tum_sim_free(s); return nullptr; }
    return s;
}

// initial plant state (batch x 7) and controller state (batch x 8): X0_sim / X0_MPC of SimulationMode_main_class.py:60-75
extern "C" int tum_sim_set_state(tum_sim *s, const double *x_sim, const double *x_mpc, int cold_start)
{
    if (!s || !x_sim || !x_mpc) return fail("null argument");
    tum_ocp *c = s->c; const size_t B = c->batch;
    DevGuard guard(c->d.device); GUARD_OK(guard);
    c->cache_valid = false;
    if (flush_inputs(c)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(s->dxsim, x_sim, sizeof(double) * B * 7, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->dx0, x_mpc, sizeof(double) * B * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy2D(s->dpose, 2 * 8, x_mpc, 8 * 8, 2 * 8, B, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(s->dhist, 0, sizeof(double) * B * 32));
    HIPCHK(hipMemset(s->dstep, 0, 2 * sizeof(int)));
    HIPCHK(hipDeviceSynchronize());          // (the fills run on the NULL stream, the loop on the capsule's non-blocking one)
    s->step = 0;
    if (s->log_cap > 0) {
        HIPCHK(hipMemcpy(s->lCiLX, x_sim, sizeof(double) * B * 7, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->lSimX, x_mpc, sizeof(double) * B * 8, hipMemcpyHostToDevice));
    }
    if (cold_start) return tum_ocp_cold_start(c);
    return 0;
}

extern "C" int tum_sim_plan(tum_sim *s)
{
    if (!s) return fail("null argument");
    tum_ocp *c = s->c;
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (flush_inputs(c)) return 1;
    hipLaunchKernelGGL(planner_kernel, dim3(c->batch), dim3(64), 0, c->stream, s->dtrack, s->n_track, s->dpose, 2, c->N + 1, s->Tp,
                       s->loop_circuit, c->dyref, 6, s->dref0, s->dclosest, s->derr, c->batch, (int *)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int tum_sim_advance(tum_sim *s)
{
    if (!s) return fail("null argument");
    tum_ocp *c = s->c;
    if (!c->solved) return fail("sim_advance: no solve yet");
    DevGuard guard(c->d.device); GUARD_OK(guard);
    c->cache_valid = false;          // (the plant re-initialises the iterate of an instance whose solve failed, and writes the next x0)
    if (flush_inputs(c)) return 1;
    SimArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.N = c->N; sa.batch = c->batch; sa.n_elem = s->n_elem; sa.step_counter = s->dstep; sa.log_cap = s->log_cap; sa.Ts = s->Ts;
    for (int i = 0; i < 8; i++) sa.win[i] = s->win[i];
    const tum_ocp_desc &d = c->d;
    PlantModel &p = sa.pm;
    p.lf = d.lf; p.lr = d.lr; p.m = d.m; p.inv_m = 1.0 / d.m; p.inv_Iz = 1.0 / d.Iz; p.ka = 0.5 * d.ro * d.S * d.Cd;
    p.Bf = d.Bf; p.Cf = d.Cf; p.Df = d.Df; p.Ef = d.Ef; p.Br = d.Br; p.Cr = d.Cr; p.Dr = d.Dr; p.Er = d.Er;
    p.Fz_f = d.m * d.lr * d.g / (d.lf + d.lr); p.Fz_r = d.m * d.lf * d.g / (d.lf + d.lr);
    p.invFmax_f = 1.0 / std::sqrt(p.Fz_f * p.Fz_f + (d.Cf * p.Fz_f) * (d.Cf * p.Fz_f));
    p.invFmax_r = 1.0 / std::sqrt(p.Fz_r * p.Fz_r + (d.Cr * p.Fz_r) * (d.Cr * p.Fz_r));
    p.fr0 = d.fr0; p.fr1 = d.fr1; p.fr4 = d.fr4;
    sa.X = c->dX; sa.U = c->dU; sa.cost = c->dcost; sa.status = c->dstatus; sa.qp_iter = c->dqpiter;
    sa.ns = c->sn ? c->sa.ns : 0; sa.XS = c->dXS; sa.xs0 = c->dxs0;
    sa.bnd = c->dbnd; sa.r2 = c->r2 ? 1 : 0; sa.r2_dmin = c->r2_dmin; sa.r2_dmax = c->r2_dmax; sa.r2_uh = c->r2_uh;
    sa.x_sim = s->dxsim; sa.x0 = c->dx0; sa.pose = s->dpose; sa.hist = s->dhist; sa.ref0 = s->dref0;
    sa.lCiLX = s->lCiLX; sa.lSimX = s->lSimX; sa.lU = s->lU; sa.lREF = s->lREF; sa.lDBG = s->lDBG;
    sa.dist_w = s->ddw; sa.dist_e = s->dde; sa.dist_len = s->dist_len;
    hipLaunchKernelGGL(plant_advance_kernel, dim3((c->batch * PLANT_LANES + 63) / 64), dim3(64), 0, c->stream, sa);
    HIPCHK(hipGetLastError());
    s->step++;
    return 0;
}

// nsteps x (planner, SQP-RTI solve, plant + estimator) on the capsule's stream; returns after the last one. The loop is
// launch-bound for small batches (3 short kernels per control step), so chunks of GRAPH_STEPS steps are captured once into
// a hipGraph and replayed; all per-step state (step counter, ring buffers, logs) lives in device memory, so the captured
// kernel arguments never change.
static const int GRAPH_STEPS = 25;
static int sim_enqueue_step_body(tum_sim *s, bool events);
static int sim_enqueue_step(tum_sim *s, bool events)
{
    // whatever goes wrong between the forked linearisation and the solve that consumes it (planner, join, a failed capture): the
    // capsule must not keep the "linearisation already done" mark for a LATER solve, which would run on stale stage records
    const int rc = sim_enqueue_step_body(s, events);
    if (rc) s->c->lin_ahead = false;
    return rc;
}
static int sim_enqueue_step_body(tum_sim *s, bool events)
{
    tum_ocp *c = s->c;
    if (lin_ahead_ok(c)) {
        // fork: the linearisation of this step's solve on the side stream, behind everything the capsule's stream holds (the plant
        // of the previous step may have re-initialised the iterate); the planner on the capsule's stream; join in front of the
        // condensing kernel. Inside a stream capture this becomes two branches of the graph.
        HIPCHK(hipEventRecord(s->evF, c->stream));
        HIPCHK(hipStreamWaitEvent(s->s2, s->evF, 0));
        launch_lin_ahead(c, s->s2);
        HIPCHK(hipEventRecord(s->evJ, s->s2));
        if (tum_sim_plan(s)) return 1;
        HIPCHK(hipStreamWaitEvent(c->stream, s->evJ, 0));
    } else if (tum_sim_plan(s)) return 1;
    if (launch(c, events)) return 1;
    return tum_sim_advance(s);
}

extern "C" int tum_sim_run(tum_sim *s, int nsteps)
{
    if (!s || nsteps < 0) return fail("bad argument");
    tum_ocp *c = s->c;
    DevGuard guard(c->d.device); GUARD_OK(guard);
    c->cache_valid = false;
    if (flush_inputs(c)) return 1;            // (pending host setters go up before anything is captured)
    if (resolve_kernel(c)) return 1;          // (workspace allocation must not happen inside the capture below)
    // ... nor the reallocation / synchronisation a changed SNMPC parameter vector can trigger, nor the deferred freeze
    if (c->sn && (sn_apply_p(c) || sn_materialise(c))) return 1;
    if (lin_ahead_ok(c) && !s->s2) {          // (created here: not inside the capture below)
        HIPCHK(hipStreamCreateWithFlags(&s->s2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&s->evF, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&s->evJ, hipEventDisableTiming));
    }
    int done = 0;
    if (nsteps >= 2 * GRAPH_STEPS) {
        // a captured chunk holds the kernel variant, the schedule flag, the SNMPC horizon / risk parameter / work-buffer
        // pointers and the R2 attachment BY VALUE: after any of them changed the chunk is captured again
        if (s->graph && (s->graph_epoch != c->epoch || s->graph_fanout != c->fanout)) { (void)hipGraphExecDestroy(s->graph); s->graph = nullptr; }
        if (!s->graph) {
            s->graph_epoch = c->epoch; s->graph_fanout = c->fanout;
            hipGraph_t g = nullptr;
            const int step0 = s->step;
            bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                c->solved = true;                                      // the captured solve precedes every captured advance
                for (int i = 0; i < GRAPH_STEPS && ok; i++) ok = sim_enqueue_step(s, false) == 0;
                ok = (hipStreamEndCapture(c->stream, &g) == hipSuccess) && ok;
                s->step = step0;                                       // nothing ran yet
            }
            if (ok && g) ok = hipGraphInstantiate(&s->graph, g, nullptr, nullptr, 0) == hipSuccess;
            if (g) (void)hipGraphDestroy(g);
            if (!ok) { s->graph = nullptr; c->lin_ahead = false; (void)hipGetLastError(); }  // fall back to plain launches
            s->graph_steps = GRAPH_STEPS;
        }
        while (s->graph && nsteps - done >= s->graph_steps) {
            HIPCHK(hipGraphLaunch(s->graph, c->stream));
            done += s->graph_steps; s->step += s->graph_steps;
            // what launch() records on the host for every solve holds for the replayed solves as well
            if (c->sn) c->xs_lazy = true;
            c->solved = true; c->solved_pipe = c->pipe; if (c->lpt && c->batch > 1024) c->order_valid = true;
        }
    }
    for (; done < nsteps; done++)
        if (sim_enqueue_step(s, true)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    int err = 0;
    HIPCHK(hipMemcpy(&err, s->derr, sizeof(int), hipMemcpyDeviceToHost));
    if (err) return fail("sim_run: planner segment longer than PLAN_MAXM points");
    return 0;
}

extern "C" int tum_sim_steps(const tum_sim *s) { return s ? s->step : -1; }

// Disturbance realisation of the loop (Utils/SimulationMode_main_class.py:121-143 with disturbance_playback, and what
// Utils/MPC_sim_utils.py:54-67 generate_disturbances draws otherwise -- drawn on the host, closed_loop.DisturbanceModel): w_deriv
// (additive on the state derivatives, constant over a control step) and e_est (additive state estimation error), n_steps x batch x 7
// each (host, step-major), either may be null. Control step i >= n_steps runs undisturbed. n_steps = 0 removes the realisation.
extern "C" int tum_sim_set_disturbances(tum_sim *s, const double *w_deriv, const double *e_est, int n_steps)
{
    if (!s || n_steps < 0) return fail("bad argument");
    tum_ocp *c = s->c;
    DevGuard guard(c->d.device); GUARD_OK(guard);
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(s->ddw); (void)hipFree(s->dde); s->ddw = s->dde = nullptr; s->dist_len = 0;
    if (s->graph) { (void)hipGraphExecDestroy(s->graph); s->graph = nullptr; }          // (the pointers are kernel arguments of a captured chunk)
    if (n_steps == 0 || (!w_deriv && !e_est)) return 0;
    const size_t n = (size_t)n_steps * c->batch * 7;
    if (w_deriv) { HIPCHK(hipMalloc((void **)&s->ddw, sizeof(double) * n)); HIPCHK(hipMemcpy(s->ddw, w_deriv, sizeof(double) * n, hipMemcpyHostToDevice)); }
    if (e_est) { HIPCHK(hipMalloc((void **)&s->dde, sizeof(double) * n)); HIPCHK(hipMemcpy(s->dde, e_est, sizeof(double) * n, hipMemcpyHostToDevice)); }
    s->dist_len = n_steps;
    return 0;
}

extern "C" int tum_sim_get(tum_sim *s, const char *field, double *out, long long len)
{
    if (!s || !field || !out) return fail("null argument");
    tum_ocp *c = s->c; const long long B = c->batch;
    const long long L = s->step < s->log_cap ? s->step : s->log_cap;
    const std::string f(field);
    DevGuard guard(c->d.device); GUARD_OK(guard);
    if (flush_inputs(c)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    const double *src = nullptr; long long want = 0;
    if (f == "x_sim") { src = s->dxsim; want = B * 7; }
    else if (f == "x_mpc") { src = c->dx0; want = B * 8; }
    else if (f == "pose") { src = s->dpose; want = B * 2; }
    else if (f == "ref0") { src = s->dref0; want = B * 4; }
    else if (f == "graph_steps") {         // control steps per captured hipGraph chunk (0: plain launches)
        if (len != 1) return fail("sim_get graph_steps: len != 1");
        out[0] = s->graph ? s->graph_steps : 0;
        return 0;
    }
    else if (f == "closest") {
        if (len != B) return fail("sim_get closest: len != batch");
        std::vector<int> t(B);
        HIPCHK(hipMemcpy(t.data(), s->dclosest, sizeof(int) * B, hipMemcpyDeviceToHost));
        for (long long i = 0; i < B; i++) out[i] = t[i];
        return 0;
    }
    else if (f == "CiLX") { src = s->lCiLX; want = (L + 1) * B * 7; }
    else if (f == "MPC_SimX") { src = s->lSimX; want = (L + 1) * B * 8; }
    else if (f == "simU") { src = s->lU; want = L * B * 2; }
    else if (f == "simREF") { src = s->lREF; want = L * B * 4; }
    else if (f == "simSolverDebug") { src = s->lDBG; want = L * B * 5; }
    else return fail("sim_get: unknown field '" + f + "'");
    if (!src) return fail("sim_get: logging disabled (log_capacity 0)");
    if (len != want) return fail("sim_get " + f + ": len mismatch");
    if (want > 0) HIPCHK(hipMemcpy(out, src, sizeof(double) * want, hipMemcpyDeviceToHost));
    return 0;
}
