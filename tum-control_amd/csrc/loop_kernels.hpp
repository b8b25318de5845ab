// loop_kernels.hpp -- the producer and the consumer of the SQP-RTI solve, on the device:
//   K8 planner_kernel        local reference extraction for every instance (Utils/MPC_sim_utils.py:137-194 PlannerEmulator)
//   K9 plant_advance_kernel  plant step + state estimation (Utils/SimulationMode_main_class.py:106-156 sim_step / StateEstimation,
//                            Vehicle_Simulator/sim_model_dynamic_stm_pacejka.py:137-195, VehicleSimulator.py:73-77)
// With them a batch of closed loops runs planner -> solve -> plant without leaving HBM.
#pragma once
#include "nmpc_device.hpp"

namespace tum {

constexpr int PLAN_MAXM = 512;      // longest extracted segment (points) the yaw-unwrap buffer holds

// np.mod(x, 2*pi): result carries the sign of the divisor
__device__ __forceinline__ double pymod_2pi(double x)
{
    const double P = 2.0 * M_PI;
    double r = fmod(x, P);
    if (r != 0.0 && r < 0.0) r += P;
    return r;
}

// One wavefront per instance. track: n x 4 row-major [pos_x, pos_y, ref_yaw, ref_v]; pose: [b][pose_stride] (x, y first).
// out: [b][npts][out_stride] gets [x, y, yaw, v] in its first four slots (the yref block: out_stride 6, slots 4,5 zeroed;
// a bare reference: out_stride 4). Arithmetic follows numpy's order of operations (no contraction): nearest waypoint =
// first minimum of (px-x)^2+(py-y)^2; walk forward while the accumulated travel time <= Tp; np.linspace / np.interp
// resampling to npts samples; the yaw channel is interpolated on the unwrapped signal when the segment crosses the
// 2*pi seam (any step > 250 degrees) and wrapped back into [0, 2*pi).
__global__ void __launch_bounds__(64) planner_kernel(const double *track, int n, const double *pose, int pose_stride, int npts, double Tp,
                                                     int loop_circuit, double *out, int out_stride, double *ref0, int *closest,
                                                     int *err, int batch, int *step_counter)
{
#pragma clang fp contract(off)
    __shared__ double sYaw[PLAN_MAXM];
    const int lane = threadIdx.x, b = blockIdx.x;
    if (b >= batch) return;
    const double x = pose[(size_t)b * pose_stride], y = pose[(size_t)b * pose_stride + 1];
    // ---- closest waypoint (np.argmin: first minimum)
    double best = INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const double dx = track[4 * i] - x, dy = track[4 * i + 1] - y;
        const double d = dx * dx + dy * dy;
        if (d < best) { best = d; bi = i; }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double od = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
    }
    const int i0 = bi;
    // ---- walk forward: indices are consecutive modulo n. The travel times of the next 64 segments are computed by the 64 lanes at
    // once (loads, hypot, division in parallel); the running sum and its termination test stay SEQUENTIAL and wave-uniform, in the
    // reference's order of additions (T = ((0 + s0) + s1) + ...: the number of points m must not depend on a summation order).
    // (One lane walking the line paid a dependent global load per point: 15 of this kernel's 21 us in the small-batch loops.)
    int m = 1; double T = 0.0; bool over = false;
    {
        auto wrap = [&](int i) -> int { return (i >= n) ? i - n * (i / n) : i; };
        bool done = false;
        for (int base = 0; !done; base += 64) {
            const int ia = i0 + base + lane, ib = ia + 1;
            const bool ok = loop_circuit || ib < n;
            const int pa = wrap(ia), pb = ok ? wrap(ib) : pa;
            const double seg = hypot(track[4 * pb] - track[4 * pa], track[4 * pb + 1] - track[4 * pa + 1]) / track[4 * pb + 3];
            for (int j = 0; j < 64; j++) {
                if (!(T <= Tp)) { done = true; break; }
                if (!loop_circuit && i0 + base + j + 1 >= n) { done = true; break; }
                T += rl(seg, j);
                m++;
                if (m >= PLAN_MAXM) { over = true; done = true; break; }
            }
        }
    }
    if (over && lane == 0 && err) atomicOr(err, 1);
    auto at = [&](int j) -> int { int i = i0 + j; return (i >= n) ? i - n * (i / n) : i; };
    // ---- does the segment cross the yaw seam?
    bool seam = false;
    for (int j = lane; j + 1 < m; j += 64) seam |= fabs(track[4 * at(j + 1) + 2] - track[4 * at(j) + 2]) > 250.0 * (M_PI / 180.0);
    seam = __any(seam);
    if (seam && m != npts) {
        if (lane == 0) {   // np.unwrap(period = 2*pi): sequential cumulative correction
            double prev = track[4 * at(0) + 2], cum = 0.0;
            sYaw[0] = prev;
            for (int j = 1; j < m; j++) {
                const double p = track[4 * at(j) + 2];
                const double dd = p - prev;
                double t = dd + M_PI;
                double md = fmod(t, 2.0 * M_PI);
                if (md != 0.0 && md < 0.0) md += 2.0 * M_PI;
                double ddmod = md - M_PI;
                if (ddmod == -M_PI && dd > 0.0) ddmod = M_PI;
                double corr = ddmod - dd;
                if (fabs(dd) < M_PI) corr = 0.0;
                cum += corr;
                sYaw[j] = p + cum;
                prev = p;
            }
        }
        __syncthreads();
    }
    // ---- resample: lane j -> sample j
    for (int j = lane; j < npts; j += 64) {
        double r[4];
        if (m == npts) {
            const int i = at(j);
#pragma unroll
            for (int c = 0; c < 4; c++) r[c] = track[4 * i + c];
        } else {
            const double step = (double)(m - 1) / (double)(npts - 1);
            double xs = (double)j * step;
            if (j == npts - 1) xs = (double)(m - 1);
            const int jj = (int)xs;
            if (jj >= m - 1) {
                const int i = at(m - 1);
#pragma unroll
                for (int c = 0; c < 4; c++) r[c] = track[4 * i + c];
                if (seam) r[2] = pymod_2pi(sYaw[m - 1]);
            } else {
                const int ia = at(jj), ib = at(jj + 1);
                const double fx = xs - (double)jj;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    double fa = track[4 * ia + c], fb = track[4 * ib + c];
                    if (c == 2 && seam) { fa = sYaw[jj]; fb = sYaw[jj + 1]; }
                    const double slope = (fb - fa) / 1.0;
                    r[c] = (fx == 0.0) ? fa : slope * fx + fa;
                }
                if (seam) r[2] = pymod_2pi(r[2]);
            }
        }
        double *o = out + ((size_t)b * npts + j) * out_stride;
#pragma unroll
        for (int c = 0; c < 4; c++) o[c] = r[c];
        for (int c = 4; c < out_stride; c++) o[c] = 0.0;
        if (j == 0 && ref0) {
#pragma unroll
            for (int c = 0; c < 4; c++) ref0[(size_t)b * 4 + c] = r[c];
        }
    }
    if (lane == 0 && closest) closest[b] = i0;
    // control-step counter of the closed loop (read by plant_advance_kernel of the same step): kept on the device so
    // that a captured hipGraph of a step can be replayed without re-baking kernel arguments
}

// xdot of the 7-state plant [posx,posy,yaw,vlong,vlat,yawrate,delta_f] with inputs (a, steering rate)
// (sim_model_dynamic_stm_pacejka.py:137-195: the prediction model's forces with the acceleration as an input)
struct PlantModel {
    double lf, lr, m, inv_m, inv_Iz, ka;
    double Bf, Cf, Df, Ef, Br, Cr, Dr, Er;
    double Fz_f, Fz_r, invFmax_f, invFmax_r;          // static axle loads, 1 / (Fz * sqrt(1 + C^2))
    double fr0, fr1, fr4;
};

// Same formulas as the reference's plant, arranged for a short instruction stream: this kernel is ONE pass of latency per
// control step (16 model evaluations in sequence), so what counts is the number of instructions a wavefront issues per
// evaluation. Four lanes (a DPP quad) serve one vehicle and split the transcendental chains of the model between them:
//   role 0  front tyre: slip angle (atan), magic formula (atan, atan, sin), friction-circle factor (sqrt)
//   role 1  rear tyre:  the same chain with the rear constants
//   role 2  sin / cos of the yaw angle          role 3  sin / cos of the steering angle
// Every lane runs the SAME instructions (the tyre chain on roles 2, 3 works on the front / rear operands again and its
// result is ignored; the one sincos serves the magic formula on roles 0, 1 and the angles on roles 2, 3), the four results
// are exchanged by DPP quad broadcasts and the cheap remainder is computed redundantly by all four. Three atan, one sincos
// and one sqrt per evaluation instead of six, four and two; every number is produced by the same operations on the same
// operands as with one lane per vehicle (constants folded on the host, one reciprocal of vlong, cos(asin(G)) = sqrt(1 - G^2)).
struct PlantLane { double B, C, D, E, invFmax; bool front; int role; };

__device__ __forceinline__ void plant_xdot(const PlantModel &p, const PlantLane &t, const double x[7], double a, double sr, double xd[7])
{
    const double yaw = x[2], vl = x[3], vt = x[4], r = x[5], de = x[6];
    const double w = 0.036 * sqrt(vl * vl + vt * vt);          // v[km/h] / 100
    const double w2 = w * w;
    const double fr = p.fr0 + p.fr1 * w + p.fr4 * w2 * w2;
    const double Fx_f = -fr * p.Fz_f;
    const double Fx_r = p.m * a - fr * p.Fz_r;
    double al = 0.0;
    if (vl > 0.001) {
        const double ivl = 1.0 / vl;
        const double nf = vt + p.lf * r, nr = p.lr * r - vt;
        const double at = fast_atan((t.front ? nf : nr) * ivl);
        al = t.front ? de - at : at;
    }
    const double xx = t.B * al;
    const double th = fast_atan(xx - t.E * (xx - fast_atan(xx)));
    double sn, cs;
    fast_sincos(t.role < 2 ? t.C * th : (t.role == 2 ? yaw : de), &sn, &cs);
    const double G = fmin(fmax((t.front ? Fx_f : Fx_r) * t.invFmax, -0.98), 0.98);
    const double Fy = (t.D * sn) * fast_sqrt_pos(1.0 - G * G);
    const double Fy_f = quad_bcast<0>(Fy), Fy_r = quad_bcast<1>(Fy);
    const double sy = quad_bcast<2>(sn), cy = quad_bcast<2>(cs), sd = quad_bcast<3>(sn), cd = quad_bcast<3>(cs);
    const double front = Fy_f * cd + Fx_f * sd;
    xd[0] = vl * cy - vt * sy;
    xd[1] = vl * sy + vt * cy;
    xd[2] = r;
    xd[3] = (Fx_r - p.ka * vl * vl - Fy_f * sd + Fx_f * cd) * p.inv_m + vt * r;
    xd[4] = (Fy_r + front) * p.inv_m - vl * r;
    xd[5] = (p.lf * front - p.lr * Fy_r) * p.inv_Iz;
    xd[6] = sr;
}

struct SimArgs {
    int N, batch, n_elem, log_cap;
    int *step_counter;                                // [0] control steps completed, [1] ticket of the blocks of this launch (device)
    double Ts;
    int win[8];
    PlantModel pm;
    double *X, *U; const double *cost; const int *status, *qp_iter;   // the solver's outputs (the iterate is rewritten on failure)
    // recovery from a failed solve (main.py:59-61 -> reintialize_solver): sample copies of an SNMPC capsule, bounds of an R2 capsule
    double *XS; const double *xs0; int ns;
    double *bnd; int r2; double r2_dmin, r2_dmax, r2_uh;
    double *x_sim, *x0, *pose, *hist;                 // [b][7], [b][8] (capsule x0), [b][2], [b][8][4]
    const double *ref0;                               // [b][4] first reference point of this step (planner output)
    // disturbance realisation played back by the loop (Utils/SimulationMode_main_class.py:121-143; nullable): additive disturbance of
    // the state DERIVATIVES, constant over a control step, and additive STATE ESTIMATION error, [step][b][7] each, dist_len steps
    const double *dist_w, *dist_e; int dist_len;
    double *lCiLX, *lSimX, *lU, *lREF, *lDBG;         // logs (nullable): (cap+1,B,7) (cap+1,B,8) (cap,B,2) (cap,B,4) (cap,B,5)
};

// Four lanes per instance (see plant_xdot): simMode 0 of sim_step. The plant takes the predicted acceleration of stage 1
// and the steering rate of stage 0, integrates Ts with classic RK4 in n_elem equal sub-steps; the estimator is a per-state
// moving average over the last win[i] samples (fewer while the buffer fills); the filtered state becomes the next x0 of
// the OCP. The four lanes of an instance share the stores (estimator states i = role, role + 4; one log each; the stages
// k = role mod 4 of a re-initialisation).
constexpr int PLANT_LANES = 4;
__global__ void __launch_bounds__(64) plant_advance_kernel(const SimArgs sa)
{
#pragma clang fp contract(off)
    const int gl = blockIdx.x * blockDim.x + threadIdx.x;
    // lanes past the batch stay in the kernel (DPP reads of a quad never cross instances; they only repeat the last one's loads)
    const bool live = (gl >> 2) < sa.batch;
    const int b = live ? (gl >> 2) : sa.batch - 1, role = gl & 3;
    const int N = sa.N, B = sa.batch;
    const int step = sa.step_counter[0];              // control steps done before this one
    PlantLane tl;
    tl.role = role; tl.front = !(role & 1);
    tl.B = tl.front ? sa.pm.Bf : sa.pm.Br; tl.C = tl.front ? sa.pm.Cf : sa.pm.Cr; tl.D = tl.front ? sa.pm.Df : sa.pm.Dr;
    tl.E = tl.front ? sa.pm.Ef : sa.pm.Er; tl.invFmax = tl.front ? sa.pm.invFmax_f : sa.pm.invFmax_r;
    double x1[8], u0[2];
#pragma unroll
    for (int i = 0; i < 8; i++) x1[i] = sa.X[((size_t)b * (N + 1) + 1) * NX + i];
    u0[0] = sa.U[(size_t)b * N * NU]; u0[1] = sa.U[(size_t)b * N * NU + 1];
    const double a_in = x1[7], sr_in = u0[1];
    const int st_b = sa.status[b];
    if (st_b != 0 && live) {
        // main.py:59-61: a failed solve is followed by MPC.reintialize_solver(x_next) -- a FRESH solver, cold-started at the
        // state the failed solve started from (x_k = x0 for all k, u = 0; NMPC_class.py:256-267), with the nominal bounds
        // (R2NMPC: the tightening of the previous solves is gone) and the sample copies at their initial conditions (SNMPC).
        // The control applied this step is still the failed solver's u0 / the last good prediction (read above).
        const double *xo = sa.x0 + (size_t)b * NX;
        double xv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) xv[i] = xo[i];
        for (int k = role; k <= N; k += PLANT_LANES)
#pragma unroll
            for (int i = 0; i < 8; i++) sa.X[((size_t)b * (N + 1) + k) * NX + i] = xv[i];
        for (int i = role; i < N * NU; i += PLANT_LANES) sa.U[(size_t)b * N * NU + i] = 0.0;
        if (sa.ns > 0)
            for (int k = role; k <= N; k += PLANT_LANES)
                for (int i = 0; i < sa.ns * NX; i++) sa.XS[((size_t)b * (N + 1) + k) * sa.ns * NX + i] = sa.xs0[(size_t)b * sa.ns * NX + i];
        if (sa.r2) {
            double *bb = sa.bnd + (size_t)b * 6 * (N + 1);
            for (int k = 1 + role; k < N; k += PLANT_LANES) { bb[2 * (N + 1) + k] = sa.r2_dmin; bb[3 * (N + 1) + k] = sa.r2_dmax; bb[5 * (N + 1) + k] = sa.r2_uh; }
        }
    }
    double x[7], xd[7], xs[7], wv[7];
#pragma unroll
    for (int i = 0; i < 7; i++) { xs[i] = sa.x_sim[(size_t)b * 7 + i]; wv[i] = 0.0; x[i] = xs[i]; }
    const double h = sa.Ts / sa.n_elem;
    // sim_step (SimulationMode_main_class.py:112-143): the plant's TRUE next state is the undisturbed step (it is what the logger
    // keeps as CiLX and what the next step starts from); with a disturbance of the state derivatives a SECOND step from the same
    // state, xdot + w (sim_model_dynamic_stm_pacejka.py:196), gives the state the estimator is fed, and the state estimation error is
    // added to that. Both passes share one copy of the integrator (pass loop not unrolled, see below).
    const bool have_w = sa.dist_w && step < sa.dist_len, have_e = sa.dist_e && step < sa.dist_len;
#pragma unroll 1
    for (int pass = 0; pass < (have_w ? 2 : 1); pass++) {
        if (pass == 1) {
#pragma unroll
            for (int i = 0; i < 7; i++) { xd[i] = x[i]; x[i] = xs[i]; wv[i] = sa.dist_w[((size_t)step * B + b) * 7 + i]; }
        }
    for (int e = 0; e < sa.n_elem; e++) {
        // The four RK stages share ONE inlined copy of the model (stage loop not unrolled): this kernel runs its code
        // once per control step, so its time is instruction fetch -- four inlined copies (63 KB) cost 107 us per call,
        // all of it cold instruction-cache misses. State loops are unrolled so the vectors stay in registers.
        double kprev[7], acc[7];
#pragma unroll
        for (int i = 0; i < 7; i++) { kprev[i] = 0.0; acc[i] = 0.0; }
#pragma unroll 1
        for (int st = 0; st < 4; st++) {
            const double ci = (st == 0) ? 0.0 : (st == 3 ? 1.0 : 0.5);
            const double wi = (st == 0 || st == 3) ? 1.0 : 2.0;
            double t[7], k[7];
#pragma unroll
            for (int i = 0; i < 7; i++) t[i] = (st == 0) ? x[i] : x[i] + ci * h * kprev[i];
            plant_xdot(sa.pm, tl, t, a_in, sr_in, k);
            if (pass == 1) {
#pragma unroll
                for (int i = 0; i < 7; i++) k[i] = k[i] + wv[i];
            }
#pragma unroll
            for (int i = 0; i < 7; i++) { acc[i] = (st == 0) ? k[i] : acc[i] + wi * k[i]; kprev[i] = k[i]; }
        }
#pragma unroll
        for (int i = 0; i < 7; i++) x[i] = x[i] + h / 6.0 * acc[i];
    }
    }
    // x: what the estimator sees (disturbed), xd: the true state
    if (have_w) {
#pragma unroll
        for (int i = 0; i < 7; i++) { const double tmp = x[i]; x[i] = xd[i]; xd[i] = tmp; }
    } else {
#pragma unroll
        for (int i = 0; i < 7; i++) xd[i] = x[i];
    }
    if (have_e) {
#pragma unroll
        for (int i = 0; i < 7; i++) xd[i] = xd[i] + sa.dist_e[((size_t)step * B + b) * 7 + i];
    }
    if (live) {
        if (role == 0) {
#pragma unroll
            for (int i = 0; i < 7; i++) sa.x_sim[(size_t)b * 7 + i] = x[i];
            sa.pose[(size_t)b * 2] = x[0]; sa.pose[(size_t)b * 2 + 1] = x[1];
        }
        // state estimation: sample number k (1-based) goes to ring slot (k-1) & 3; this lane filters states role and role + 4
        const int k = step + 1;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = role + 4 * j;
            double v = a_in;
#pragma unroll
            for (int q = 0; q < 7; q++) v = (i == q) ? xd[q] : v;
            double *hst = sa.hist + ((size_t)b * 8 + i) * 4;
            hst[(k - 1) & 3] = v;
            const int wn = (role == 0) ? sa.win[4 * j] : (role == 1) ? sa.win[4 * j + 1] : (role == 2) ? sa.win[4 * j + 2] : sa.win[4 * j + 3];
            const int cnt = (wn < k) ? wn : k;
            double s = (cnt == 1) ? v : hst[(k - cnt) & 3];
            for (int t = k - cnt + 1; t < k; t++) s = s + ((t == k - 1) ? v : hst[t & 3]);
            sa.x0[(size_t)b * NX + i] = s / (double)cnt;
        }
        if (sa.lCiLX && step < sa.log_cap) {
            const size_t s = step;
            if (role == 0) {
#pragma unroll
                for (int i = 0; i < 7; i++) sa.lCiLX[((s + 1) * B + b) * 7 + i] = x[i];
            } else if (role == 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) sa.lSimX[((s + 1) * B + b) * 8 + i] = x1[i];
            } else if (role == 2) {
                sa.lU[(s * B + b) * 2] = u0[0]; sa.lU[(s * B + b) * 2 + 1] = u0[1];
                for (int i = 0; i < 4; i++) sa.lREF[(s * B + b) * 4 + i] = sa.ref0[(size_t)b * 4 + i];
            } else {
                double *d = sa.lDBG + (s * B + b) * 5;
                d[0] = sa.cost[b]; d[1] = 0.0; d[2] = 1.0; d[3] = (double)sa.qp_iter[b]; d[4] = (double)st_b;
            }
        }
    }
    // the last block to get here closes the control step (every block has read the counter by then)
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&sa.step_counter[1], 1) == (int)gridDim.x - 1) { sa.step_counter[1] = 0; atomicAdd(&sa.step_counter[0], 1); }
    }
}

}  // namespace tum
