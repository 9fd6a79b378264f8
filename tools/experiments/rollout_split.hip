// rollout_split_kernel: the re-conditioned rollout with ONE WAVE PER RIGHT-HAND SIDE.  gfx950, wave64.
//
// Why: the chain of a sample is latency-bound on one wave (every pivot of the forward substitution waits ~16 cycles
// per v_readlane, tools/ubench/bcast.hip), and that latency only overlaps ACROSS waves of a SIMD.  The three label
// slots (value, d/dx0, d/dx1) of the test point are three right-hand sides of the same triangular system, so a
// workgroup of three waves runs one sample: wave b owns right-hand side b (and the appended row of task b).  Four
// such workgroups share a CU (LDS-limited: ~40 KB each at H = 30), i.e. three waves per SIMD from different samples
// and different phases.
//
// The two dense products against the real data are split the other way, by COLUMN thirds, so that no wave has to hold
// a full row of L_rr^-1 or L_hr (that would not fit 168 VGPRs): wave w keeps columns [NTH w, NTH w + NTH) of this
// lane's rows of L_rr^-1 and L_hr, forms partial products for all three right-hand sides, and the partials meet in LDS.
//
//   registers : a third of this lane's rows of L_rr^-1, L_hr bank 0 (rows 0..63) and L_hr bank 1 (rows 64..), the
//               wave's right-hand side v0 / v1, 1/L_pp and w_p of the lane's own rows
//   LDS       : k_r then v_r (every wave writes and reads only its own third), the exchange buffers, the nine reduced
//               sums + the published 3x3 results, and L_hh (layout: lhh_layout.hpp)
//
// Five workgroup barriers per step (3 waves each):
//   #1 partial products of L_rr^-1 k_r exchanged     #2 partial products of L_hr v_r exchanged
//   #3 v_h of all right-hand sides visible           #4 the nine reduced sums visible
//   #5 sample / chol(S + noise) published (wave 0 / wave 1)
#include "gpmpc_host.hpp"
#include "lhh_layout.hpp"
#include "rollout_args.hpp"

namespace gpmpc {

__device__ long long g_split_phase_cycles[16];

#ifdef GPMPC_PHASE_TIMERS
#define SPHASE_DECL                                   \
    long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    long long tph = __builtin_readcyclecounter()
#define SPHASE(idx)                                              \
    do {                                                         \
        const long long _n = __builtin_readcyclecounter();       \
        ph[idx] += _n - tph;                                     \
        tph = _n;                                                \
    } while (0)
#define SPHASE_STORE                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0)                     \
        for (int i = 0; i < 10; ++i) g_split_phase_cycles[i] = ph[i]
#else
#define SPHASE_DECL
#define SPHASE(idx)
#define SPHASE_STORE
#endif

constexpr int kSplitRed = 20;   // doubles: [0..8] reduced sums, [10..12] y, [13..15] 1/C_jj, [16..18] C10 C20 C21

// cov(task_a(x), task_b(x')) = k * (fa * fb + [a == b > 0] / l_a^2),  fa = a ? -q[a-1] : 1,  fb = b ? q[b-1] : 1
// (same values as kern_entry; this form takes run-time task indices without divergent control flow)
__device__ __forceinline__ double kern_entry_rt(const double (&q)[2], double k, const double* il2, int a, int b) {
    const double fa = (a == 0) ? 1.0 : ((a == 1) ? -q[0] : -q[1]);
    const double fb = (b == 0) ? 1.0 : ((b == 1) ? q[0] : q[1]);
    const double dg = (a == b && a > 0) ? ((a == 1) ? il2[0] : il2[1]) : 0.0;
    return k * fma(fa, fb, dg);
}

template <int NR, int ENV>
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(3, 3))) void rollout_split_kernel(const RolloutArgs a) {
    constexpr int T = 3, D = 2;
    constexpr int NX = (ENV == GPMPC_ENV_PENDULUM1D) ? 2 : 4;
    constexpr int NU = (ENV == GPMPC_ENV_PENDULUM1D) ? 1 : 2;
    constexpr int NTH = (((NR + 2) / 3) + 1) & ~1;                // columns of L_rr^-1 per wave (even)
    constexpr int NRP = 3 * NTH;                                  // k_r / v_r row length
    constexpr int NPAIR = NR / 2;
    static_assert(9 * NRP >= 3 * kWave, "v_h bank 0 re-uses the partial-product area");
    static_assert(NR % 2 == 0, "odd N_r needs a tail term");
    extern __shared__ __attribute__((aligned(16))) double smem[];

    const GpParams& gp = a.gp;
    // b: this wave's right-hand side / appended task / column third; wave-uniform, so keep it (and every address
    // derived from it) in SGPRs
    const int b = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long s = blockIdx.x;
    const int H = a.H, nh_max = a.nh_max;
    const int nhp = (nh_max + 1) & ~1;
    const int nb1p = (nhp > kWave) ? nhp - kWave : 0;

    double* kvs = smem;                                           // [T][NRP]
    // px : [w][c][NRP]  wave w's partial of (L_rr^-1 k_r)[c]; after barrier #2 the same area carries v_h bank 0 [T][64]
    // cx : [6][64 + nb1p]  foreign partials of (L_hr v_r)[c]: slots 2c, 2c+1 in increasing wave index
    // vx1: [T][nb1p]  v_h bank 1
    double* px = kvs + T * NRP;
    double* cx = px + 9 * NRP;
    double* vx1 = cx + 6 * (kWave + nb1p);
    double* red = vx1 + T * nb1p;                                 // [kSplitRed]
    double* xhL = red + kSplitRed;                                // [nhp][2]  GP input of the point each appended row belongs to
    double* Lhh = xhL + 2 * nhp;                                  // lhh_doubles(nh_max)
    const int cxs = kWave + nb1p;

    for (int e = threadIdx.x; e < a.lds_per_wave; e += blockDim.x) smem[e] = 0.0;   // lds_per_wave: doubles per workgroup

    double il2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) il2[d] = gp.inv_l2[0][d];
    const double os = gp.os[0];
    // A phase: lanes 0..NTH-1 of wave b evaluate the kernel at real point NTH b + lane
    const int jpt = NTH * b + lane;
    const bool a_act = (lane < NTH) && (jpt < NR);
    double xr[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xr[d] = a_act ? a.X_r[jpt * D + d] : 0.0;
    const double w_lane = (lane < NR) ? plan_w(a.plan, gp, 0)[lane] : 0.0;
    // L_rr^-1[lane][NTH b + jj] is re-read from the plan (L1/L2-resident, 10 KB) every step: 2 NTH registers that are
    // only live during phases A-B instead of persistent ones
    const double* linv_col = plan_LinvT(a.plan, gp, 0) + (long)NTH * b * NR + ((lane < NR) ? lane : 0);
    // bank-1 rows of L_hr (this wave's column third) live in the HBM/L2 workspace: [sample][wave][row][NTH]
    double* lhr1_ws = a.ws + (((long)s * T + b) * max(nb1p, 1) + min(lane, max(nb1p - 1, 0))) * NTH;

    double x[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) x[d] = a.x0[(a.x0_per_sample ? s * NX : 0) + d];

    double Lhr0[NTH];                                             // L_hr[lane][NTH b + ii]
#pragma unroll
    for (int i = 0; i < NTH; ++i) Lhr0[i] = 0.0;
    double dinv0 = 0.0, dinv1 = 0.0, wown0 = 0.0, wown1 = 0.0;
    int info_acc = 0;
    int n_h = 0;
    const double2_t* row0 = reinterpret_cast<const double2_t*>(Lhh + lhh_rowofs(min(lane, nh_max - 1)));
    const double2_t* row1 = reinterpret_cast<const double2_t*>(Lhh + lhh_rowofs(min(lane + kWave, nh_max - 1)));
    const int a0t = lane - (lane / T) * T, a1t = (lane + kWave) - ((lane + kWave) / T) * T;
    const bool own_third = (lane >= NTH * b) && (lane < NTH * b + NTH) && (lane < NR);
    const int bn = (b == 2) ? 0 : b + 1;                          // the neighbour whose cross term this wave reduces
    SPHASE_DECL;
    __syncthreads();

#pragma unroll 1
    for (int t = 0; t < H; ++t) {
        // ---- input, GP input (every wave) ------------------------------------------------------------------------
        double u[NU], xi[D];
        {
            const double* uf = a.u_ff + (long)t * NU;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                if (a.env.use_feedback) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc += (a.env.x_goal[j] - x[j]) * a.env.K[i][j];
                    u[i] = -acc + uf[i];
                } else {
                    u[i] = uf[i];
                }
            }
            xi[0] = (ENV == GPMPC_ENV_PENDULUM1D) ? x[0] : x[2];
            xi[1] = u[0];
        }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + t] = x[d];
            if (a.Xi) {
#pragma unroll
                for (int d = 0; d < D; ++d) a.Xi[(s * H + t) * D + d] = xi[d];
            }
        }

        // ---- A: this wave's third of k_r, all three tasks ------------------------------------------------------------
        double lth[NTH];
        {
            const double* lp = linv_col;
            asm volatile("" : "+v"(lp));                               // keep the loads inside the step loop
#pragma unroll
            for (int jj = 0; jj < NTH; ++jj) lth[jj] = (NTH * b + jj < NR) ? lp[jj * NR] : 0.0;
        }
        {
            double q[D];
            const double k = kern_scalar<D>(xr, xi, il2, os, q);
            if (a_act) {
#pragma unroll
                for (int c = 0; c < T; ++c) kvs[c * NRP + jpt] = kern_entry<D>(q, k, il2, 0, c);
            }
        }
        wave_sync_lds();
        SPHASE(0);

        // ---- B: partial products over this wave's columns, exchanged through LDS -------------------------------------
        double vr[T];
        {
            double part[T] = {0.0, 0.0, 0.0};
            constexpr int JB = 2;
#pragma unroll
            for (int jp0 = 0; jp0 < NTH / 2; jp0 += JB) {
                double2_t kk[JB][T];
#pragma unroll
                for (int q = 0; q < JB; ++q)
                    if (jp0 + q < NTH / 2) {
#pragma unroll
                        for (int c = 0; c < T; ++c)
                            kk[q][c] = *reinterpret_cast<const double2_t*>(kvs + c * NRP + NTH * b + 2 * (jp0 + q));
                    }
#pragma unroll
                for (int q = 0; q < JB; ++q)
                    if (jp0 + q < NTH / 2) {
#pragma unroll
                        for (int c = 0; c < T; ++c) {
                            part[c] = fma(lth[2 * (jp0 + q)], kk[q][c].x, part[c]);
                            part[c] = fma(lth[2 * (jp0 + q) + 1], kk[q][c].y, part[c]);
                        }
                    }
                asm volatile("" ::: "memory");
            }
            if (lane < NR) {
#pragma unroll
                for (int c = 0; c < T; ++c) px[(b * T + c) * NRP + lane] = part[c];
            }
            __syncthreads();                                           // #1
#pragma unroll
            for (int c = 0; c < T; ++c) {
                double acc = 0.0;
                if (lane < NR) {
                    const double p0 = px[(0 * T + c) * NRP + lane], p1 = px[(1 * T + c) * NRP + lane];
                    const double p2 = px[(2 * T + c) * NRP + lane];
                    acc = (p0 + p1) + p2;                              // every wave forms the same v_r (waves 0,1,2 in order)
                }
                vr[c] = acc;
                if (own_third) kvs[c * NRP + lane] = vr[c];            // own third: k_r -> v_r (only this wave reads it)
            }
        }
        wave_sync_lds();
        SPHASE(1);

        double v0 = 0.0, v1 = 0.0;                                    // rows lane / lane+64 of v_h, right-hand side b
        const bool two = n_h > kWave;
        if (n_h > 0) {
            const bool ex0 = lane < n_h, ex1 = lane + kWave < n_h;
            // ---- C: rhs = k_h - L_hr v_r ----------------------------------------------------------------------------
            double2_t l1[NTH / 2];
            if (two) {
#pragma unroll
                for (int ip = 0; ip < NTH / 2; ++ip) l1[ip] = reinterpret_cast<const double2_t*>(lhr1_ws)[ip];
            }
            {
                const double2_t xh = *reinterpret_cast<const double2_t*>(xhL + 2 * min(lane, nhp - 1));
                const double xhv[D] = {xh.x, xh.y};
                double q[D];
                const double k = kern_scalar<D>(xhv, xi, il2, os, q);
                v0 = ex0 ? kern_entry_rt(q, k, il2, a0t, b) : 0.0;
            }
            if (two) {
                const double2_t xh = *reinterpret_cast<const double2_t*>(xhL + 2 * min(lane + kWave, nhp - 1));
                const double xhv[D] = {xh.x, xh.y};
                double q[D];
                const double k = kern_scalar<D>(xhv, xi, il2, os, q);
                v1 = ex1 ? kern_entry_rt(q, k, il2, a1t, b) : 0.0;
            }
            {
                double cp0[T] = {0.0, 0.0, 0.0}, cp1[T] = {0.0, 0.0, 0.0};
                constexpr int IB = 2;
#pragma unroll
                for (int ip0 = 0; ip0 < NTH / 2; ip0 += IB) {
                    double2_t vv[IB][T];
#pragma unroll
                    for (int q = 0; q < IB; ++q)
                        if (ip0 + q < NTH / 2) {
#pragma unroll
                            for (int c = 0; c < T; ++c)
                                vv[q][c] = *reinterpret_cast<const double2_t*>(kvs + c * NRP + NTH * b + 2 * (ip0 + q));
                        }
#pragma unroll
                    for (int q = 0; q < IB; ++q)
                        if (ip0 + q < NTH / 2) {
#pragma unroll
                            for (int c = 0; c < T; ++c) {
                                cp0[c] = fma(Lhr0[2 * (ip0 + q)], vv[q][c].x, cp0[c]);
                                cp0[c] = fma(Lhr0[2 * (ip0 + q) + 1], vv[q][c].y, cp0[c]);
                                if (two) {
                                    cp1[c] = fma(l1[ip0 + q].x, vv[q][c].x, cp1[c]);
                                    cp1[c] = fma(l1[ip0 + q].y, vv[q][c].y, cp1[c]);
                                }
                            }
                        }
                    asm volatile("" ::: "memory");
                }
                double own0 = cp0[0], own1 = cp1[0];
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    if (c == b) {
                        own0 = cp0[c];
                        own1 = cp1[c];
                    } else {
                        const int slot = 2 * c + ((b > c) ? b - 1 : b);
                        cx[slot * cxs + lane] = cp0[c];
                        if (two && lane < nb1p) cx[slot * cxs + kWave + lane] = cp1[c];
                    }
                }
                __syncthreads();                                       // #2
                {
                    const double q0 = cx[(2 * b) * cxs + lane], q1 = cx[(2 * b + 1) * cxs + lane];
                    v0 -= (b == 2) ? (q0 + q1) + own0 : (own0 + q0) + q1;
                }
                if (two && lane < nb1p) {
                    const double q0 = cx[(2 * b) * cxs + kWave + lane], q1 = cx[(2 * b + 1) * cxs + kWave + lane];
                    v1 -= (b == 2) ? (q0 + q1) + own1 : (own1 + q0) + q1;
                }
            }
            SPHASE(2);

            // ---- D: forward substitution for one right-hand side (ring of 4 row pairs per bank) ------------------------
            if (!two) {
                double2_t ra[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) ra[k] = row0[k];
#pragma unroll 1
                for (int p0 = 0; p0 < n_h; p0 += 8) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double2_t la2 = ra[k];
                        ra[k] = row0[(p0 >> 1) + 4 + k];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int p = p0 + 2 * k + h;
                            const double sp = readlane_f64(v0, p);
                            const double la = (lane > p) ? (h ? la2.y : la2.x) : 0.0;
                            v0 = fma(-la, sp, v0);
                        }
                    }
                }
            } else {
                double2_t ra[4], rb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ra[k] = row0[k];
                    rb[k] = row1[k];
                }
#pragma unroll 1
                for (int p0 = 0; p0 < kWave; p0 += 8) {                  // pivots owned by bank 0
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double2_t la2 = ra[k], lb2 = rb[k];
                        ra[k] = row0[(p0 >> 1) + 4 + k];
                        rb[k] = row1[(p0 >> 1) + 4 + k];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int p = p0 + 2 * k + h;
                            const double sp = readlane_f64(v0, p);
                            const double la = (lane > p) ? (h ? la2.y : la2.x) : 0.0;
                            const double lb = h ? lb2.y : lb2.x;
                            v0 = fma(-la, sp, v0);
                            v1 = fma(-lb, sp, v1);
                        }
                    }
                }
#pragma unroll 1
                for (int p0 = kWave; p0 < n_h; p0 += 8) {                 // pivots owned by bank 1
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double2_t lb2 = rb[k];
                        rb[k] = row1[(p0 >> 1) + 4 + k];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int p = p0 + 2 * k + h;
                            const double sp = readlane_f64(v1, p - kWave);
                            const double lb = (lane + kWave > p) ? (h ? lb2.y : lb2.x) : 0.0;
                            v1 = fma(-lb, sp, v1);
                        }
                    }
                }
            }
            v0 = ex0 ? v0 * dinv0 : 0.0;                              // v = rhs / L_pp; rows that do not exist are dropped
            v1 = ex1 ? v1 * dinv1 : 0.0;
            px[b * kWave + lane] = v0;                                 // every wave is past its px reads (#2)
            if (two && lane < nb1p) vx1[b * nb1p + lane] = v1;
        }
        SPHASE(3);
        __syncthreads();                                               // #3

        // ---- E: this wave's three sums; the L_hr rows of the slots appended at the end of this step -------------------
        {
            double oc0 = 0.0, oc1 = 0.0;
            const double vrb = (b == 0) ? vr[0] : ((b == 1) ? vr[1] : vr[2]);
            const double ocr = (b == 0) ? vr[1] : ((b == 1) ? vr[2] : vr[0]);
            if (n_h > 0) {
                oc0 = px[bn * kWave + lane];
                if (two) oc1 = (lane < nb1p) ? vx1[bn * nb1p + lane] : 0.0;
            }
            const double pm = fma(v1, wown1, fma(v0, wown0, vrb * w_lane));
            const double pbb = fma(v1, v1, fma(v0, v0, vrb * vrb));
            const double pbc = fma(v1, oc1, fma(v0, oc0, vrb * ocr));
            const double r0 = wave_sum(pm), r1 = wave_sum(pbb), r2 = wave_sum(pbc);
            if (lane == 0) {
                red[3 * b] = r0;
                red[3 * b + 1] = r1;
                red[3 * b + 2] = r2;
            }
            if (t + 1 < H) {
                const int base = n_h;
                const int a0 = lane - base, a1 = lane + kWave - base;
                // the new rows' L_hr entries are v_r^T: this wave's column third comes from its own third of kvs
                if (a0 >= 0 && a0 < T) {
                    const double* src = kvs + a0 * NRP + NTH * b;
#pragma unroll
                    for (int ip = 0; ip < NTH / 2; ++ip) {
                        const double2_t vv = *reinterpret_cast<const double2_t*>(src + 2 * ip);
                        Lhr0[2 * ip] = vv.x;
                        Lhr0[2 * ip + 1] = vv.y;
                    }
                }
                if (a1 >= 0 && a1 < T) {                                // bank 1: lane's own workspace row (only this lane reads it)
                    const double* src = kvs + a1 * NRP + NTH * b;
#pragma unroll
                    for (int ip = 0; ip < NTH / 2; ++ip)
                        reinterpret_cast<double2_t*>(lhr1_ws)[ip] = *reinterpret_cast<const double2_t*>(src + 2 * ip);
                }
            }
        }
        SPHASE(4);
        __syncthreads();                                               // #4

        // ---- F: wave 0 draws the sample, wave 1 factors S + noise ------------------------------------------------------
        {
            double mu[T], S[T][T];
#pragma unroll
            for (int c = 0; c < T; ++c) {
                mu[c] = red[3 * c];
                S[c][c] = ((c == 0) ? os : os * il2[c - 1]) - red[3 * c + 1];
            }
            S[1][0] = S[0][1] = -red[2];      // wave 0: (0, 1)
            S[2][1] = S[1][2] = -red[5];      // wave 1: (1, 2)
            S[2][0] = S[0][2] = -red[8];      // wave 2: (2, 0)
            if (b == 0) {
                double var[T];
                bool all_zero = (a.var_zero_thr >= 0.0);
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    var[c] = S[c][c];
                    if (var[c] < gp.var_floor) {
                        var[c] = gp.var_floor;
                        info_acc |= GPMPC_INFO_VAR_CLAMPED;
                    }
                    all_zero = all_zero && (var[c] <= a.var_zero_thr);
                }
                double R[T][T];
                info_acc |= root_small_fast<T>(S, gp.jitter, R);
                const double* zt = a.z + (long)t * a.z_step_stride + s * T;
                double y[T];
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int e = 0; e <= c; ++e) acc = fma(R[c][e], zt[e], acc);
                    double yb = acc + mu[c];
                    if (all_zero) yb = mu[c];
                    const double dlt = yb - mu[c];
                    if (dlt * dlt > a.beta * a.beta * var[c]) {
                        const double sd = a.beta * sqrt(var[c]);
                        yb = fmin(fmax(yb, mu[c] - sd), mu[c] + sd);
                    }
                    y[c] = yb;
                }
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < T; ++c) red[10 + c] = y[c];
                    if (a.Y) {
#pragma unroll
                        for (int c = 0; c < T; ++c) a.Y[(s * H + t) * T + c] = y[c];
                    }
                }
            } else if (b == 1) {
                double Sn[T][T], C[T][T], cinv[T];
#pragma unroll
                for (int c = 0; c < T; ++c)
#pragma unroll
                    for (int e = 0; e < T; ++e) Sn[c][e] = S[c][e] + ((c == e) ? gp.noise[c] : 0.0);
                const bool c_ok = chol_small_fast<T>(Sn, C, cinv);
                if (!c_ok && t + 1 < H) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < T; ++c) red[13 + c] = cinv[c];
                    red[16] = C[1][0];
                    red[17] = C[2][0];
                    red[18] = C[2][1];
                }
            }
        }
        SPHASE(5);
        __syncthreads();                                               // #5

        // ---- G: append [v^T, chol(S + noise)] and w to the factor (A.9) ------------------------------------------------
        const double y0 = red[10];
        if (t + 1 < H) {
            double wn[T], cinv[T], mu[T];
#pragma unroll
            for (int c = 0; c < T; ++c) {
                cinv[c] = red[13 + c];
                mu[c] = red[3 * c];
            }
            const double C10 = red[16], C20 = red[17], C21 = red[18];
            wn[0] = (y0 - mu[0]) * cinv[0];
            wn[1] = fma(-C10, wn[0], red[11] - mu[1]) * cinv[1];
            wn[2] = fma(-C21, wn[1], fma(-C20, wn[0], red[12] - mu[2])) * cinv[2];
            const int base = n_h;
            // (1) new row base+b of L'': lane p owns the entry in column p
            double* rowc = Lhh + lhh_rowofs(base + b);
            if (lane < base) rowc[lane] = v0 * dinv0;
            if (lane + kWave < base) rowc[lane + kWave] = v1 * dinv1;
            // (2) its part of the new diagonal block, column-scaled
            if (lane == 0) {
                if (b == 1) rowc[base] = C10 * cinv[0];
                if (b == 2) {
                    rowc[base] = C20 * cinv[0];
                    rowc[base + 1] = C21 * cinv[1];
                }
            }
            // (3) owners of the new rows (every wave keeps its own copy): 1/L_pp, w_p, the point's GP input
            {
                const int a0 = lane - base;
                const int a1 = lane + kWave - base;
                const bool new0 = (a0 >= 0 && a0 < T), new1 = (a1 >= 0 && a1 < T);
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    if (new0 && a0 == c) {
                        dinv0 = cinv[c];
                        wown0 = wn[c];
                    }
                    if (new1 && a1 == c) {
                        dinv1 = cinv[c];
                        wown1 = wn[c];
                    }
                }
                if (b == 0 && lane < T) {
                    xhL[2 * (base + lane)] = xi[0];
                    xhL[2 * (base + lane) + 1] = xi[1];
                }
            }
            n_h += T;
        }

        // ---- state hand-over ---------------------------------------------------------------------------------------
        if (ENV == GPMPC_ENV_PENDULUM1D) {
            const double x0n = x[0] + x[1] * a.env.dt;
            x[1] = x[1] + y0;
            x[0] = x0n;
        }
        SPHASE(6);
    }

    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + H] = x[d];
    }
    __syncthreads();
    int* ired = reinterpret_cast<int*>(red);
    if (lane == 0) ired[b] = info_acc;
    __syncthreads();
    if (threadIdx.x == 0) a.info[s] = ired[0] | ired[1] | ired[2];
    SPHASE_STORE;
}

// ---------------------------------------------------------------------------------------------------------------
// host side: eligibility + launch
// ---------------------------------------------------------------------------------------------------------------
static long split_lds_doubles(int NR, int H) {
    const int T = 3;
    const int NTH = (((NR + 2) / 3) + 1) & ~1, NRP = 3 * NTH;
    const int nh_max = 3 * (H - 1), nhp = (nh_max + 1) & ~1;
    const int nb1p = (nhp > kWave) ? nhp - kWave : 0;
    return (long)T * NRP + 9 * NRP + 6 * (kWave + nb1p) + T * nb1p + kSplitRed + 2 * nhp + lhh_doubles(nh_max);
}

bool rollout_split_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode, int hall_tasks, int H) {
    const char* e = std::getenv("GPMPC_ENABLE_SPLIT_ROLLOUT");      // experimental: slower than rollout_fast
    if (!(e && e[0] == '1')) return false;
    if (!rollout_fast_eligible(gp, env, mode, hall_tasks, H)) return false;
    if (env->env_id != GPMPC_ENV_PENDULUM1D) return false;
    return (size_t)split_lds_doubles(gp->N_r, H) * sizeof(double) <= 160 * 1024 - 64;
}

size_t rollout_split_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int H) {
    const int NTH = (((gp->N_r + 2) / 3) + 1) & ~1;
    const int nhp = (3 * (H - 1) + 1) & ~1;
    const int nb1p = (nhp > kWave) ? nhp - kWave : 0;
    return (size_t)Ns * 3 * (size_t)nb1p * NTH * sizeof(double);      // 0: every row fits bank 0, the workspace is not touched
}

int rollout_split_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, void* ws,
                         size_t ws_bytes, hipStream_t st) {
    (void)env;
    const size_t need = rollout_split_workspace_bytes(gp, args.Ns, args.H);
    if (need && (!ws || ws_bytes < need))
        return fail(GPMPC_E_WORKSPACE, "gpmpc_rollout: workspace too small");
    args.nh_max = 3 * (args.H - 1);
    const long dbl = split_lds_doubles(gp->N_r, args.H);
    args.lds_shared = 0;
    args.lds_per_wave = (int)dbl;                 // doubles per workgroup (zero-initialised by the kernel)
    args.linv_in_lds = 0;
    const size_t lds_bytes = (size_t)dbl * sizeof(double);
    auto k = rollout_split_kernel<36, GPMPC_ENV_PENDULUM1D>;
    GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k, dim3((unsigned)args.Ns), dim3(192), lds_bytes, st, args);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_split_phases(long long* out /*[host] 16*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_split_phase_cycles), 16 * sizeof(long long)));
    return GPMPC_OK;
}
