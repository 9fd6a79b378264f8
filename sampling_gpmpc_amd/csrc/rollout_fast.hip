// rollout_fast_kernel: the tuned re-conditioned rollout for the BASELINE shapes (T = 3 label slots, n_h <= 128
// appended slots, value-only real labels on an N_r = 36 / 45 grid).  gfx950, wave64.
//
// One wave per (sample, output) chain; a workgroup holds WAVES = G_NY * SPW chains (pendulum1D: 4 independent samples,
// car: the 3 outputs of one sample, which meet at one barrier per step).  Per chain:
//
//   registers : the chain's rows of L_hr (row = lane, lane+64; NR doubles each), the running right-hand side,
//               1/L_pp and w_p of the lane's own rows
//   LDS       : L_rr^-1 (shared by the workgroup's chains of the same output), k_r / v_r broadcast buffers, and
//               - when it fits (LHH_LDS) - the chain's L_hh; otherwise L_hh streams from an HBM/L2 workspace
//
// L_hh is stored column-major and COLUMN-SCALED (L''[i][p] = L[i][p] / L[p][p]): the forward substitution
//   rhs_i -= L''[i][p] * rhs_p     (p = 0 .. n_h-1)
// then has a dependency chain of exactly v_readlane -> v_fma_f64 per pivot (no divide, no LDS access on the chain);
// v_p = rhs_p / L_pp is formed once after the loop.  Columns are prefetched PF pivots ahead into a register ring, and
// the pivot loop is split at the 64-row bank boundary so no per-pivot select is needed.
#include "gpmpc_host.hpp"
#include "rollout_args.hpp"

namespace gpmpc {

__device__ long long g_fast_phase_cycles[16];

template <int T, int NR, int G_NY, int ENV, bool LHH_LDS>
__global__ __launch_bounds__(256, 1) void rollout_fast_kernel(const RolloutArgs a) {
    constexpr int D = 2;
    constexpr int NS = T * (T + 1) / 2;
    constexpr int NX = (ENV == GPMPC_ENV_PENDULUM1D) ? 2 : 4;
    constexpr int NU = (ENV == GPMPC_ENV_PENDULUM1D) ? 1 : 2;
    constexpr int PF = LHH_LDS ? 4 : 8;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_info[4];

    const GpParams& gp = a.gp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int spw = (blockDim.x >> 6) / G_NY;           // samples per workgroup
    const int sw = wave / G_NY, o = wave - sw * G_NY;
    const long s = (long)blockIdx.x * spw + sw;
    const bool valid = s < a.Ns;
    const int H = a.H, nh_max = a.nh_max;

    // ---- LDS carve ---------------------------------------------------------------------------------------------
    double* LinvT_all = smem;                                   // [G_NY][NR*NR]
    double* wr_all = LinvT_all + G_NY * NR * NR;                // [G_NY][NR]
    double* ybuf_all = wr_all + G_NY * NR;                      // [spw][2][G_NY]
    double* wb = smem + a.lds_shared + (long)wave * a.lds_per_wave;
    double* krs = wb;                                           // [T][NR]
    double* vrs = krs + T * NR;                                 // [T][NR]
    double* Xh = vrs + T * NR;                                  // [H][D]
    double* xbuf = Xh + H * D;                                  // [NX][H+1]
    double* yout = xbuf + NX * (H + 1);                         // [H][T]
    double* Lhh = LHH_LDS ? (yout + H * T) : (a.ws + (s * G_NY + o) * a.ws_chain_stride);
    const double* LinvT = LinvT_all + o * NR * NR;
    const double* w_r = wr_all + o * NR;
    double* ybuf = ybuf_all + sw * 2 * G_NY;

    for (int e = threadIdx.x; e < G_NY * NR * NR; e += blockDim.x) {
        const int oo = e / (NR * NR);
        LinvT_all[e] = plan_LinvT(a.plan, gp, oo)[e - oo * NR * NR];
    }
    for (int e = threadIdx.x; e < G_NY * NR; e += blockDim.x) {
        const int oo = e / NR;
        wr_all[e] = plan_w(a.plan, gp, oo)[e - oo * NR];
    }
    if (threadIdx.x < 4) s_info[threadIdx.x] = 0;
    __syncthreads();
    if (!valid) return;                                 // whole workgroups are valid when G_NY > 1 (spw == 1)

    for (long e = lane; e < a.ws_chain_stride; e += kWave) Lhh[e] = 0.0;   // see the substitution loop
    double il2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) il2[d] = gp.inv_l2[o][d];
    const double os = gp.os[o];
    double xr[D];                                       // this lane's real training input
#pragma unroll
    for (int d = 0; d < D; ++d) xr[d] = (lane < NR) ? a.X_r[lane * D + d] : 0.0;
    const double w_lane = (lane < NR) ? w_r[lane] : 0.0;

    double x[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) x[d] = valid ? a.x0[(a.x0_per_sample ? s * NX : 0) + d] : 0.0;

    double Lhr0[NR], Lhr1[NR];                          // rows lane / lane+64 of L_hr
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        Lhr0[i] = 0.0;
        Lhr1[i] = 0.0;
    }
    double dinv0 = 0.0, dinv1 = 0.0, wown0 = 0.0, wown1 = 0.0;
    int info_acc = 0;
    int n_h = 0;
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tph = __builtin_readcyclecounter();
#define FPHASE(idx)                                              \
    do {                                                         \
        const long long _n = __builtin_readcyclecounter();       \
        ph[idx] += _n - tph;                                     \
        tph = _n;                                                \
    } while (0)

#pragma unroll 1
    for (int t = 0; t < H; ++t) {
        // ---- input, GP input -----------------------------------------------------------------------------------
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
            if (ENV == GPMPC_ENV_PENDULUM1D) {
                xi[0] = x[0];
                xi[1] = u[0];
            } else {
                xi[0] = x[2];
                xi[1] = u[0];
            }
        }
        if (lane == 0) {
            if (o == 0) {
#pragma unroll
                for (int d = 0; d < NX; ++d) xbuf[d * (H + 1) + t] = x[d];
            }
#pragma unroll
            for (int d = 0; d < D; ++d) Xh[t * D + d] = xi[d];
        }

        // ---- kernel row against the real data (lane = real point) ---------------------------------------------
        {
            double q[D];
            const double k = kern_scalar<D>(xr, xi, il2, os, q);
            if (lane < NR) {
#pragma unroll
                for (int b = 0; b < T; ++b) krs[b * NR + lane] = kern_entry<D>(q, k, il2, 0, b);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        FPHASE(0);

        // ---- v_r = L_rr^-1 k_r (lane = row) ---------------------------------------------------------------------
        double vr[T];
#pragma unroll
        for (int b = 0; b < T; ++b) vr[b] = 0.0;
        {
            const int li = (lane < NR) ? lane : 0;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const double l = LinvT[j * NR + li];     // zero above the diagonal
#pragma unroll
                for (int b = 0; b < T; ++b) vr[b] = fma(l, krs[b * NR + j], vr[b]);
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // keep <= 32 LDS loads in flight (VGPR budget)
            }
            if (lane >= NR) {
#pragma unroll
                for (int b = 0; b < T; ++b) vr[b] = 0.0;
            }
        }
        double pm[T], pss[NS];
        {
            int e = 0;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                if (lane < NR) vrs[b * NR + lane] = vr[b];
                pm[b] = vr[b] * w_lane;
#pragma unroll
                for (int c = 0; c <= b; ++c) pss[e++] = vr[b] * vr[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        FPHASE(1);

        double v0[T], v1[T];                             // rows lane / lane+64 of v_h
#pragma unroll
        for (int b = 0; b < T; ++b) {
            v0[b] = 0.0;
            v1[b] = 0.0;
        }
        if (n_h > 0) {
            // ---- rhs = k_h - L_hr v_r -----------------------------------------------------------------------------
            const bool two = n_h > kWave;
            {
                const int j0 = (lane < n_h) ? lane / T : 0, a0 = lane - (lane / T) * T;
                double q[D];
                const double k = kern_scalar<D>(Xh + j0 * D, xi, il2, os, q);
#pragma unroll
                for (int b = 0; b < T; ++b) v0[b] = (lane < n_h) ? kern_entry<D>(q, k, il2, a0, b) : 0.0;
            }
            if (two) {
                const int sl = lane + kWave;
                const int j1 = (sl < n_h) ? sl / T : 0, a1 = sl - (sl / T) * T;
                double q[D];
                const double k = kern_scalar<D>(Xh + j1 * D, xi, il2, os, q);
#pragma unroll
                for (int b = 0; b < T; ++b) v1[b] = (sl < n_h) ? kern_entry<D>(q, k, il2, a1, b) : 0.0;
            }
            if (two) {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
#pragma unroll
                    for (int b = 0; b < T; ++b) {
                        const double vb = vrs[b * NR + i];
                        v0[b] = fma(-Lhr0[i], vb, v0[b]);
                        v1[b] = fma(-Lhr1[i], vb, v1[b]);
                    }
                    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
#pragma unroll
                    for (int b = 0; b < T; ++b) v0[b] = fma(-Lhr0[i], vrs[b * NR + i], v0[b]);
                    if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
            FPHASE(2);

            // ---- forward substitution with column-scaled L_hh ------------------------------------------------------
            // L_hh storage is zero-initialised and its diagonal slots are never written, so a load of column p with
            // the row clamped into [p, nh_max-1] returns L''[row][p] for p < row < n_h and exactly 0.0 otherwise:
            // no select, no branch, every load unconditional (the compiler keeps them in flight across pivots).
            // Pivots beyond n_h are harmless no-ops (their broadcast value and their column are both zero).
            const int rmax = nh_max - 1;
            auto colA = [&](int p) -> long { return col_ofs(p, nh_max) - p; };     // element (row, p) at colA(p) + row
            auto ld0 = [&](int p) -> double {                                       // L''[lane][p]
                const int pc = min(p, rmax);
                return Lhh[colA(pc) + min(max(lane, pc), rmax)];
            };
            auto ld1 = [&](int p) -> double {                                       // L''[lane+64][p]
                const int pc = min(p, rmax);
                return Lhh[colA(pc) + min(max(lane + kWave, pc), rmax)];
            };
            if (!two) {
                double r0[PF];
#pragma unroll
                for (int k = 0; k < PF; ++k) r0[k] = ld0(k);
#pragma unroll 1
                for (int p0 = 0; p0 < n_h; p0 += PF) {
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        const int p = p0 + k;
                        const double la = r0[k];
                        r0[k] = ld0(p + PF);
#pragma unroll
                        for (int b = 0; b < T; ++b) v0[b] = fma(-la, readlane_f64(v0[b], p), v0[b]);
                    }
                }
            } else {
                double r0[PF], r1[PF];
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    r0[k] = ld0(k);
                    r1[k] = ld1(k);
                }
#pragma unroll 1
                for (int p0 = 0; p0 < kWave; p0 += PF) {                // pivots owned by bank 0
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        const int p = p0 + k;
                        const double la = r0[k], lb = r1[k];
                        r0[k] = ld0(p + PF);
                        r1[k] = ld1(p + PF);
#pragma unroll
                        for (int b = 0; b < T; ++b) {
                            const double sp = readlane_f64(v0[b], p);
                            v0[b] = fma(-la, sp, v0[b]);
                            v1[b] = fma(-lb, sp, v1[b]);
                        }
                    }
                }
#pragma unroll 1
                for (int p0 = kWave; p0 < n_h; p0 += PF) {               // pivots owned by bank 1
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        const int p = p0 + k;
                        const double lb = r1[k];
                        r1[k] = ld1(p + PF);
#pragma unroll
                        for (int b = 0; b < T; ++b) v1[b] = fma(-lb, readlane_f64(v1[b], p - kWave), v1[b]);
                    }
                }
            }
            // v = rhs / L_pp  (own rows); partial sums
            {
                int e = 0;
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    v0[b] *= dinv0;
                    v1[b] *= dinv1;
                }
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    pm[b] += v0[b] * wown0 + v1[b] * wown1;
#pragma unroll
                    for (int c = 0; c <= b; ++c) pss[e++] += v0[b] * v0[c] + v1[b] * v1[c];
                }
            }
        }
        FPHASE(3);

        // ---- posterior mean / covariance at the test point ------------------------------------------------------
        double mu[T], S[T][T];
        {
            int e = 0;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                mu[b] = wave_sum(pm[b]);
#pragma unroll
                for (int c = 0; c <= b; ++c) {
                    const double kss = (b == c) ? ((b == 0) ? os : os * il2[b - 1]) : 0.0;
                    const double val = kss - wave_sum(pss[e++]);
                    S[b][c] = val;
                    S[c][b] = val;
                }
            }
        }
        FPHASE(4);
        double var[T];
        bool all_zero = (a.var_zero_thr >= 0.0);
#pragma unroll
        for (int b = 0; b < T; ++b) {
            var[b] = S[b][b];
            if (var[b] < gp.var_floor) {
                var[b] = gp.var_floor;
                info_acc |= GPMPC_INFO_VAR_CLAMPED;
            }
            all_zero = all_zero && (var[b] <= a.var_zero_thr);
        }
        double R[T][T];
        info_acc |= root_small<T>(S, gp.jitter, R);
        const double* zt = a.z + (long)t * a.z_step_stride + (s * G_NY + o) * T;
        double y[T];
#pragma unroll
        for (int b = 0; b < T; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c <= b; ++c) acc += R[b][c] * (valid ? zt[c] : 0.0);
            double yb = acc + mu[b];
            if (all_zero) yb = mu[b];
            const double sd = a.beta * sqrt(var[b]);
            yb = fmax(yb, mu[b] - sd);
            yb = fmin(yb, mu[b] + sd);
            y[b] = yb;
        }
        FPHASE(5);

        // ---- append [v^T, chol(S + noise)] and w to the chain's factor (A.9) -------------------------------------
        if (t + 1 < H) {
            double C[T][T], wn[T], Sn[T][T];
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int c = 0; c < T; ++c) Sn[b][c] = S[b][c] + ((b == c) ? gp.noise[b] : 0.0);
            if (!chol_small<T>(Sn, C)) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
            double cinv[T];
#pragma unroll
            for (int b = 0; b < T; ++b) {
                cinv[b] = 1.0 / C[b][b];
                double acc = y[b] - mu[b];
#pragma unroll
                for (int c = 0; c < b; ++c) acc -= C[b][c] * wn[c];
                wn[b] = acc * cinv[b];
            }
            const int base = n_h;
            // (1) columns of the old slots get T new rows:  L''[base+c][slot] = v_slot[c] / L_slot,slot
            if (lane < n_h) {
                const long co = col_ofs(lane, nh_max) - lane;
#pragma unroll
                for (int c = 0; c < T; ++c) Lhh[co + base + c] = v0[c] * dinv0;
            }
            if (lane + kWave < n_h) {
                const long co = col_ofs(lane + kWave, nh_max) - (lane + kWave);
#pragma unroll
                for (int c = 0; c < T; ++c) Lhh[co + base + c] = v1[c] * dinv1;
            }
            // (2) the new T x T diagonal block, column-scaled
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < T; ++c)
#pragma unroll
                    for (int e = 0; e < T; ++e)
                        if (e < c) Lhh[col_ofs(base + e, nh_max) + (c - e)] = C[c][e] * cinv[e];
            }
            // (3) the lanes that own the new rows: 1/L_pp, w_p and the L_hr row (= v_r^T) into registers
            {
                const int a0 = lane - base;                               // task index if this lane's bank-0 row is new
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
                if (base < kWave) {                                       // uniform: some new row lives in bank 0
                    const int ac = new0 ? a0 : 0;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const double val = vrs[ac * NR + i];
                        Lhr0[i] = new0 ? val : Lhr0[i];
                        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (base + T > kWave) {                                   // uniform: some new row lives in bank 1
                    const int ac = new1 ? a1 : 0;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const double val = vrs[ac * NR + i];
                        Lhr1[i] = new1 ? val : Lhr1[i];
                        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            n_h += T;
        }
        FPHASE(6);

        // ---- state hand-over ---------------------------------------------------------------------------------------
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < T; ++b) yout[t * T + b] = y[b];
        }
        double g[G_NY];
        if (G_NY == 1) {
            g[0] = y[0];
        } else {
            double* yb_t = ybuf + (t & 1) * G_NY;
            if (lane == 0) yb_t[o] = y[0];
            __syncthreads();
#pragma unroll
            for (int oo = 0; oo < G_NY; ++oo) g[oo] = yb_t[oo];
        }
        if (ENV == GPMPC_ENV_PENDULUM1D) {
            const double x0n = x[0] + x[1] * a.env.dt;
            x[1] = x[1] + g[0];
            x[0] = x0n;
        } else {
            const double vv = x[3];
            x[0] = x[0] + vv * g[0];
            x[1] = x[1] + vv * g[G_NY > 1 ? 1 : 0];
            x[2] = x[2] + vv * g[G_NY > 2 ? 2 : 0];
            x[3] = x[3] + u[NU - 1] * a.env.dt;
        }
        FPHASE(7);
    }

    if (lane == 0 && o == 0) {
#pragma unroll
        for (int d = 0; d < NX; ++d) xbuf[d * (H + 1) + H] = x[d];
    }
    if (info_acc && lane == 0) atomicOr(&s_info[sw], info_acc);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (G_NY > 1) __syncthreads();
    if (valid) {
        if (o == 0)
            for (int e = lane; e < NX * (H + 1); e += kWave) a.X_traj[s * NX * (H + 1) + e] = xbuf[e];
        if (a.Y)
            for (int e = lane; e < H * T; e += kWave) a.Y[(s * G_NY + o) * H * T + e] = yout[e];
        if (a.Xi && o == 0)
            for (int e = lane; e < H * D; e += kWave) a.Xi[s * H * D + e] = Xh[e];
        if (G_NY == 1) {
            if (lane == 0) a.info[s] = info_acc;
        } else {
            if (o == 0 && lane == 0) a.info[s] = s_info[sw];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_fast_phase_cycles[i] = ph[i];
#undef FPHASE
}

// ---------------------------------------------------------------------------------------------------------------
// host side: eligibility + launch
// ---------------------------------------------------------------------------------------------------------------
struct FastPlan {
    int spw, waves, lds_shared, lds_per_wave;
    bool lhh_lds;
    size_t lds_bytes;
    long chain_doubles;
};

static bool fast_disabled() {
    const char* e = std::getenv("GPMPC_DISABLE_FAST_ROLLOUT");
    return e && e[0] == '1';
}

bool rollout_fast_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode, int hall_tasks, int H) {
    if (fast_disabled()) return false;
    if (mode != GPMPC_MODE_RECONDITIONED || gp->T != 3 || gp->D != 2 || hall_tasks != 3 || gp->real_has_grad) return false;
    if (!(gp->N_r == 36 || gp->N_r == 45)) return false;
    if (3 * (H - 1) > 128 || H < 2) return false;
    if (env->env_id == GPMPC_ENV_PENDULUM1D) return gp->g_ny == 1 && gp->N_r == 36;
    if (env->env_id == GPMPC_ENV_CAR_RESIDUAL) return gp->g_ny == 3 && gp->N_r == 45;
    return false;
}

static void fast_plan(const gpmpc_gp_desc_t* gp, int nx, int H, bool force_global, FastPlan* fp) {
    const int NR = gp->N_r, T = 3, D = 2, G = gp->g_ny;
    const int nh_max = 3 * (H - 1);
    fp->chain_doubles = ((long)nh_max * (nh_max + 1)) / 2;
    const int vec = 2 * T * NR + H * D + nx * (H + 1) + H * T;
    const size_t budget = 160 * 1024 - 512;
    int max_spw = (G == 1) ? 4 : 1;
    fp->lhh_lds = false;
    fp->spw = max_spw;
    for (int spw = max_spw; spw >= 1 && !force_global; --spw) {
        const int shared = G * NR * NR + G * NR + spw * 2 * G;
        const long per = vec + fp->chain_doubles;
        const size_t bytes = ((size_t)((shared + 1) & ~1) + (size_t)spw * G * ((per + 1) & ~1L)) * sizeof(double);
        if (bytes <= budget) {
            fp->lhh_lds = true;
            fp->spw = spw;
            break;
        }
    }
    fp->waves = fp->spw * G;
    const int shared = G * NR * NR + G * NR + fp->spw * 2 * G;
    fp->lds_shared = (shared + 1) & ~1;
    const long per = vec + (fp->lhh_lds ? fp->chain_doubles : 0);
    fp->lds_per_wave = (int)((per + 1) & ~1L);
    fp->lds_bytes = ((size_t)fp->lds_shared + (size_t)fp->waves * fp->lds_per_wave) * sizeof(double);
}

size_t rollout_fast_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int H) {
    const int nh_max = 3 * (H - 1);
    return (size_t)Ns * gp->g_ny * (((size_t)nh_max * (nh_max + 1)) / 2) * sizeof(double);
}

template <int NR, int G_NY, int ENV>
static int launch_fast(RolloutArgs& args, const FastPlan& fp, hipStream_t st) {
    const long nblk = (args.Ns + fp.spw - 1) / fp.spw;
    const dim3 grid((unsigned)nblk), block(64 * fp.waves);
    if (fp.lhh_lds) {
        auto k = rollout_fast_kernel<3, NR, G_NY, ENV, true>;
        GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes));
        hipLaunchKernelGGL(k, grid, block, fp.lds_bytes, st, args);
    } else {
        auto k = rollout_fast_kernel<3, NR, G_NY, ENV, false>;
        GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes));
        hipLaunchKernelGGL(k, grid, block, fp.lds_bytes, st, args);
    }
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

int rollout_fast_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, void* ws,
                        size_t ws_bytes, hipStream_t st) {
    const char* e = std::getenv("GPMPC_FORCE_GLOBAL_FACTOR");
    const bool force_global = e && e[0] == '1';
    FastPlan fp;
    fast_plan(gp, env->nx, args.H, force_global, &fp);
    args.nh_max = 3 * (args.H - 1);
    args.lds_shared = fp.lds_shared;
    args.lds_per_wave = fp.lds_per_wave;
    args.ws_chain_stride = fp.chain_doubles;
    if (!fp.lhh_lds) {
        if (!ws || ws_bytes < rollout_fast_workspace_bytes(gp, args.Ns, args.H))
            return fail(GPMPC_E_WORKSPACE, "gpmpc_rollout: workspace too small");
    }
    if (env->env_id == GPMPC_ENV_PENDULUM1D) return launch_fast<36, 1, GPMPC_ENV_PENDULUM1D>(args, fp, st);
    return launch_fast<45, 3, GPMPC_ENV_CAR_RESIDUAL>(args, fp, st);
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_fast_phases(long long* out /*[host] 16*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_fast_phase_cycles), 16 * sizeof(long long)));
    return GPMPC_OK;
}
