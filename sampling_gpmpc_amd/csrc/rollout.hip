// gpmpc_rollout: whole-horizon rollout of Ns sampled dynamics functions in ONE launch (gfx950).
//
// Mapping: one workgroup per sample, one 64-lane wave per GP output ("chain").  Each chain keeps the Cholesky
// factor of its own growing training set in append-row form:
//
//      L = [ L_rr   0   ]     L_rr  : shared real-data block, factorised once by gpmpc_plan_build (HBM, L2 resident)
//          [ L_hr  L_hh ]     L_hr  : rows v_r(x_j)^T  of the sample's own previous draws      (n_h x n_r)
//                             L_hh  : lower-triangular Schur factor of those draws            (n_h x n_h)
//
// Per step and chain: kernel row vs real data -> v_r = L_rr^-1 k_r (dense product with the precomputed inverse)
// -> rhs = k_h - L_hr v_r -> v_h = L_hh^-1 rhs (column-oriented substitution, rows in registers, the pivot value
// travels through v_readlane as a scalar operand) -> mu = v^T w, S = k** - v^T v (DPP wave reductions) -> T x T
// root with gpytorch's jitter-on-failure chain -> y = mu + R z -> clip -> append [v^T, chol(S + noise)] and w.
// The per-sample factor lives in LDS when it fits (FAC_LDS) and in an HBM workspace otherwise; both are laid out
// column-major so that lane == row gives conflict-free LDS / coalesced HBM access.
#include "gpmpc_host.hpp"
#include "rollout_args.hpp"

namespace gpmpc {


// Debug only (not part of the public ABI): per-phase shader-cycle totals of block 0 / wave 0, read back through
// gpmpc_debug_read_phases().  Compiled in always; costs one s_memtime per phase.
__device__ long long g_phase_cycles[16];
#define GPMPC_PHASE(idx)                                                  \
    do {                                                                  \
        const long long _now = __builtin_readcyclecounter();              \
        ph[idx] += _now - tph;                                            \
        tph = _now;                                                       \
    } while (0)

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}


template <int T, int RPL, bool FAC_LDS>
__global__ __launch_bounds__(64 * GPMPC_MAX_NY) void rollout_kernel(const RolloutArgs a) {
    constexpr int D = 2;
    constexpr int NS = T * (T + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_info;

    const GpParams& gp = a.gp;
    const EnvParams& env = a.env;
    const long s = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = wave;
    const int H = a.H, nx = env.nx, nu = env.nu;
    const int n_r = gp.n_r, Tr = gp.real_has_grad ? T : 1;
    const int nh_max = a.nh_max, Th = a.hall_tasks;
    const bool recond = (a.mode == GPMPC_MODE_RECONDITIONED);

    double* xbuf = smem;                          // [nx][H+1]
    double* Xh = xbuf + nx * (H + 1);             // [max_points][D]   GP inputs of the seed points, then of every step
    double* ybuf = Xh + a.max_points * D;         // [2][MAX_NY] value samples, double-buffered over t parity
    double* wb = smem + a.lds_shared + (long)wave * a.lds_per_wave;
    double* kr = wb;                              // [T][n_r]
    double* vr = kr + T * n_r;                    // [T][n_r]
    double* wh = vr + T * n_r;                    // [nh_max]
    double* invd = wh + nh_max;                   // [nh_max]
    double* yout = invd + nh_max;                 // [H][T]
    double* LinvT_s = yout + H * T;               // [n_r][n_r] staged copy of the plan's L_rr^-1 (transposed), if it fits
    double* w_r = LinvT_s + (a.linv_in_lds ? n_r * n_r : 0);   // [n_r]
    // factor storage: LDS, the HBM workspace, or - when the caller keeps the factor state - the state buffer itself
    double* st_s = a.state ? a.state + s * a.state_stride : nullptr;
    double* st_c = a.state ? st_s + 4 + (long)a.state_points * D + (long)o * state_chain_doubles(n_r, nh_max) : nullptr;
    double* fac = FAC_LDS ? (w_r + n_r) : (a.state ? st_c : a.ws + (s * gp.g_ny + o) * a.ws_chain_stride);
    double* LhrT = fac;                           // [n_r][nh_max]   LhrT[i*nh_max + slot] = L_hr[slot][i]
    double* Lhh = fac + (long)n_r * nh_max;       // packed lower, column-major: (slot,p) at col_ofs(p)+slot-p

    const double* LinvT = a.linv_in_lds ? LinvT_s : plan_LinvT(a.plan, gp, o);
    {
        const double* gL = plan_LinvT(a.plan, gp, o);
        const double* gw = plan_w(a.plan, gp, o);
        if (a.linv_in_lds)
            for (int e = lane; e < n_r * n_r; e += kWave) LinvT_s[e] = gL[e];
        for (int e = lane; e < n_r; e += kWave) w_r[e] = gw[e];
    }
    double il2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) il2[d] = gp.inv_l2[o][d];
    const double os = gp.os[o];

    if (threadIdx.x == 0) s_info = 0;
    double x[GPMPC_MAX_NX];
    for (int d = 0; d < nx; ++d) x[d] = a.x0[(a.x0_per_sample ? s * nx : 0) + d];
    int info_acc = 0;
    int n_h = 0;
    // Label slots are point-major: the n_seed seed points carry all T tasks, the appended points Th tasks each.
    // (value-only seed points count as appended points: they carry Th tasks like the rollout's own draws)
    int n_seed = a.resume ? 0 : a.n_h0, n_app = 0;                // seed points / appended points in the factor
    if (a.resume) {                                               // continue from the exported factor state
        n_seed = (int)st_s[0];
        n_app = (int)st_s[1];
        n_h = n_seed * T + n_app * Th;
        for (int e = threadIdx.x; e < (n_seed + n_app) * D; e += blockDim.x) Xh[e] = st_s[4 + e];
        const double* sw = st_c + (long)n_r * nh_max + ((long)nh_max * (nh_max + 1)) / 2;
        for (int e = lane; e < n_h; e += kWave) {
            wh[e] = sw[e];
            invd[e] = sw[nh_max + e];
        }
    }
    const int seed_slots = n_seed * T;
    const int pt0 = n_seed + n_app + (a.resume ? 0 : a.n_v0);     // point index of rollout step 0
    const int n_pre = a.resume ? 0 : a.n_h0 + a.n_v0;             // conditioning-only passes before step 0
    __syncthreads();

    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tph = __builtin_readcyclecounter();
    for (int tt = -n_pre; tt < H; ++tt) {
        const bool seeding = tt < 0;                              // uniform: condition on a given point, draw nothing
        const int t = seeding ? 0 : tt;
        const int pt = seeding ? tt + n_pre : pt0 + tt;           // this pass's point
        const bool seed_full = seeding && pt < a.n_h0;            // seed with all T tasks (else: Th tasks)
        double u[GPMPC_MAX_NU], xi[D];
        if (seeding) {
            const double* xs = seed_full ? a.X_h0 + ((s * gp.g_ny + o) * (long)a.n_h0 + pt) * D
                                         : a.X_v0 + ((s * gp.g_ny + o) * (long)a.n_v0 + (pt - a.n_h0)) * D;
            for (int d = 0; d < D; ++d) xi[d] = xs[d];
            for (int i = 0; i < nu; ++i) u[i] = 0.0;
        } else {
            apply_feedback(env, x, a.u_ff + (long)t * nu, u);
            gp_input(env, x, u, xi);
        }
        if (threadIdx.x == 0) {
            if (!seeding)
                for (int d = 0; d < nx; ++d) xbuf[d * (H + 1) + t] = x[d];
            for (int d = 0; d < D; ++d) Xh[pt * D + d] = xi[d];
        }

        // ---- kernel row block against the real data ------------------------------------------------------
        for (int sl = lane; sl < n_r; sl += kWave) {
            const int i = sl / Tr, ar = sl - i * Tr;
            double q[D];
            const double k = kern_scalar<D>(a.X_r + i * D, xi, il2, os, q);
#pragma unroll
            for (int b = 0; b < T; ++b) kr[b * n_r + sl] = kern_entry<D>(q, k, il2, ar, b);
        }
        wave_lds_sync();
        GPMPC_PHASE(0);

        // ---- v_r = L_rr^-1 k_r ; partial sums of mu and v^T v -------------------------------------------
        double pm[T], pss[NS];
#pragma unroll
        for (int b = 0; b < T; ++b) pm[b] = 0.0;
#pragma unroll
        for (int e = 0; e < NS; ++e) pss[e] = 0.0;
        for (int i = lane; i < n_r; i += kWave) {
            double acc[T];
#pragma unroll
            for (int b = 0; b < T; ++b) acc[b] = 0.0;
#pragma unroll 4
            for (int j = 0; j <= i; ++j) {
                const double l = LinvT[j * n_r + i];
#pragma unroll
                for (int b = 0; b < T; ++b) acc[b] += l * kr[b * n_r + j];
            }
            const double wi = w_r[i];
            int e = 0;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                vr[b * n_r + i] = acc[b];
                pm[b] += acc[b] * wi;
#pragma unroll
                for (int c = 0; c <= b; ++c) pss[e++] += acc[b] * acc[c];
            }
        }
        wave_lds_sync();
        GPMPC_PHASE(1);

        // ---- rows of the sample's own previous draws: rhs = k_h - L_hr v_r --------------------------------
        double rhs[RPL][T];
#pragma unroll
        for (int r = 0; r < RPL; ++r)
#pragma unroll
            for (int b = 0; b < T; ++b) rhs[r][b] = 0.0;
        if (n_h > 0) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int slot = lane + kWave * r;
                if (slot < n_h) {
                    int j, ah;                                    // point and task of the slot
                    if (slot < seed_slots) {
                        j = slot / T;
                        ah = slot - j * T;
                    } else {
                        const int rel = slot - seed_slots;
                        j = rel / Th;
                        ah = rel - j * Th;
                        j += n_seed;
                    }
                    double q[D];
                    const double k = kern_scalar<D>(Xh + j * D, xi, il2, os, q);
                    double acc[T];
#pragma unroll
                    for (int b = 0; b < T; ++b) acc[b] = kern_entry<D>(q, k, il2, ah, b);
#pragma unroll 4
                    for (int i = 0; i < n_r; ++i) {
                        const double l = LhrT[(long)i * nh_max + slot];
#pragma unroll
                        for (int b = 0; b < T; ++b) acc[b] -= l * vr[b * n_r + i];
                    }
#pragma unroll
                    for (int b = 0; b < T; ++b) rhs[r][b] = acc[b];
                }
            }

            GPMPC_PHASE(2);
            // ---- v_h = L_hh^-1 rhs: column-oriented forward substitution, next column prefetched ------------
            double lnext[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int slot = lane + kWave * r;
                lnext[r] = (slot > 0 && slot < n_h) ? Lhh[slot] : 0.0;
            }
            for (int p = 0; p < n_h; ++p) {
                double lcur[RPL];
#pragma unroll
                for (int r = 0; r < RPL; ++r) lcur[r] = lnext[r];
                if (p + 1 < n_h) {
                    const long co = col_ofs(p + 1, nh_max) - (p + 1);
#pragma unroll
                    for (int r = 0; r < RPL; ++r) {
                        const int slot = lane + kWave * r;
                        lnext[r] = (slot > p + 1 && slot < n_h) ? Lhh[co + slot] : 0.0;
                    }
                }
                const int owner = p & 63, bank = p >> 6;
                const double dinv = invd[p];
                double vp[T];
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    double val = rhs[0][b];
#pragma unroll
                    for (int r = 1; r < RPL; ++r) val = (bank == r) ? rhs[r][b] : val;
                    vp[b] = readlane_f64(val, owner) * dinv;
                }
#pragma unroll
                for (int r = 0; r < RPL; ++r)
#pragma unroll
                    for (int b = 0; b < T; ++b) rhs[r][b] = fma(-lcur[r], vp[b], rhs[r][b]);
                if (lane == owner) {
#pragma unroll
                    for (int r = 0; r < RPL; ++r)
#pragma unroll
                        for (int b = 0; b < T; ++b) rhs[r][b] = (bank == r) ? vp[b] : rhs[r][b];
                }
            }
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int slot = lane + kWave * r;
                if (slot < n_h) {
                    const double wi = wh[slot];
                    int e = 0;
#pragma unroll
                    for (int b = 0; b < T; ++b) {
                        pm[b] += rhs[r][b] * wi;
#pragma unroll
                        for (int c = 0; c <= b; ++c) pss[e++] += rhs[r][b] * rhs[r][c];
                    }
                }
            }
        }

        GPMPC_PHASE(3);
        // ---- posterior mean / covariance of the T label slots at the test point ---------------------------
        double mu[T], S[T][T];
        {
            int e = 0;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                mu[b] = wave_sum(pm[b]);
#pragma unroll
                for (int c = 0; c <= b; ++c) {
                    const double kss = (b == c) ? ((b == 0) ? os : os * il2[b - 1]) : 0.0;
                    const double v = kss - wave_sum(pss[e++]);
                    S[b][c] = v;
                    S[c][b] = v;
                }
            }
        }
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

        GPMPC_PHASE(4);
        // ---- sample: y = mu + R z, post-processing of sample_gp -------------------------------------------
        double y[T];
        if (seeding) {                                            // the given labels of the seed point
            const double* ys = seed_full ? a.Y_h0 + ((s * gp.g_ny + o) * (long)a.n_h0 + pt) * T
                                         : a.Y_v0 + ((s * gp.g_ny + o) * (long)a.n_v0 + (pt - a.n_h0)) * T;
#pragma unroll
            for (int b = 0; b < T; ++b) y[b] = ys[b];
        } else {
            double R[T][T];
            info_acc |= root_small<T>(S, gp.jitter, R);
            const double* zt = a.z + (long)t * a.z_step_stride + (s * gp.g_ny + o) * T;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c <= b; ++c) acc += R[b][c] * zt[c];
                double yb = acc + mu[b];
                if (all_zero) yb = mu[b];
                const double sd = a.beta * sqrt(var[b]);
                yb = fmax(yb, mu[b] - sd);
                yb = fmin(yb, mu[b] + sd);
                y[b] = yb;
            }
        }

        GPMPC_PHASE(5);
        // ---- append the draw to the chain's own training set (A.9) ----------------------------------------
        // the last step's draw conditions nothing inside this rollout; it is appended only when the factor state is kept
        const int Tc = seed_full ? T : Th;                        // tasks observed at this pass's point
        bool do_append = recond && (seeding || t + 1 < H || a.state);
        // a resumed state without room for this point: label slots, or (kept state) its point list
        if (do_append && (n_h + Tc > nh_max || (a.state && n_seed + n_app + (seed_full ? 0 : 1) > a.state_points))) {
            info_acc |= GPMPC_INFO_STATE_FULL;
            do_append = false;
        }
        if (do_append) {
            double C[T][T], wn[T];
            bool ok = true;
            if (Tc == T) {
                double Sn[T][T];
#pragma unroll
                for (int b = 0; b < T; ++b)
#pragma unroll
                    for (int c = 0; c < T; ++c) Sn[b][c] = S[b][c] + ((b == c) ? gp.noise[b] : 0.0);
                ok = chol_small<T>(Sn, C);
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    double acc = y[b] - mu[b];
#pragma unroll
                    for (int c = 0; c < b; ++c) acc -= C[b][c] * wn[c];
                    wn[b] = acc / C[b][b];
                }
            } else {   // value-only label at the appended point (reference src/agent.py:402)
                const double d0 = S[0][0] + gp.noise[0];
                ok = (d0 > 0.0);
                C[0][0] = sqrt(d0);
                wn[0] = (y[0] - mu[0]) / C[0][0];
            }
            if (!ok) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
            const int base = n_h;
            for (int i = lane; i < n_r; i += kWave)
                for (int c = 0; c < Tc; ++c) LhrT[(long)i * nh_max + base + c] = vr[c * n_r + i];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int slot = lane + kWave * r;
                if (slot < n_h) {
                    const long co = col_ofs(slot, nh_max) - slot;
#pragma unroll
                    for (int c = 0; c < T; ++c)
                        if (c < Tc) Lhh[co + base + c] = rhs[r][c];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    if (c < Tc) {
#pragma unroll
                        for (int e = 0; e < T; ++e)
                            if (e <= c) Lhh[col_ofs(base + e, nh_max) + (c - e)] = C[c][e];
                        invd[base + c] = 1.0 / C[c][c];
                        wh[base + c] = wn[c];
                    }
                }
            }
            n_h += Tc;
            if (!seed_full) ++n_app;
        }
        if (seeding) {                                            // the appended rows are read by other lanes next pass
            __syncthreads();
            continue;
        }

        GPMPC_PHASE(6);
        // ---- hand the value samples of all outputs to every chain, advance the state ----------------------
        double* yb_t = ybuf + (t & 1) * GPMPC_MAX_NY;
        if (lane == 0) {
            yb_t[o] = y[0];
#pragma unroll
            for (int b = 0; b < T; ++b) yout[t * T + b] = y[b];
        }
        __syncthreads();
        double g[GPMPC_MAX_NY], xn[GPMPC_MAX_NX];
        for (int oo = 0; oo < gp.g_ny; ++oo) g[oo] = yb_t[oo];
        env_step(env, x, u, g, xn);
        for (int d = 0; d < nx; ++d) x[d] = xn[d];
        GPMPC_PHASE(7);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_phase_cycles[i] = ph[i];

    if (threadIdx.x == 0)
        for (int d = 0; d < nx; ++d) xbuf[d * (H + 1) + H] = x[d];
    if (info_acc) atomicOr(&s_info, info_acc);
    __syncthreads();
    for (int e = threadIdx.x; e < nx * (H + 1); e += blockDim.x) a.X_traj[s * nx * (H + 1) + e] = xbuf[e];
    if (a.Y)
        for (int e = lane; e < H * T; e += kWave) a.Y[(s * gp.g_ny + o) * H * T + e] = yout[e];
    if (a.Xi)
        for (int e = threadIdx.x; e < H * D; e += blockDim.x) a.Xi[s * H * D + e] = Xh[pt0 * D + e];
    if (a.state) {                                                // export: counts, points, w and 1/diag (the factor is already there)
        if (threadIdx.x == 0) {
            st_s[0] = n_seed;
            st_s[1] = n_app;
            st_s[2] = Th;
            st_s[3] = nh_max;
        }
        for (int e = threadIdx.x; e < min(n_seed + n_app, a.state_points) * D; e += blockDim.x) st_s[4 + e] = Xh[e];
        double* sw = st_c + (long)n_r * nh_max + ((long)nh_max * (nh_max + 1)) / 2;
        for (int e = lane; e < n_h; e += kWave) {
            sw[e] = wh[e];
            sw[nh_max + e] = invd[e];
        }
    }
    if (threadIdx.x == 0) a.info[s] = s_info;
}

struct RolloutPlan {
    int nh_max, rpl, lds_shared, lds_per_wave, linv_in_lds, max_points;
    long chain_doubles;
    bool fac_lds;
    size_t lds_bytes;
};

static bool force_global_factor() {
    const char* e = std::getenv("GPMPC_FORCE_GLOBAL_FACTOR");
    return e && e[0] == '1';
}

static int plan_rollout(const gpmpc_gp_desc_t* gp, int nx, int mode, int hall_tasks, int H, RolloutPlan* rp,
                        int n_h0 = 0, int n_v0 = 0, int state_slots = 0, int state_points = 0) {
    const int n_r = observed_real_slots(gp);
    const int T = gp->T;
    int nh_max = (mode == GPMPC_MODE_RECONDITIONED) ? n_h0 * T + hall_tasks * (n_v0 + H - 1) : 0;
    if (state_slots > 0) nh_max = state_slots;                    // the exported factor's leading dimension
    if (nh_max < 1) nh_max = 1;
    // LDS point list: with a kept / resumed state the points already in it are unknown on the host (up to state_points),
    // and every step of this call records its GP input whether or not the factor still has room for it
    rp->max_points = (state_slots > 0) ? state_points + H : n_h0 + n_v0 + H;
    rp->nh_max = nh_max;
    rp->rpl = (nh_max + 63) / 64;
    if (rp->rpl > 4) return fail(GPMPC_E_UNSUPPORTED, "rollout: more than 256 hallucinated label slots per chain");
    if (rp->rpl == 3) rp->rpl = 4;
    rp->chain_doubles = (long)n_r * nh_max + ((long)nh_max * (nh_max + 1)) / 2;
    rp->lds_shared = nx * (H + 1) + rp->max_points * gp->D + 2 * GPMPC_MAX_NY;
    rp->lds_shared = (rp->lds_shared + 1) & ~1;
    // stage L_rr^-1 per wave when it leaves room for the rest (n_r <= ~60); otherwise it is read through L2
    rp->linv_in_lds = ((size_t)gp->g_ny * n_r * n_r * sizeof(double) <= 64 * 1024) ? 1 : 0;
    const int vec = 2 * T * n_r + 2 * nh_max + H * T + (rp->linv_in_lds ? n_r * n_r : 0) + n_r;
    const long with_fac = vec + rp->chain_doubles;
    const size_t bytes_fac = ((size_t)rp->lds_shared + (size_t)gp->g_ny * ((with_fac + 1) & ~1L)) * sizeof(double);
    rp->fac_lds = (mode == GPMPC_MODE_RECONDITIONED) && bytes_fac <= (size_t)(160 * 1024 - 256) && !force_global_factor() &&
                  state_slots == 0;                               // a kept factor state lives in the caller's buffer
    rp->lds_per_wave = (int)(((rp->fac_lds ? with_fac : (long)vec) + 1) & ~1L);
    rp->lds_bytes = ((size_t)rp->lds_shared + (size_t)gp->g_ny * rp->lds_per_wave) * sizeof(double);
    if (rp->lds_bytes > (size_t)(160 * 1024 - 256)) return fail(GPMPC_E_UNSUPPORTED, "rollout: horizon too long for LDS vectors");
    return GPMPC_OK;
}

template <int T, int RPL>
static int launch_rollout(const RolloutArgs& args, const RolloutPlan& rp, int g_ny, hipStream_t stream) {
    const dim3 grid((unsigned)args.Ns), block(64 * g_ny);
    if (rp.fac_lds) {
        auto k = rollout_kernel<T, RPL, true>;
        GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rp.lds_bytes));
        hipLaunchKernelGGL(k, grid, block, rp.lds_bytes, stream, args);
    } else {
        auto k = rollout_kernel<T, RPL, false>;
        GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rp.lds_bytes));
        hipLaunchKernelGGL(k, grid, block, rp.lds_bytes, stream, args);
    }
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

int g_rollout_pin = GPMPC_KERNEL_AUTO;

}  // namespace gpmpc

using namespace gpmpc;

extern "C" {

// debug helpers, deliberately not declared in include/gpmpc_hip.h
static int g_last_rollout_path = -1;     // 0 generic, 1 tuned re-conditioned (rollout_fast), 2 thread-per-sample (rollout_indep), 3 tiled (rollout_tiles)
int gpmpc_debug_last_rollout_path(void) { return g_last_rollout_path; }
int gpmpc_rollout_last_kernel(void) { return g_last_rollout_path; }
// the kernel an (unseeded) launch of this shape and size takes under the current pin / environment
static int select_rollout_kernel(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode, int hall_tasks, int H, int64_t Ns) {
    if (rollout_one_eligible(gp, env, mode, hall_tasks, H, Ns)) return GPMPC_KERNEL_ONE;
    if (rollout_tiles_eligible(gp, env, mode, hall_tasks, H, Ns)) return GPMPC_KERNEL_TILES;
    if (rollout_fast_eligible(gp, env, mode, hall_tasks, H)) return GPMPC_KERNEL_FAST;
    if (rollout_indep_eligible(gp, env, mode)) return GPMPC_KERNEL_INDEP;
    return GPMPC_KERNEL_GENERIC;
}
int gpmpc_rollout_kernel_for(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int32_t mode, int32_t hall_tasks, int64_t Ns,
                             int32_t H) {
    if (check_gp(gp) != GPMPC_OK || check_env(gp, env) != GPMPC_OK) return GPMPC_KERNEL_AUTO;
    if (mode == GPMPC_MODE_INDEPENDENT) hall_tasks = gp->T;
    return select_rollout_kernel(gp, env, mode, hall_tasks, H, Ns);
}
int gpmpc_rollout_pin_kernel(int32_t kernel) {
    const int prev = gpmpc::g_rollout_pin;
    gpmpc::g_rollout_pin = (kernel >= GPMPC_KERNEL_GENERIC && kernel <= GPMPC_KERNEL_ONE) ? kernel : GPMPC_KERNEL_AUTO;
    return prev;
}

int gpmpc_debug_read_phases(long long* out /*[host] 16*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(long long)));
    return GPMPC_OK;
}

size_t gpmpc_rollout_workspace_bytes(const gpmpc_gp_desc_t* gp, int32_t mode, int32_t hall_tasks, int64_t Ns,
                                     int32_t H) {
    if (check_gp(gp) != GPMPC_OK) return 0;
    RolloutPlan rp;
    if (plan_rollout(gp, GPMPC_MAX_NX, mode, hall_tasks, H, &rp) != GPMPC_OK) return 0;
    if (mode != GPMPC_MODE_RECONDITIONED) return 256;
    size_t need = align_up((size_t)Ns * gp->g_ny * rp.chain_doubles * sizeof(double), 256) + 2048;   // >= the tuned path's need (its zero page included)
    // the tiled throughput kernel keeps the whole tile matrix of a wave's four chains in the workspace (whichever env /
    // size it ends up serving: the query does not know the env, so the larger of the two layouts is reported)
    if (gp->T == 3 && (hall_tasks == 3 || hall_tasks == 1) && 3 * (H - 1) <= 192 && H >= 2) {
        const size_t tl = rollout_tiles_workspace_bytes(gp, Ns, H) + 256;
        need = tl > need ? tl : need;
    }
    return need;
}

size_t gpmpc_rollout_seeded_workspace_bytes(const gpmpc_gp_desc_t* gp, int32_t mode, int32_t hall_tasks, int64_t Ns,
                                            int32_t H, int32_t n_h0, int32_t n_v0) {
    if (check_gp(gp) != GPMPC_OK || n_h0 < 0 || n_v0 < 0) return 0;
    if (mode != GPMPC_MODE_RECONDITIONED) return 256;
    RolloutPlan rp;
    if (plan_rollout(gp, GPMPC_MAX_NX, mode, hall_tasks, H, &rp, n_h0, n_v0) != GPMPC_OK) return 0;
    size_t seeded = align_up((size_t)Ns * gp->g_ny * rp.chain_doubles * sizeof(double), 256) + 2048;
    // the tiled kernel, should the call be routed there (three row slots per point, value-only points included)
    if (gp->T == 3 && 3 * (n_h0 + n_v0 + H - 1) <= 192) {
        const size_t tl = rollout_tiles_workspace_bytes(gp, Ns, H, n_h0 + n_v0) + 256;
        seeded = tl > seeded ? tl : seeded;
    }
    const size_t plain = gpmpc_rollout_workspace_bytes(gp, mode, hall_tasks, Ns, H);
    return seeded > plain ? seeded : plain;
}

static int rollout_impl(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, const void* plan, const double* X_r,
                        int32_t mode, int32_t hall_tasks, double var_zero_thr, double beta, int64_t Ns, int32_t H,
                        const double* x0, int32_t x0_per_sample, const double* u_ff, const double* z,
                        int64_t z_step_stride, double* X_traj, double* Y, double* Xi, int32_t* info, void* ws,
                        size_t ws_bytes, void* stream, const double* X_h0, const double* Y_h0, int32_t n_h0,
                        const double* X_v0, const double* Y_v0, int32_t n_v0, void* state,
                        int32_t state_slots, int32_t state_points, int32_t resume) {
    if (int rc = check_gp(gp)) return rc;
    if (int rc = check_env(gp, env)) return rc;
    if (!plan || !X_r || !x0 || !u_ff || !z || !X_traj || !info) return fail(GPMPC_E_ARG, "gpmpc_rollout: NULL pointer");
    if (Ns < 1 || H < 1) return fail(GPMPC_E_ARG, "gpmpc_rollout: Ns and H must be >= 1");
    if (mode != GPMPC_MODE_INDEPENDENT && mode != GPMPC_MODE_RECONDITIONED) return fail(GPMPC_E_ARG, "bad mode");
    if (mode == GPMPC_MODE_RECONDITIONED && !(hall_tasks == gp->T || hall_tasks == 1))
        return fail(GPMPC_E_ARG, "hall_tasks must be T or 1");
    if (mode == GPMPC_MODE_INDEPENDENT) hall_tasks = gp->T;
    const bool seeded = (n_h0 > 0) || (n_v0 > 0) || state != nullptr;
    if (n_h0 < 0 || (n_h0 > 0 && (!X_h0 || !Y_h0))) return fail(GPMPC_E_ARG, "gpmpc_rollout_seeded: seed points missing");
    if (n_v0 < 0 || (n_v0 > 0 && (!X_v0 || !Y_v0))) return fail(GPMPC_E_ARG, "gpmpc_rollout_seeded: value-only seed points missing");
    if (seeded && mode != GPMPC_MODE_RECONDITIONED) return fail(GPMPC_E_ARG, "seed points / factor state need GPMPC_MODE_RECONDITIONED");
    if (resume && (!state || n_h0 > 0 || n_v0 > 0)) return fail(GPMPC_E_ARG, "resume needs the factor state and no seed points");
    if (state) {
        // capacity: the seeds, what the state may already hold (unknown here when resuming: the caller sized it), H new points
        if (state_slots < n_h0 * gp->T + hall_tasks * (n_v0 + H) || state_points < n_h0 + n_v0 + H)
            return fail(GPMPC_E_ARG, "gpmpc_rollout_seeded: factor state too small for the seeds plus H appended points");
    }
    RolloutPlan rp;
    if (int rc = plan_rollout(gp, env->nx, mode, hall_tasks, H, &rp, n_h0, n_v0, state ? state_slots : 0, state_points)) return rc;

    RolloutArgs args;
    args.gp = make_gp_params(gp);
    args.env = make_env_params(env);
    args.plan = (const double*)plan;
    args.X_r = X_r;
    args.mode = mode;
    args.hall_tasks = hall_tasks;
    args.var_zero_thr = var_zero_thr;
    args.beta = beta;
    args.Ns = Ns;
    args.H = H;
    args.x0 = x0;
    args.x0_per_sample = x0_per_sample;
    args.u_ff = u_ff;
    args.z = z;
    args.z_step_stride = z_step_stride;
    args.X_traj = X_traj;
    args.Y = Y;
    args.Xi = Xi;
    args.info = (int*)info;
    args.ws = (double*)ws;
    args.ws_chain_stride = rp.chain_doubles;
    args.nh_max = rp.nh_max;
    args.lds_shared = rp.lds_shared;
    args.lds_per_wave = rp.lds_per_wave;
    args.linv_in_lds = rp.linv_in_lds;
    args.X_h0 = X_h0;
    args.Y_h0 = Y_h0;
    args.n_h0 = n_h0;
    args.X_v0 = X_v0;
    args.Y_v0 = Y_v0;
    args.n_v0 = n_v0;
    args.state = (double*)state;
    args.state_points = state_points;
    args.state_stride = state ? state_sample_doubles(gp->g_ny, observed_real_slots(gp), state_slots, state_points, gp->D) : 0;
    args.resume = resume;
    args.max_points = rp.max_points;
    hipStream_t st = (hipStream_t)stream;
    // seed points without a kept factor state are conditioning-only passes of the tiled kernel's step body; a kept / resumed
    // state is the generic kernel's own factor layout
    int kernel = GPMPC_KERNEL_GENERIC;
    if (!seeded) kernel = select_rollout_kernel(gp, env, mode, hall_tasks, H, Ns);
    else if (!state && rollout_tiles_eligible(gp, env, mode, hall_tasks, H, Ns, n_h0, n_v0)) kernel = GPMPC_KERNEL_TILES;
    // the generic kernel's factor lives in the workspace (the tiled / fast kernels check their own needs in their launchers; the
    // one-chain-per-wave MFMA kernel keeps the factor in registers and needs none)
    if (kernel == GPMPC_KERNEL_GENERIC && mode == GPMPC_MODE_RECONDITIONED && !rp.fac_lds && !state) {
        const size_t need = (size_t)Ns * gp->g_ny * rp.chain_doubles * sizeof(double);
        if (!ws || ws_bytes < need) return fail(GPMPC_E_WORKSPACE, "gpmpc_rollout: workspace too small");
    }
    g_last_rollout_path = kernel;
    if (kernel == GPMPC_KERNEL_ONE) return rollout_one_launch(gp, env, args, st);
    if (kernel == GPMPC_KERNEL_TILES) return rollout_tiles_launch(gp, env, args, ws, ws_bytes, st);
    if (kernel == GPMPC_KERNEL_FAST) return rollout_fast_launch(gp, env, args, ws, ws_bytes, st);
    if (kernel == GPMPC_KERNEL_INDEP) return rollout_indep_launch(gp, env, args, st);
    const int T = gp->T;
    if (T == 1) {
        if (rp.rpl == 1) return launch_rollout<1, 1>(args, rp, gp->g_ny, st);
        if (rp.rpl == 2) return launch_rollout<1, 2>(args, rp, gp->g_ny, st);
        return launch_rollout<1, 4>(args, rp, gp->g_ny, st);
    } else if (T == 3) {
        if (rp.rpl == 1) return launch_rollout<3, 1>(args, rp, gp->g_ny, st);
        if (rp.rpl == 2) return launch_rollout<3, 2>(args, rp, gp->g_ny, st);
        return launch_rollout<3, 4>(args, rp, gp->g_ny, st);
    }
    return fail(GPMPC_E_UNSUPPORTED, "rollout: only T = 1 and T = 3 (D = 2) are instantiated");
}

int gpmpc_rollout(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, const void* plan, const double* X_r,
                  int32_t mode, int32_t hall_tasks, double var_zero_thr, double beta, int64_t Ns, int32_t H,
                  const double* x0, int32_t x0_per_sample, const double* u_ff, const double* z,
                  int64_t z_step_stride, double* X_traj, double* Y, double* Xi, int32_t* info, void* ws,
                  size_t ws_bytes, void* stream) {
    return rollout_impl(gp, env, plan, X_r, mode, hall_tasks, var_zero_thr, beta, Ns, H, x0, x0_per_sample, u_ff, z,
                        z_step_stride, X_traj, Y, Xi, info, ws, ws_bytes, stream, nullptr, nullptr, 0, nullptr, nullptr, 0,
                        nullptr, 0, 0, 0);
}

size_t gpmpc_rollout_state_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int32_t state_slots, int32_t state_points) {
    if (check_gp(gp) != GPMPC_OK || state_slots < 1 || state_slots > 256 || state_points < 1) return 0;
    return align_up((size_t)Ns * state_sample_doubles(gp->g_ny, observed_real_slots(gp), state_slots, state_points, gp->D) *
                        sizeof(double), 256);
}

int gpmpc_rollout_seeded(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, const void* plan, const double* X_r,
                         int32_t mode, int32_t hall_tasks, double var_zero_thr, double beta, int64_t Ns, int32_t H,
                         const double* x0, int32_t x0_per_sample, const double* u_ff, const double* z,
                         int64_t z_step_stride, double* X_traj, double* Y, double* Xi, int32_t* info, void* ws,
                         size_t ws_bytes, void* stream, const double* X_h0, const double* Y_h0, int32_t n_h0,
                         const double* X_v0, const double* Y_v0, int32_t n_v0, void* state,
                         int32_t state_slots, int32_t state_points, int32_t resume) {
    return rollout_impl(gp, env, plan, X_r, mode, hall_tasks, var_zero_thr, beta, Ns, H, x0, x0_per_sample, u_ff, z,
                        z_step_stride, X_traj, Y, Xi, info, ws, ws_bytes, stream, X_h0, Y_h0, n_h0, X_v0, Y_v0, n_v0, state,
                        state_slots, state_points, resume);
}

}  // extern "C"
