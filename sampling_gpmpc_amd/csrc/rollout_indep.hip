// rollout_indep_kernel: mode "I" rollout (every step conditions on the shared real data only; the reference's
// simulate_forward_sampling_car.py as shipped, use_model_without_derivatives: True, T = 1).  gfx950, wave64.
//
// The factor is identical for every sample, so the natural mapping is ONE SAMPLE PER LANE:
//   * v = L_rr^-1 k_r is column-oriented: for each real point j the lane forms k_j and updates its NR private
//     accumulators  acc_i += L_rr^-1[i][j] * k_j  (i >= j): NR independent FMA chains per lane (no dependency stalls).
//     The matrix entry is uniform across lanes: the packed lower triangle of all outputs (24.8 KB for the car) is
//     staged once in LDS and read as broadcast ds_read_b128 (two entries per LDS instruction).  (Scalar loads were
//     measured 4x slower: three 16 KB matrices thrash the 16 KB scalar cache.)
//   * the training inputs form a tensor-product grid (gpmpc_gp_desc_t::grid_n0/n1), so the RBF row is separable:
//     k_j = os * E0[a] * E1[c] with N0 + N1 exponentials per output instead of N0*N1.
//   * mu = acc . w_r, S = os - acc . acc, y = mu + sqrt(S) z (1x1 root: plain sqrt, App. A.7), floor, clip, env step.
// Algorithmic work per trajectory-step (car): 3 * (1035 + 90) FMA + 42 exp; 112 B of unavoidable HBM traffic.
#include "gpmpc_host.hpp"
#include "rollout_args.hpp"

namespace gpmpc {

typedef double double2_i __attribute__((ext_vector_type(2)));

// packed lower triangle, column-major, every column start 16-byte aligned: column j holds rows j..NR-1
template <int NR>
__host__ __device__ constexpr int tri_col_ofs(int j) {
    int o = 0;
    for (int k = 0; k < j; ++k) o += ((NR - k) + 1) & ~1;
    return o;
}

template <int ENV, int N0, int N1, int G_NY>
__global__ __launch_bounds__(256) void rollout_indep_kernel(const RolloutArgs a) {
    constexpr int NR = N0 * N1;
    constexpr int TRI = tri_col_ofs<NR>(NR);                      // doubles per output
    __shared__ __attribute__((aligned(16))) double Ltri[G_NY * TRI];
    __shared__ double wr_s[G_NY * NR];
    constexpr int NX = (ENV == GPMPC_ENV_PENDULUM1D) ? 2 : 4;
    constexpr int NU = (ENV == GPMPC_ENV_PENDULUM1D) ? 1 : 2;
    const GpParams& gp = a.gp;
    for (int e = threadIdx.x; e < G_NY * NR * NR; e += blockDim.x) {
        const int o = e / (NR * NR), rem = e - o * NR * NR, j = rem / NR, i = rem - j * NR;
        if (i >= j) Ltri[o * TRI + tri_col_ofs<NR>(j) + (i - j)] = a.plan[o * a.gp.plan_stride + (long)NR * NR + j * NR + i];
    }
    for (int e = threadIdx.x; e < G_NY * NR; e += blockDim.x) {
        const int o = e / NR;
        wr_s[e] = a.plan[o * a.gp.plan_stride + 2L * NR * NR + (e - o * NR)];
    }
    __syncthreads();
    const long sraw = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = sraw < a.Ns;
    const long s = active ? sraw : a.Ns - 1;
    const int H = a.H;
    const double* __restrict__ Xr = a.X_r;

    double x[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) x[d] = a.x0[(a.x0_per_sample ? s * NX : 0) + d];
    int info_acc = 0;

#pragma unroll 1
    for (int t = 0; t < H; ++t) {
        double u[NU], xi[2];
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
        if (active) {
#pragma unroll
            for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + t] = x[d];
            if (a.Xi) {
                a.Xi[(s * H + t) * 2 + 0] = xi[0];
                a.Xi[(s * H + t) * 2 + 1] = xi[1];
            }
        }
        double g[G_NY];
#pragma unroll
        for (int o = 0; o < G_NY; ++o) {
            const double il0 = gp.inv_l2[o][0], il1 = gp.inv_l2[o][1], os = gp.os[o];
            double E0[N0], E1[N1];
#pragma unroll
            for (int q = 0; q < N0; ++q) {
                const double r = Xr[(q * N1) * 2 + 0] - xi[0];
                E0[q] = os * exp(-0.5 * r * r * il0);
            }
#pragma unroll
            for (int c = 0; c < N1; ++c) {
                const double r = Xr[c * 2 + 1] - xi[1];
                E1[c] = exp(-0.5 * r * r * il1);
            }
            const double* LT = Ltri + o * TRI;
            const double* wr = wr_s + o * NR;
            double acc[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) acc[i] = 0.0;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const double kj = E0[j / N1] * E1[j % N1];
                const double* col = LT + tri_col_ofs<NR>(j);          // rows j.. of column j, 16-byte aligned
#pragma unroll
                for (int i = j; i + 1 < NR; i += 2) {
                    const double2_i l = *reinterpret_cast<const double2_i*>(col + (i - j));
                    acc[i] = fma(l.x, kj, acc[i]);
                    acc[i + 1] = fma(l.y, kj, acc[i + 1]);
                }
                if ((NR - j) & 1) acc[NR - 1] = fma(col[NR - 1 - j], kj, acc[NR - 1]);
                // keep at most one column of LDS loads in flight: without this fence the scheduler hoists hundreds
                // of ds_read_b128 ahead of their FMAs and spills the accumulators
                asm volatile("" ::: "memory");
            }
            double mu = 0.0, ss = 0.0;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                mu = fma(acc[i], wr[i], mu);
                ss = fma(acc[i], acc[i], ss);
            }
            const double S = os - ss;
            double var = S;
            if (var < gp.var_floor) {
                var = gp.var_floor;
                info_acc |= GPMPC_INFO_VAR_CLAMPED;
            }
            if (S < 0.0) info_acc |= GPMPC_INFO_NEG_1x1;
            const double z = a.z[(long)t * a.z_step_stride + (s * G_NY + o)];
            double y = fma(sqrt(S), z, mu);
            if (a.var_zero_thr >= 0.0 && var <= a.var_zero_thr) y = mu;
            const double sd = a.beta * sqrt(var);
            y = fmin(fmax(y, mu - sd), mu + sd);
            g[o] = y;
            if (active && a.Y) a.Y[(s * G_NY + o) * H + t] = y;
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
    }
    if (active) {
#pragma unroll
        for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + H] = x[d];
        a.info[s] = info_acc;
    }
}

bool rollout_indep_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode) {
    const char* e = std::getenv("GPMPC_DISABLE_FAST_ROLLOUT");
    if (e && e[0] == '1') return false;
    if (mode != GPMPC_MODE_INDEPENDENT || gp->T != 1 || gp->D != 2 || gp->real_has_grad) return false;
    if (env->env_id == GPMPC_ENV_CAR_RESIDUAL) return gp->g_ny == 3 && gp->grid_n0 == 5 && gp->grid_n1 == 9;
    if (env->env_id == GPMPC_ENV_PENDULUM1D) return gp->g_ny == 1 && gp->grid_n0 == 4 && gp->grid_n1 == 9;
    return false;
}

int rollout_indep_launch(const gpmpc_env_desc_t* env, RolloutArgs& args, hipStream_t st) {
    const dim3 grid((unsigned)((args.Ns + 255) / 256)), block(256);
    if (env->env_id == GPMPC_ENV_CAR_RESIDUAL)
        hipLaunchKernelGGL((rollout_indep_kernel<GPMPC_ENV_CAR_RESIDUAL, 5, 9, 3>), grid, block, 0, st, args);
    else
        hipLaunchKernelGGL((rollout_indep_kernel<GPMPC_ENV_PENDULUM1D, 4, 9, 1>), grid, block, 0, st, args);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // namespace gpmpc
