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

#include <type_traits>
#include <utility>

namespace gpmpc {

__device__ long long g_indep_phase_cycles[16];
#ifdef GPMPC_PHASE_TIMERS
#define IPHASE_DECL                                   \
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};       \
    long long tph = __builtin_readcyclecounter()
#define IPHASE(idx)                                              \
    do {                                                         \
        const long long _n = __builtin_readcyclecounter();       \
        ph[idx] += _n - tph;                                     \
        tph = _n;                                                \
    } while (0)
#define IPHASE_STORE                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0)                     \
        for (int i = 0; i < 8; ++i) g_indep_phase_cycles[i] = ph[i]
#else
#define IPHASE_DECL
#define IPHASE(idx)
#define IPHASE_STORE
#endif

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

// ---------------------------------------------------------------------------------------------------------------
// rollout_indep_grid_kernel: the same rollout with the GRID ROOT of the real block (plan, gpmpc_device.hpp).
// The real inputs are the tensor grid meshgrid(axis0[N0], axis1[N1]) with value-only labels, so
//   (K_rr + s2 I)^-1 = W^T W,  W = D^-1/2 (Qa (x) Qb)^T,  k_r = os (ea (x) eb)   =>   W k_r = dsc . (A (x) B),
//   A = Qa^T ea (N0 x N0),  B = Qb^T eb (N1 x N1),
//   mu = sum_c B_c sum_a m1[a][c] A_a,      k_r^T (K_rr + s2 I)^-1 k_r = sum_c B_c^2 sum_a m2[a][c] A_a^2
// with m1 = dsc . wE and m2 = dsc^2 from the plan: N0^2 + N1^2 + 2 N0 N1 + 2 N1 = 214 FMA per output and step (car)
// instead of the 1035 + 90 of the triangular product above - the posterior mean and variance only see W^T W, so the
// two forms agree to round-off (tests/test_hip_parity.py compares both with the oracle).
// Mapping: one SAMPLE per lane, one OUTPUT per wave (workgroup = G_NY waves = 64 samples; the outputs of a step meet
// in a double-buffered LDS exchange, one barrier per step).  Every table entry is uniform across the wave and an
// output's tables are 1.9 KB (plan_tabi_*: 248 doubles), so they are REGISTER RESIDENT: 15 VGPR pairs, each holding 16
// entries replicated in the four DPP rows, loaded once before the step loop, and every FMA reads its entry as a DPP
// row_newbcast operand.  The step loop touches no table memory at all.  (Measured on the way, git history: the tables
// streamed through hand-pipelined s_load_dwordx16 into SGPR operands - scalar loads return out of order, so every wait
// is lgkmcnt(0) and a wave can hide a load only behind the 16 FMAs of the previous group: 15 exposed K$ round trips
// per step, 0.67 ms at Ns = 262144 and 0.114 ms at 32768; LDS broadcast reads are bound by the LDS pipe, 3.0 ms.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kIndepMaxH = 256;                                          // input sequence staged in LDS up to this horizon
// Table access: the table lives in VGPRs, entry f in register f / 16, replicated in every DPP row of 16 lanes (lane l
// holds entry 16 r + (l & 15)).  v_fmac_f64_dpp with row_newbcast:(f % 16) reads it as a
// wave-uniform operand.  The table registers are never written inside the step loop, so the DPP read-after-VALU-write
// hazard (two wait states) cannot arise.
template <int F, int NREG>
__device__ __forceinline__ void fmac_tab(double& acc, const double (&T)[NREG], double x) {                 // acc += tab[F] * x
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(T[F / 16]), "v"(x), "n"(F % 16));
}
// (gfx950 has no DPP form of v_mul_f64 / v_add_f64: products and differences against a table entry go through the fmac)
template <int F, int NREG>
__device__ __forceinline__ double mul_tab(const double (&T)[NREG], double x) {                            // tab[F] * x
    double r = 0.0;
    fmac_tab<F>(r, T, x);
    return r;
}
template <int F, int NREG>
__device__ __forceinline__ double sub_from_tab(const double (&T)[NREG], double x) {                       // tab[F] - x
    double r = -x;
    fmac_tab<F>(r, T, 1.0);
    return r;
}
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// exp(x) for x <= 0 (the arithmetic of expn_neg, gpmpc_device.hpp, as a single chain: with several waves per SIMD the
// hardware interleaves the chains of different waves)
__device__ __forceinline__ double exp_neg1(double x) {
    const double log2e = bits_f64(0x3FF71547652B82FEull), nln2h = bits_f64(0xBFE62E42FEFA39EFull),
                 nln2l = bits_f64(0xBC7ABC9E3B39803Full);
    const double c2 = bits_f64(0x3FE000000000000Bull), c3 = bits_f64(0x3FC5555555555511ull), c4 = bits_f64(0x3FA55555555502A1ull),
                 c5 = bits_f64(0x3F81111111122322ull), c6 = bits_f64(0x3F56C16C1852B7B0ull), c7 = bits_f64(0x3F2A01A014761F6Eull),
                 c8 = bits_f64(0x3EFA01997C89E6B0ull), c9 = bits_f64(0x3EC71DEE623FDE64ull), c10 = bits_f64(0x3E928AF3FCA7AB0Cull),
                 c11 = bits_f64(0x3E5ADE156A5DCB37ull);
    const double n = rint(x * log2e);
    double r = fma(n, nln2h, x);
    r = fma(n, nln2l, r);
    const double r2 = r * r, r4 = r2 * r2;
    const double b0 = fma(fma(c3, r, c2), r2, 1.0 + r);
    const double b1 = fma(fma(c7, r, c6), r2, fma(c5, r, c4));
    const double b2 = fma(fma(c11, r, c10), r2, fma(c9, r, c8));
    return ldexp(fma(fma(b2, r4, b1), r4, b0), (int)n);
}

template <int ENV, int N0, int N1, int G_NY>
__global__ __launch_bounds__(64 * G_NY) void rollout_indep_grid_kernel(const RolloutArgs a) {
    constexpr int NX = (ENV == GPMPC_ENV_PENDULUM1D) ? 2 : 4;
    constexpr int NU = (ENV == GPMPC_ENV_PENDULUM1D) ? 1 : 2;
    constexpr int QA = plan_tabi_qa(N0, N1), QB = plan_tabi_qb(N0, N1), M1 = plan_tabi_m1(N0, N1), M2 = plan_tabi_m2(N0, N1);
    static_assert(N0 + N1 <= 14 && QA == 32, "table groups 0 / 1 hold the recurrence constants / the axis points");
    __shared__ double ybuf[2][G_NY][kWave];
    __shared__ __attribute__((aligned(16))) double xtile[8][kWave * NX + 8];   // eight steps of the workgroup's trajectories
    __shared__ double uffs[kIndepMaxH * NU];                              // the input sequence
    __shared__ __attribute__((aligned(16))) double fbk[NX + NU * NX];     // x_goal, K (row-major)
    __shared__ int s_info[kWave];
    const GpParams& gp = a.gp;
    const int lane = threadIdx.x & 63;
    const int o = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // this wave's output
    // X_traj is (Ns, nx, H+1): a lane storing its state every step writes 8 bytes into a different cache line per
    // lane and step (the lines are evicted long before the next step fills them: ~8x write amplification).  The states
    // of eight steps are staged in LDS instead (xtile[step][sample * NX + d], step stride padded for the transposed read)
    // and written out by the whole workgroup as 64-byte row segments: thread -> (step tt = tid & 7, row tid >> 3 + k * RPI),
    // the row of global sample smp, dimension d being row (smp * NX + d) of X_traj viewed as (Ns * NX, H + 1).
    auto flush_tile = [&](int t0, int cnt) {
        const int tt = threadIdx.x & 7, RPI = blockDim.x >> 3;
        int r = threadIdx.x >> 3;
        const long rows_left = (a.Ns - (long)blockIdx.x * kWave) * NX;    // rows of this workgroup that exist
        double* dst = a.X_traj + ((long)blockIdx.x * kWave * NX + r) * (a.H + 1) + t0 + tt;
        const long dstep = (long)RPI * (a.H + 1);
        if (tt < cnt) {
            for (; r < kWave * NX; r += RPI, dst += dstep)
                if (r < rows_left) *dst = xtile[tt][r];
        }
    };
    const long sraw = (long)blockIdx.x * kWave + lane;
    const bool active = sraw < a.Ns;
    const long s = active ? sraw : a.Ns - 1;
    const int H = a.H;
    constexpr int NREG = plan_tabi_doubles(N0, N1) / 16;
    const double* tabg = plan_grid_tabi(a.plan, gp, o);
    double T[NREG];                                                       // the output's tables (see fmac_tab)
#pragma unroll
    for (int r = 0; r < NREG; ++r) T[r] = (r == 1) ? 0.0 : tabg[16 * r + (lane & 15)];      // group 1 = axis points: fallback only
    const bool rec_ok = tabg[1] == tabg[1];                               // equispaced axes (plan): recurrence constants valid
    if (rec_ok) {
        // the recurrence yields ea_i = E_0 rho^i G_i: the constant G_i is folded into row i of Qa (Qb likewise) once, here
#pragma unroll
        for (int r = QA / 16; r <= (M1 - 1) / 16; ++r) {
            const int f = 16 * r + (lane & 15);
            double gf = 1.0;
            if (f >= QA && f < QA + N0 * N0) {
                const int i = (f - QA) / N0;
                if (i > 0) gf = tabg[3 + i];
            } else if (f >= QB && f < QB + N1 * N1) {
                const int j = (f - QB) / N1;
                if (j > 0) gf = tabg[3 + (N0 - 1) + j];
            }
            T[r] *= gf;
        }
    }
    const double il0 = gp.inv_l2[o][0], il1 = gp.inv_l2[o][1], os = gp.os[o];
    if (threadIdx.x < kWave) s_info[threadIdx.x] = 0;
    // The input sequence and the feedback law are staged in LDS once: read from the kernel arguments / HBM inside the
    // step loop they are a chain of ~7 dependent scalar-load round trips at the head of every step (pointer, element,
    // x_goal, a row of K, ...), which nothing in a wave's own instruction stream hides.
    const bool uff_lds = H <= kIndepMaxH;                                 // longer horizons read the sequence from memory
    for (int e = threadIdx.x; e < H * NU && uff_lds; e += blockDim.x) uffs[e] = a.u_ff[e];
    if (threadIdx.x < NX) fbk[threadIdx.x] = a.env.x_goal[threadIdx.x];
    if (threadIdx.x < NU * NX) fbk[NX + threadIdx.x] = a.env.K[threadIdx.x / NX][threadIdx.x % NX];
    const bool use_fb = a.env.use_feedback != 0;
    // Base samples: one 8-byte read per lane and step, requested at the head of the step and used at its end.  (A
    // chunked prefetch - eight steps ahead through registers and a per-lane LDS column - was measured and made no
    // difference: the read's latency is already covered by the ~2000 issue cycles between request and use.)
    const double* zp = a.z + (s * G_NY + o);                             // this lane's base sample of step t: zp[t * stride]
    const long zstride = a.z_step_stride;
    __syncthreads();

    double x[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) x[d] = a.x0[(a.x0_per_sample ? s * NX : 0) + d];
    int info_acc = 0;
    const double sqrt_floor = sqrt(gp.var_floor);

    IPHASE_DECL;
#pragma unroll 1
    for (int t = 0; t < H; ++t) {
        double u[NU], xi[2];
        const double z = zp[(long)t * zstride];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = uff_lds ? uffs[t * NU + i] : a.u_ff[t * NU + i];
        if (use_fb) {                                                     // uniform
            double dx[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) dx[j] = fbk[j] - x[j];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) acc += dx[j] * fbk[NX + i * NX + j];
                u[i] = -acc + u[i];
            }
        }
        xi[0] = (ENV == GPMPC_ENV_PENDULUM1D) ? x[0] : x[2];
        xi[1] = u[0];
        if (o == 0) {
#pragma unroll
            for (int d = 0; d < NX; ++d) xtile[t & 7][lane * NX + d] = x[d];
            if (active && a.Xi) {
                a.Xi[(s * H + t) * 2 + 0] = xi[0];
                a.Xi[(s * H + t) * 2 + 1] = xi[1];
            }
        }

        // ---- axis factors ea_q = exp(-il0 (xa_q - xi0)^2 / 2), eb_c likewise ------------------------------------------
        // Equispaced axes (the reference's linspace grids; the plan checks): with r_0 = x_0 - xi, x_k = x_0 + k h,
        //   exp(-il (r_0 + k h)^2 / 2) = E_0 rho^k G_k,   E_0 = exp(-il r_0^2 / 2), rho = exp(-il h r_0), G_k = exp(-il (k h)^2 / 2)
        // i.e. TWO exponentials per axis and 2 (N - 1) + log-depth multiplies instead of N exponentials; G_k comes from
        // the plan.  Range: the plan admits the recurrence only for il ((N-1) h)^2 <= 200, and |il h r_0| (N-1) is clamped
        // to 700 - beyond that every factor of the axis is < 1e-271 and E_0 underflows to exactly 0, so the clamped
        // product is 0 as well.  Otherwise (il0*h0 = NaN in the table) the axis points are fetched and N exponentials run.
        IPHASE(0);
        double ea[N0], eb[N1];
        if (rec_ok) {                                                     // uniform
            // table group 0: x_first(axis 0), il0 h0, x_first(axis 1), il1 h1 (G0_k, G1_k: folded into Qa / Qb)
            const double r0a = sub_from_tab<0>(T, xi[0]), r0b = sub_from_tab<2>(T, xi[1]);
            constexpr double capa = (N0 > 1) ? 700.0 / (N0 - 1) : 700.0, capb = (N1 > 1) ? 700.0 / (N1 - 1) : 700.0;
            const double aa = fmin(fmax(-mul_tab<1>(T, r0a), -capa), capa), ab = fmin(fmax(-mul_tab<3>(T, r0b), -capb), capb);
            const double E0a = exp_neg1(-0.5 * r0a * r0a * il0), E0b = exp_neg1(-0.5 * r0b * r0b * il1);
            double pa[N0], pb[N1];                                        // rho^k, log depth
            pa[0] = 1.0, pb[0] = 1.0;
            if constexpr (N0 > 1) pa[1] = exp_neg1(aa);
            if constexpr (N1 > 1) pb[1] = exp_neg1(ab);
#pragma unroll
            for (int k = 2; k < N0; ++k) pa[k] = pa[k / 2] * pa[k - k / 2];
#pragma unroll
            for (int k = 2; k < N1; ++k) pb[k] = pb[k / 2] * pb[k - k / 2];
            ea[0] = E0a, eb[0] = E0b;
#pragma unroll
            for (int k = 1; k < N0; ++k) ea[k] = E0a * pa[k];             // G_k lives in the table rows (see the prologue)
#pragma unroll
            for (int k = 1; k < N1; ++k) eb[k] = E0b * pb[k];
        } else {
            const double* axg = tabg + plan_tabi_axis(N0, N1);
#pragma unroll
            for (int q = 0; q < N0; ++q) {
                const double r = axg[q] - xi[0];
                ea[q] = exp_neg1(-0.5 * r * r * il0);
            }
#pragma unroll
            for (int c = 0; c < N1; ++c) {
                const double r = axg[N0 + c] - xi[1];
                eb[c] = exp_neg1(-0.5 * r * r * il1);
            }
        }

        IPHASE(1);
        double A[N0], B[N1], tm[N1], tv[N1], A2[N0];
#pragma unroll
        for (int k = 0; k < N0; ++k) A[k] = 0.0;
#pragma unroll
        for (int k = 0; k < N1; ++k) B[k] = 0.0, tm[k] = 0.0, tv[k] = 0.0;
        static_for<N0 * N0>([&](auto ec) {                                // A_k += Qa[i][k] ea_i
            constexpr int e = decltype(ec)::value, i = e / N0, k = e % N0;
            fmac_tab<QA + e>(A[k], T, ea[i]);
        });
        static_for<N1 * N1>([&](auto ec) {                                // B_k += Qb[j][k] eb_j
            constexpr int e = decltype(ec)::value, j = e / N1, k = e % N1;
            fmac_tab<QB + e>(B[k], T, eb[j]);
        });
#pragma unroll
        for (int i = 0; i < N0; ++i) A2[i] = A[i] * A[i];
        static_for<N0 * N1>([&](auto ec) {                                // tm_c += m1[a][c] A_a, tv_c += m2[a][c] A_a^2
            constexpr int e = decltype(ec)::value, i = e / N1, c = e % N1;
            fmac_tab<M1 + e>(tm[c], T, A[i]);
            fmac_tab<M2 + e>(tv[c], T, A2[i]);
        });
        double mu = 0.0, ss = 0.0;
#pragma unroll
        for (int c = 0; c < N1; ++c) {
            mu = fma(tm[c], B[c], mu);
            ss = fma(tv[c], B[c] * B[c], ss);
        }
        IPHASE(2);
        const double S = os - ss;
        double var = S;
        if (var < gp.var_floor) {
            var = gp.var_floor;
            info_acc |= GPMPC_INFO_VAR_CLAMPED;
        }
        if (S < 0.0) info_acc |= GPMPC_INFO_NEG_1x1;
        const double sq = sqrt(S);                                       // S < 0: NaN, as the reference's 1x1 root
        double y = fma(sq, z, mu);
        if (a.var_zero_thr >= 0.0 && var <= a.var_zero_thr) y = mu;
        const double sd = a.beta * ((S >= gp.var_floor) ? sq : sqrt_floor);   // sqrt(var): var = max(S, floor)
        y = fmin(fmax(y, mu - sd), mu + sd);
        if (active && a.Y) a.Y[(s * G_NY + o) * H + t] = y;

        IPHASE(3);
        double g[G_NY];
        if constexpr (G_NY == 1) {
            g[0] = y;
        } else {
            ybuf[t & 1][o][lane] = y;
            __syncthreads();
#pragma unroll
            for (int oo = 0; oo < G_NY; ++oo) g[oo] = ybuf[t & 1][oo][lane];
        }
        IPHASE(4);
        if ((t & 7) == 7) {                                               // uniform
            if constexpr (G_NY == 1) __syncthreads();
            flush_tile(t - 7, 8);
            __syncthreads();                                              // the tile is rewritten from the next step on
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
        IPHASE(5);
    }
    IPHASE_STORE;
    if (info_acc) atomicOr(&s_info[lane], info_acc);
    if (o == 0) {
#pragma unroll
        for (int d = 0; d < NX; ++d) xtile[H & 7][lane * NX + d] = x[d];
    }
    __syncthreads();
    flush_tile(H & ~7, (H & 7) + 1);
    if (active && o == 0) a.info[s] = s_info[lane];
}


bool rollout_indep_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode) {
    if (g_rollout_pin == GPMPC_KERNEL_GENERIC) return false;    // (the other pins name mode-R kernels: mode I is unaffected)
    const char* e = std::getenv("GPMPC_DISABLE_FAST_ROLLOUT");
    if (e && e[0] == '1') return false;
    if (mode != GPMPC_MODE_INDEPENDENT || gp->T != 1 || gp->D != 2 || gp->real_has_grad) return false;
    if (env->env_id == GPMPC_ENV_CAR_RESIDUAL) return gp->g_ny == 3 && gp->grid_n0 == 5 && gp->grid_n1 == 9;
    if (env->env_id == GPMPC_ENV_PENDULUM1D) return gp->g_ny == 1 && gp->grid_n0 == 4 && gp->grid_n1 == 9;
    return false;
}

int rollout_indep_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, hipStream_t st) {
    // the grid root of the plan unless disabled (GPMPC_DISABLE_GRID_ROOT=1 keeps the triangular L_rr^-1 form, used by
    // the tests to compare the two)
    const char* eg = std::getenv("GPMPC_DISABLE_GRID_ROOT");
    const bool grid_root = !(eg && eg[0] == '1') && plan_has_grid_root(gp->grid_n0, gp->grid_n1, gp->real_has_grad);
    if (grid_root) {
        const dim3 grid((unsigned)((args.Ns + 63) / 64));
        if (env->env_id == GPMPC_ENV_CAR_RESIDUAL)
            hipLaunchKernelGGL((rollout_indep_grid_kernel<GPMPC_ENV_CAR_RESIDUAL, 5, 9, 3>), grid, dim3(64 * 3), 0, st, args);
        else
            hipLaunchKernelGGL((rollout_indep_grid_kernel<GPMPC_ENV_PENDULUM1D, 4, 9, 1>), grid, dim3(64), 0, st, args);
    } else {
        const dim3 grid((unsigned)((args.Ns + 255) / 256)), block(256);
        if (env->env_id == GPMPC_ENV_CAR_RESIDUAL)
            hipLaunchKernelGGL((rollout_indep_kernel<GPMPC_ENV_CAR_RESIDUAL, 5, 9, 3>), grid, block, 0, st, args);
        else
            hipLaunchKernelGGL((rollout_indep_kernel<GPMPC_ENV_PENDULUM1D, 4, 9, 1>), grid, block, 0, st, args);
    }
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_indep_phases(long long* out /*[host] 16*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_indep_phase_cycles), 16 * sizeof(long long)));
    return GPMPC_OK;
}
