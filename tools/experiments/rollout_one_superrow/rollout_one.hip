// rollout_one_kernel: the LATENCY form of the re-conditioned rollout (mode R, T = 3 label slots per point, value-only real
// labels on an N0 x 9 tensor grid, at most 96 appended label rows: H <= 33).  gfx950, wave64, ONE chain per wave, one wave
// per SIMD (512 registers).  This is the kernel of BASELINE configs[1] (pendulum1D, Ns = 1024, H = 30: 1024 chains = one per
// SIMD of the chip), where the run time is one wave's latency through H steps.
//
// rollout_fast.hip walks the chain's triangular solve pivot by pivot on the VALU (v_fmac_f64_dpp, ~22 cycles per pivot on
// the dependency chain) and its L_hr v_r product row by row (36 DPP fmacs x 3): 49 % of its step.  Here both run on the
// FP64 matrix pipe with the four blocks of v_mfma_f64_4x4x4_4b_f64 working on FOUR TILE ROWS OF THE SAME CHAIN:
//
//   * lane maps (tools/ubench/mfma64_layout.hip): with kq = lane >> 4, bm = (lane >> 2) & 3, jq = lane & 3, block bm computes
//     D[kq][jq] += sum_k A[.][k] B[k][jq]; B and D use the "natural" map (row kq, column jq), and a natural register X used
//     as the A operand acts as X^T.  A SUPER ROW R is the 16 label rows 16 R .. 16 R + 15 = tile rows 4 R + bm.
//   * the factor is a set of PANELS, one FP64 register each, pinned in AGPRs (rollout_one_gen.inc): panel (R, p) holds
//     L[16 R + 4 bm + jq][4 p + kq] - the A operand that multiplies column tile p into all four tile rows of super row R at
//     once.  The whitened real-data block (grid root, gpmpc_device.hpp) is simply the first NKT = N_r / 4 column tiles.
//   * the right-hand sides are 16 x 4 blocks in the natural map: three kernel-entry columns and the whitened-label column
//     (so the mean falls out of the same Gram product as the covariance, no reduction ladder); the column tile V_p of the
//     solution is kept REPLICATED in all four blocks (the B operand every tile row needs).
//   * left-looking over super rows:  acc = rhs_R - sum_kt PR[R][kt] v_r[kt] - sum_{p < 4R} PH[R][p] V_p  (MFMAs only, the
//     panels straight from AGPRs), then the 16 x 16 diagonal block tile row by tile row: W = GD acc (block q valid),
//     copy block q to the other three blocks with three bank-masked DPP row rotations, acc -= PC[R][q] W.
//   * appended rows are LANES of the panels: in step t the right-hand side of task c sits in column (n_h + c) & 3, so the
//     lane that holds v_p[c] is the lane of the new row's entry (the trick of rollout_tiles.hip) and appending is an
//     EXEC-masked v_accvgpr_write per panel.  No LDS or HBM traffic for the factor at all.
//   * everything else (kernel entries, the grid-root product, the 3 x 3 roots, the sample) is the VALU code of
//     rollout_fast.hip with lane == conditioning point; two small LDS buffers convert "lane = point" into the natural map.
//   * the step loop is unrolled by EPOCH K = n_h >> 4 (number of complete super rows): every register index is static.
//     Super rows 0 .. 4 are register resident; the sixth (rows 80 .. 95: three steps of a 30-step horizon) lives in LDS.
//
// Reference: the loop of benchmarking/simulate_true_reachable_set.py:179-258 / src/agent.py:362-415 (one launch here).
#include "gpmpc_host.hpp"
#include "rollout_args.hpp"

#include <type_traits>
#include <utility>

namespace gpmpc {

#include "rollout_one_gen.inc"

__device__ long long g_one_phase_cycles[16];
__device__ double g_one_dbg[64 * 64];

#ifdef GPMPC_PHASE_TIMERS
#define OPH_DECL long long oph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long opht_ = __builtin_readcyclecounter()
#define OPH(i) do { const long long n_ = __builtin_readcyclecounter(); oph_[i] += n_ - opht_; opht_ = n_; } while (0)
#define OPH_STORE do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) g_one_phase_cycles[i_] = oph_[i_]; } while (0)
#else
#define OPH_DECL
#define OPH(i)
#define OPH_STORE
#endif
#ifndef GPMPC_ONE_DEBUG_STEP
#define GPMPC_ONE_DEBUG_STEP 2
#endif
#ifdef GPMPC_ONE_DEBUG
#define ODBG(slot, val) do { if (blockIdx.x == 0 && t == GPMPC_ONE_DEBUG_STEP) g_one_dbg[(slot) * 64 + lane] = (val); } while (0)
#else
#define ODBG(slot, val)
#endif

constexpr int kOneMaxRows = 96;                                  // six super rows
constexpr int kOneRes = 5;                                       // of which register resident
constexpr int kOneRS = 5;                                        // row stride (doubles) of the lane-map converters: conflict-free b64 access

__device__ __forceinline__ void one_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int B, int E, class F>
__device__ __forceinline__ void one_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        one_for<B + 1, E>(f);
    }
}
// D = A B (C = 0), operands in ordinary registers
__device__ __forceinline__ double one_mfma_zero(double a, double b) {
    double d;
    asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0\n\ts_nop 5" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
// acc -= A B
__device__ __forceinline__ void one_mfma_nacc(double& acc, double a, double b) {
    asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0 neg:[1,0,0]\n\ts_nop 5" : "+v"(acc) : "v"(a), "v"(b));
}
// Copy block Q (the Q-th quad of every DPP row) of w into the other three blocks: three bank-masked row rotations per
// dword.  row_ror:n moves lane i of a row to lane i + n; the write is confined to bank (Q + d) & 3.
template <int Q>
__device__ __forceinline__ double one_replicate(double w) {
    int lo = __double2loint(w), hi = __double2hiint(w);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x124, 0xf, 1 << ((Q + 1) & 3), false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x124, 0xf, 1 << ((Q + 1) & 3), false);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xf, 1 << ((Q + 2) & 3), false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xf, 1 << ((Q + 2) & 3), false);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x12c, 0xf, 1 << ((Q + 3) & 3), false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x12c, 0xf, 1 << ((Q + 3) & 3), false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double one_bpermute(double v, int addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// acc0 += R0@(lane I of the DPP row) * l ; acc1 += R1@(lane I) * l
template <int I>
__device__ __forceinline__ void one_fmac2_bcast(double& acc0, double& acc1, double R0, double R1, double l) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %3, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(acc0), "+v"(acc1)
        : "v"(l), "v"(R0), "v"(R1), "n"(I));
}
template <int N, int OFS, int J = 0>
__device__ __forceinline__ void one_axis_product(double& P0, double& P1, double R0, double R1, const double (&coef)[N]) {
    if constexpr (J < N) {
        one_fmac2_bcast<OFS + J>(P0, P1, R0, R1, coef[J]);
        one_axis_product<N, OFS, J + 1>(P0, P1, R0, R1, coef);
    }
}

__device__ __forceinline__ double one_pick3(int i, double v0, double v1, double v2) {
    const double t = (i == 1) ? v1 : v2;
    return (i == 0) ? v0 : t;
}

template <int N0>
struct OneLds {
    static constexpr int NKT = (N0 * 9 + 3) / 4;
    static constexpr int NP5 = NKT + 4 * kOneRes + 4;             // panels of the LDS-resident super row: PR | PH | PC | GD
    static constexpr int VR = 0;                                  // [4 NKT][RS]  v_r rows (natural-map source)
    static constexpr int HS = ((4 * NKT * kOneRS + 1) & ~1);      // [96][RS]     right-hand sides of the appended rows
    static constexpr int P5 = HS + kOneMaxRows * kOneRS;          // [NP5][64]
    static constexpr int TOTAL = P5 + NP5 * 64;
};

template <int N0, int ENV>
__global__ __launch_bounds__(64, 1) void rollout_one_kernel(const RolloutArgs a) {
    static_assert(ENV == GPMPC_ENV_PENDULUM1D && N0 == 4, "instantiated for the pendulum1D 4 x 9 grid");
    constexpr int D = 2, T = 3, N1 = 9, NR = N0 * N1, NX = 2;
    using L = OneLds<N0>;
    constexpr int NKT = L::NKT;
    using Panels = OnePanels_k9;
    static_assert(NKT == 9 && N0 + N1 <= 16, "panel map generated for NKT = 9");
    extern __shared__ __attribute__((aligned(16))) double smem[];

    const GpParams& gp = a.gp;
    const int lane = threadIdx.x;
    const int kq = lane >> 4, bm = (lane >> 2) & 3, jq = lane & 3;
    const long s = blockIdx.x;
    const int H = a.H;
    double* VRb = smem + L::VR;
    double* HSb = smem + L::HS;
    double* P5b = smem + L::P5;

    // ---- per-lane constants ------------------------------------------------------------------------------------------
    const double il0 = gp.inv_l2[0][0], il1 = gp.inv_l2[0][1], os = gp.os[0];
    // lane = real point (ga, gc) of the grid: columns of Qa / Qb, os / sqrt(D) and the whitened label of the point
    const int lr = (lane < NR) ? lane : 0, ga = lr / N1, gc = lr - ga * N1;
    double qa[N0], qb[N1];
#pragma unroll
    for (int j = 0; j < N0; ++j) qa[j] = plan_grid_Qa(a.plan, gp, 0)[j * N0 + ga];
#pragma unroll
    for (int j = 0; j < N1; ++j) qb[j] = plan_grid_Qb(a.plan, gp, 0)[j * N1 + gc];
    const double dsc = (lane < NR) ? plan_grid_dsc(a.plan, gp, 0)[lr] : 0.0;
    const double w_lane = (lane < NR) ? plan_grid_w(a.plan, gp, 0)[lr] : 0.0;
    // axis lanes: lane j < N0 holds axis-0 point j (real point N1 j), lane N0 + j axis-1 point j
    const bool g_ax0 = lane < N0;
    const int g_pt = g_ax0 ? lane * N1 : ((lane < N0 + N1) ? lane - N0 : 0);
    const double g_x = a.X_r[g_pt * D + (g_ax0 ? 0 : 1)], g_il2 = g_ax0 ? il0 : il1;
    const int bp_addr = (lane & 15) << 2;                         // ds_bpermute address of "my lane of DPP row 0"
    const double Inat = (kq == jq) ? 1.0 : 0.0;
    // lane-map converter addresses (doubles): natural-map reads
    const int vr_rd = kq * kOneRS + jq;                           // + 4 kt RS
    const int hs_rd = (4 * bm + kq) * kOneRS + jq;                // + 16 R RS
    const int rA = 4 * bm + jq;                                   // row of this lane's panel entry inside its super row

    double x[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) x[d] = a.x0[(a.x0_per_sample ? s * NX : 0) + d];
    double xq[NX] = {0.0, 0.0};                                   // trajectory, one step per lane
    double zq[T], uq;
#pragma unroll
    for (int c = 0; c < T; ++c) zq[c] = (lane < H) ? a.z[(long)lane * a.z_step_stride + s * T + c] : 0.0;
    uq = (lane < H) ? a.u_ff[lane] : 0.0;
    double xh[D] = {0.0, 0.0}, yt[T] = {0.0, 0.0, 0.0};           // lane = appended point: its GP input and label residuals
    // the diagonal tiles of the super row being appended to (one tile per block): L^T in the natural map and 1 / diag along its
    // rows / columns; the same for the next super row (rows that wrap into it)
    double ud = Inat, drow = 1.0, dcol = 1.0, ud1 = Inat, drow1 = 1.0, dcol1 = 1.0;
    int info_acc = 0;
    int n_h = 0, t = 0;

    // ---- the factor --------------------------------------------------------------------------------------------------
    Panels P;
    {
        const unsigned long long all = ~0ull;
        double zeros[4 * kOneRes];
#pragma unroll
        for (int i = 0; i < 4 * kOneRes; ++i) zeros[i] = 0.0;
        one_for<0, kOneRes>([&](auto Rc) {
            constexpr int R = decltype(Rc)::value;
            one_set_pr_k9<R>(P, all, zeros);
            one_set_ph_k9<R>(P, all, zeros);
            one_set_pc_k9<R, 0>(P, all, 0.0);
            one_set_pc_k9<R, 1>(P, all, 0.0);
            one_set_pc_k9<R, 2>(P, all, 0.0);
            one_set_gd_k9<R>(P, all, Inat);
        });
#pragma unroll
        for (int i = 0; i < L::NP5; ++i) P5b[i * 64 + lane] = (i == L::NP5 - 1) ? Inat : 0.0;
    }
    one_sync_lds();
    OPH_DECL;

    auto step = [&](auto Kc) {
        constexpr int K = decltype(Kc)::value;                    // complete super rows; row K is the partial one
        constexpr int NTK = 4 * K + 4;                            // column tiles this epoch can touch
        const int i0 = n_h & 3, ycol = (i0 + 3) & 3, npts = t;
        const int cb0 = i0, cb1 = (i0 + 1) & 3, cb2 = (i0 + 2) & 3;
        // ---- input, GP input ---------------------------------------------------------------------------------------
        double u, xi[D];
        {
            const double uf = readlane_f64(uq, t);
            if (a.env.use_feedback) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) acc += (a.env.x_goal[j] - x[j]) * a.env.K[0][j];
                u = -acc + uf;
            } else {
                u = uf;
            }
            xi[0] = x[0];
            xi[1] = u;
        }
#pragma unroll
        for (int d = 0; d < NX; ++d) xq[d] = (lane == t) ? x[d] : xq[d];
        if (lane == 0 && a.Xi) {
#pragma unroll
            for (int d = 0; d < D; ++d) a.Xi[(s * H + t) * D + d] = xi[d];
        }

        // ---- kernel factors: the grid axis factor of this lane and the appended point of this lane, in lockstep --------
        double ea, gq, kk, q0, q1;
        {
            const double gr = g_x - (g_ax0 ? xi[0] : xi[1]);
            gq = gr * g_il2;
            const double d0 = xh[0] - xi[0], d1 = xh[1] - xi[1];
            q0 = d0 * il0;
            q1 = d1 * il1;
            const double a2[2] = {-0.5 * gr * gq, -0.5 * (d0 * q0 + d1 * q1)};
            double e2[2];
            expn_neg<2>(a2, e2);
            ea = e2[0];
            kk = (lane < npts) ? os * e2[1] : 0.0;
        }
        // ---- v_r = W k_r through the grid root (rollout_fast.hip, step 2): lane = real point ------------------------------
        double vr[T];
        {
            const double R0 = one_bpermute(ea, bp_addr), R1 = one_bpermute(ea * gq, bp_addr);
            double PA0 = 0.0, PA1 = 0.0, PB0 = 0.0, PB1 = 0.0;
            one_axis_product<N0, 0>(PA0, PA1, R0, R1, qa);
            one_axis_product<N1, N0>(PB0, PB1, R0, R1, qb);
            const double s0 = dsc * PB0;
            vr[0] = s0 * PA0;
            vr[1] = s0 * PA1;
            vr[2] = dsc * PA0 * PB1;
        }
        if (lane < 4 * NKT) {                                     // rows of v_r, task column c at (i0 + c) & 3, whitened label beside
            double* dst = VRb + lane * kOneRS;
            dst[cb0] = vr[0];
            dst[cb1] = vr[1];
            dst[cb2] = vr[2];
            dst[ycol] = w_lane;
        }
        // ---- right-hand sides of the appended rows: lane = point, cov(task a of the point, task b of the test point) ----
        if (n_h > 0 && lane < kOneMaxRows / 3) {
            const double Aa[T] = {1.0, -q0, -q1}, Bb[T] = {1.0, q0, q1}, cd[T] = {0.0, il0, il1};
#pragma unroll
            for (int aa = 0; aa < T; ++aa) {
                double* dst = HSb + (3 * lane + aa) * kOneRS;
                dst[cb0] = kk * Aa[aa];
                dst[cb1] = kk * fma(Aa[aa], Bb[1], (aa == 1) ? cd[1] : 0.0);
                dst[cb2] = kk * fma(Aa[aa], Bb[2], (aa == 2) ? cd[2] : 0.0);
                dst[ycol] = (lane < npts) ? yt[aa] : 0.0;
            }
        }
        one_sync_lds();
        double VrRep[NKT], RN[K + 1];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) VrRep[kt] = VRb[vr_rd + 4 * kt * kOneRS];
#pragma unroll
        for (int R = 0; R <= K; ++R) RN[R] = HSb[hs_rd + 16 * R * kOneRS];
        OPH(0);

        // ---- Gram of the real block: S_r' = sum_kt v_r[kt]^T v_r[kt] (every block holds the whole sum) ------------------
        double Sr0 = 0.0, Sr1 = 0.0;
        one_pchain<NKT>(Sr0, Sr1, VrRep, VrRep);
        OPH(1);

        // ---- forward substitution, left-looking over super rows; S_h' += V_p^T V_p --------------------------------------
        double Vrep[NTK];
#pragma unroll
        for (int p = 0; p < NTK; ++p) Vrep[p] = 0.0;
        double Sh0 = Sr0, Sh1 = Sr1;                              // the appended rows accumulate on top of the real block
        double Vinc = 0.0;                                        // V of the incomplete tile row (phase H wants it)
        if (n_h > 0) {
            one_for<0, K + 1>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                if (R < K || n_h > 16 * K) {                      // (uniform) the partial super row may be empty
                    double c0 = RN[R], c1 = 0.0;
                    double gd5 = 0.0, pc5[3] = {0.0, 0.0, 0.0};
                    if constexpr (R < kOneRes) {
                        one_off_k9<R>(P, c0, c1, VrRep, Vrep);
                    } else {
                        // the LDS-resident super row: panels as ordinary operands, a dozen at a time
                        double A[12];
#pragma unroll
                        for (int i = 0; i < NKT; ++i) A[i] = P5b[i * 64 + lane];
                        one_nchain<NKT>(c0, c1, A, VrRep);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 12; ++i) A[i] = P5b[(NKT + i) * 64 + lane];
                        one_nchain<12>(c0, c1, A, Vrep);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 8; ++i) A[i] = P5b[(NKT + 12 + i) * 64 + lane];
                        one_nchain<8>(c0, c1, A, Vrep + 12);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 3; ++i) pc5[i] = P5b[(NKT + 20 + i) * 64 + lane];
                        gd5 = P5b[(NKT + 23) * 64 + lane];
                    }
                    double acc = c0 + c1;
                    if constexpr (R == 0) {
                        ODBG(10, acc);
                        ODBG(12, one_gdm_k9<0>(P, Inat));
                    }
                    // tile rows of the partial super row that do not exist yet are skipped (early-exit chain: a taken branch
                    // of a lone wave costs an instruction fetch)
                    const int rem = n_h - 16 * K;
                    auto tile_from = [&](auto self, auto qc) -> void {
                        constexpr int q = decltype(qc)::value;
                        if constexpr (q < 4) {
                            if (R < K || rem > 4 * q) {
                                double w;
                                if constexpr (R < kOneRes) w = one_gdm_k9<R>(P, acc);
                                else w = one_mfma_zero(gd5, acc);
                                if constexpr (R == 0 && q == 0) ODBG(11, w);
                                w = one_replicate<q>(w);
                                Vrep[4 * R + q] = w;
                                if constexpr (q < 3) {
                                    if constexpr (R < kOneRes) one_pcm_k9<R, q>(P, acc, w);
                                    else one_mfma_nacc(acc, pc5[q], w);
                                }
                                if constexpr (R == K) one_pchain<1>(Sh0, Sh1, Vrep + 4 * R + q, Vrep + 4 * R + q);
                                self(self, std::integral_constant<int, q + 1>{});
                            }
                        }
                    };
                    tile_from(tile_from, std::integral_constant<int, 0>{});
                    if constexpr (R < K) one_pchain<4>(Sh0, Sh1, Vrep + 4 * R, Vrep + 4 * R);
                }
            });
            if ((n_h & 3) != 0) {
                const int bt = (n_h >> 2) & 3;
                Vinc = (bt == 0) ? Vrep[4 * K] : ((bt == 1) ? Vrep[4 * K + 1] : ((bt == 2) ? Vrep[4 * K + 2] : Vrep[4 * K + 3]));
            }
        }
        ODBG(0, Vrep[0]);
        ODBG(1, Vrep[1]);
        ODBG(2, Vrep[2]);
        ODBG(3, Vrep[3]);
        ODBG(4, VrRep[0]);
        ODBG(5, RN[0]);
        OPH(2);

        // ---- S' to scalars: entry [k][j] sits in lane 16 k + j of block 0 ------------------------------------------------
        double mu[T], S[T][T];
        {
            const double Stot = Sh0 + Sh1;
            ODBG(6, Stot);
            const int cb[T] = {cb0, cb1, cb2};
#pragma unroll
            for (int b = 0; b < T; ++b) {
                mu[b] = readlane_f64(Stot, 16 * cb[b] + ycol);
#pragma unroll
                for (int c = 0; c <= b; ++c) {
                    const double kss = (b == c) ? ((b == 0) ? os : ((b == 1) ? os * il0 : os * il1)) : 0.0;
                    const double val = kss - readlane_f64(Stot, 16 * cb[b] + cb[c]);
                    S[b][c] = val;
                    S[c][b] = val;
                }
            }
        }
        OPH(3);
        // ---- variance floor, roots, sample (as sample_gp, src/agent.py:629-708) -------------------------------------------
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
        double Rt[T][T], C[T][T], cinv[T];
        bool c_ok;
        {
            double Sn[T][T], rinv[T];
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int c = 0; c < T; ++c) Sn[b][c] = S[b][c] + ((b == c) ? gp.noise[b] : 0.0);
            bool r_ok;
            chol3_pair_fast(Sn, S, C, Rt, cinv, rinv, c_ok, r_ok);
            if (!r_ok) info_acc |= root_small_fast_retry<T>(S, gp.jitter, Rt);
        }
        double zt[T];
#pragma unroll
        for (int c = 0; c < T; ++c) zt[c] = readlane_f64(zq[c], t);
        double y[T];
#pragma unroll
        for (int b = 0; b < T; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c <= b; ++c) acc = fma(Rt[b][c], zt[c], acc);
            double yb = acc + mu[b];
            if (all_zero) yb = mu[b];
            const double dlt = yb - mu[b];
            if (dlt * dlt > a.beta * a.beta * var[b]) {
                const double sd = a.beta * sqrt(var[b]);
                yb = fmin(fmax(yb, mu[b] - sd), mu[b] + sd);
            }
            y[b] = yb;
        }
        if (lane == 0 && a.Y) {
#pragma unroll
            for (int b = 0; b < T; ++b) a.Y[(s * H + t) * T + b] = y[b];
        }
        OPH(4);

        // ---- append the point (A.9): three rows of the factor = lanes of the panels ----------------------------------------
        if (t + 1 < H) {
            if (!c_ok) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
            {
                const bool mine = lane == npts;
                xh[0] = mine ? xi[0] : xh[0];
                xh[1] = mine ? xi[1] : xh[1];
#pragma unroll
                for (int b = 0; b < T; ++b) yt[b] = mine ? y[b] : yt[b];   // the label column is whitened by the same MFMAs (PR w_r = mu_real)
            }
            const int tn = n_h >> 2, bt = tn & 3, lo = n_h - 16 * K;
            // C[ci][ck] by lane-varying indices (clamped to 0 .. 2); by VALUE: a select between captured references is a
            // select of addresses, which hipcc turns into a table of pointers in scratch memory
            const double c00 = C[0][0], c10 = C[1][0], c11 = C[1][1], c20 = C[2][0], c21 = C[2][1], c22 = C[2][2];
            const double ci0 = cinv[0], ci1 = cinv[1], ci2 = cinv[2];
            auto c_pick = [=](int ci, int ck) -> double {
                const double r0 = one_pick3(ci, c00, c10, c20);
                const double r1 = one_pick3(ci, c11, c11, c21);
                return one_pick3(ck, r0, r1, c22);
            };
            // A new row (super-row-local index ri = 0 .. 2 counted from the first new row) against column rk (same origin):
            // old columns (rk < 0) carry v of the incomplete tile row, new ones the 3 x 3 factor.  `up` = 16 for the lanes
            // whose new row wrapped into super row K + 1.
            auto new_entry = [&](int ri, int rk) -> double {
                const int ci = min(max(ri, 0), 2), ck = min(max(rk, 0), 2);
                const double cval = (rk <= ri) ? c_pick(ci, ck) : 0.0;
                return (rk < 0) ? Vinc : cval;
            };
            // rows lo .. lo + 2 of super row K (panel lanes: row rA = 4 bm + jq), or wrapped into K + 1 (its block 0)
            const bool newK = (rA >= lo) && (rA < lo + 3);
            const bool newK1 = rA < lo + 3 - 16;
            const unsigned long long mK = __ballot(newK), mK1 = __ballot(newK1);
            const int ri = rA - lo + (newK1 ? 16 : 0);
            const double mixC = new_entry(ri, kq - (n_h & 3));                    // against the columns of the incomplete tile tn
            // super row K
            if constexpr (K < kOneRes) {
                one_set_pr_k9<K>(P, mK, VrRep);
                one_set_ph_k9<K>(P, mK, Vrep);
                one_for<0, 3>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const unsigned long long mq = __ballot(newK && bm > q);
                    one_set_pc_k9<K, q>(P, mq, (bt == q) ? mixC : Vrep[4 * K + q]);
                });
            } else {
                if (newK) {
#pragma unroll
                    for (int i = 0; i < NKT; ++i) P5b[i * 64 + lane] = VrRep[i];
#pragma unroll
                    for (int i = 0; i < 4 * kOneRes; ++i) P5b[(NKT + i) * 64 + lane] = Vrep[i];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        if (bm > q) P5b[(NKT + 20 + q) * 64 + lane] = (bt == q) ? mixC : Vrep[4 * K + q];
                    }
                }
            }
            OPH(5);
            // The diagonal tiles of super row K: ud = L^T of the lane's own tile (natural map, one tile per block), drow / dcol =
            // 1 / diag along its rows / columns.  New rows enter by select; ALL FOUR tile inverses come out of the same three
            // MFMAs, U^-1 = (I + M)(I + M^2) D^-1 with U = D (I - M) (rollout_tiles.hip, phase H) - complete tiles reproduce
            // what they had, so the whole register is committed.
            auto inverse_tiles = [&](double U, double dr, double dcl) -> double {
                const double M = Inat - dr * U;
                const double Mt = one_mfma_zero(M, Inat);                     // M^T
                const double M2 = one_mfma_zero(Mt, M);                       // M M
                const double Pq = one_mfma_zero(Inat + Mt, Inat + M2);        // (I + M)(I + M^2)
                return Pq * dcl;
            };
            {
                // natural map of L^T: row index of L = 4 bm + jq (= rA), column index = 4 bm + kq
                const int rkD = 4 * bm + kq - lo;
                const double mixD = new_entry(rA - lo, rkD);
                ud = newK ? mixD : ud;
                const bool newRowK = (rkD >= 0) && (rkD < 3);            // the lane's L-column index is a new row
                const double cK = one_pick3(min(max(rkD, 0), 2), ci0, ci1, ci2);
                const double cJ = one_pick3(min(max(rA - lo, 0), 2), ci0, ci1, ci2);
                drow = newRowK ? cK : drow;
                dcol = newK ? cJ : dcol;
                const double Gt = inverse_tiles(ud, drow, dcol);
                ODBG(8, ud);
                ODBG(9, Gt);
                if constexpr (K < kOneRes) one_set_gd_k9<K>(P, ~0ull, Gt);
                else P5b[(L::NP5 - 1) * 64 + lane] = Gt;
            }
            // rows that wrap into super row K + 1 (its block 0): every column tile of super row K is old or the mix
            if (lo + 3 > 16) {                                    // uniform
                double Vt[NTK];
#pragma unroll
                for (int p = 0; p < NTK; ++p) Vt[p] = Vrep[p];
                Vt[NTK - 1] = mixC;                               // wrapping implies tn = 4 K + 3
                if constexpr (K + 1 < kOneRes) {
                    one_set_pr_k9<K + 1>(P, mK1, VrRep);
                    one_set_ph_k9<K + 1>(P, mK1, Vt);
                } else if constexpr (K + 1 == kOneRes) {
                    if (newK1) {
#pragma unroll
                        for (int i = 0; i < NKT; ++i) P5b[i * 64 + lane] = VrRep[i];
#pragma unroll
                        for (int i = 0; i < NTK; ++i) P5b[(NKT + i) * 64 + lane] = Vt[i];
                    }
                }
                // its first diagonal tile: rows ri = rA + 16 - lo, columns 4 bm + kq + 16 - lo (all new)
                const int rkD1 = 4 * bm + kq + 16 - lo;
                const double mixD1 = new_entry(rA + 16 - lo, rkD1);
                ud1 = newK1 ? mixD1 : ud1;
                const double cK1 = one_pick3(min(max(rkD1, 0), 2), ci0, ci1, ci2);
                const double cJ1 = one_pick3(min(max(rA + 16 - lo, 0), 2), ci0, ci1, ci2);
                drow1 = (rkD1 < 3) ? cK1 : drow1;
                dcol1 = newK1 ? cJ1 : dcol1;
                const double G1 = inverse_tiles(ud1, drow1, dcol1);
                if constexpr (K + 1 < kOneRes) one_set_gd_k9<K + 1>(P, ~0ull, G1);
                else if constexpr (K + 1 == kOneRes) P5b[(L::NP5 - 1) * 64 + lane] = G1;
            }
            n_h += T;
            one_sync_lds();
        }
        OPH(6);

        // ---- state hand-over ---------------------------------------------------------------------------------------------
        {
            const double x0n = x[0] + x[1] * a.env.dt;
            x[1] = x[1] + y[0];
            x[0] = x0n;
        }
        t += 1;
        OPH(7);
    };

    one_for<0, kOneMaxRows / 16>([&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
#pragma unroll 1
        while (t < H && (n_h >> 4) == K) step(Kc);
        ud = ud1, drow = drow1, dcol = dcol1;                     // the next super row becomes the current one
        ud1 = Inat, drow1 = 1.0, dcol1 = 1.0;
    });

    if (lane <= H) {
#pragma unroll
        for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + lane] = (lane == H) ? x[d] : xq[d];
    }
    if (lane == 0) a.info[s] = info_acc;
    OPH_STORE;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int one_mode() {                                          // 0 auto, 1 forced, -1 disabled
    const char* e = std::getenv("GPMPC_ROLLOUT_ONE");
    if (!e) return 0;
    return (e[0] == '1') ? 1 : ((e[0] == '0') ? -1 : 0);
}

bool rollout_one_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode, int hall_tasks, int H, int64_t Ns) {
    const int md = one_mode();
    if (md < 0) return false;
    const char* e = std::getenv("GPMPC_DISABLE_FAST_ROLLOUT");
    if (e && e[0] == '1') return false;
    const char* eg = std::getenv("GPMPC_DISABLE_GRID_ROOT");
    if (eg && eg[0] == '1') return false;
    const char* ef = std::getenv("GPMPC_FORCE_GLOBAL_FACTOR");
    if (ef && ef[0] == '1') return false;
    if (mode != GPMPC_MODE_RECONDITIONED || gp->T != 3 || gp->D != 2 || hall_tasks != 3 || gp->real_has_grad) return false;
    if (!plan_has_grid_root(gp->grid_n0, gp->grid_n1, gp->real_has_grad)) return false;
    if (env->env_id != GPMPC_ENV_PENDULUM1D || gp->g_ny != 1 || gp->grid_n0 != 4 || gp->grid_n1 != 9) return false;
    if (H < 2 || 3 * (H - 1) > kOneMaxRows) return false;
    if (md > 0) return true;
    // one chain per wave, one wave per SIMD: up to two rounds of the chip (2048 chains) it beats four chains per wave
    return Ns <= 2048;
}

int rollout_one_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, hipStream_t st) {
    (void)gp;
    (void)env;
    const size_t lds = (size_t)OneLds<4>::TOTAL * sizeof(double);
    auto k = rollout_one_kernel<4, GPMPC_ENV_PENDULUM1D>;
    GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)args.Ns), dim3(64), lds, st, args);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_one_phases(long long* out /*[host] 16*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_one_phase_cycles), 16 * sizeof(long long)));
    return GPMPC_OK;
}
extern "C" int gpmpc_debug_read_one(double* out /*[host] 4096*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_one_dbg), 64 * 64 * sizeof(double)));
    return GPMPC_OK;
}
