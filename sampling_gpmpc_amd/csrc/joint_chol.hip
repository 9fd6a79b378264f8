// joint_chol_mfma_kernel: Cholesky of the Schur complement of the NEW hallucinated rows of a joint draw (mode "J") on the FP64
// matrix pipe, one WAVE per (sample, output) chain, the whole matrix in registers (gfx950).
//
// Where it stands: the matrix-pipe path of gpmpc_joint_sample (joint.hip) extends the chain's factor by the rows of this SQP
// iteration's points (reference src/agent.py:164-202 appends them, src/agent.py:241-250 / 640 re-factorises everything on the
// next model_i(x) call): joint_test_mfma_kernel (JOINT_MFMA_FACTOR) forms the new rows' entries against the OLD columns and
// leaves the Schur complement S = K_nn + noise - L_no L_no^T (n_new x n_new, n_new <= 128: one iteration's H T slots) in the
// chain's S buffer; its Cholesky factor is the new rows' entries against the NEW columns.  joint_kernel's CHOL phase did that
// with one label row per thread - 120 sequential pivots behind workgroup barriers, the matrix in LDS (one chain per CU):
// 0.42-0.46 ms per draw at the configs[4] shard, 3-5 % of the FP64 peak's worth of a 1.8 GFLOP job, 10-15 % of every draw
// from the third SQP iteration on (profiles/r5_closed_loop_trace.md).
//
// Here: S is held as UPPER-triangular 16 x 16 tiles A_kj (k <= j), four FP64 registers per tile in the D layout of
// v_mfma_f64_16x16x4_f64 (register v, lane l: row 4 v + (l >> 4), column l & 15), 36 tiles = 288 registers, one wave per SIMD.
// That layout is at once the instruction's B layout (K-slice v) and - read as the A operand - the TRANSPOSE of the tile
// (A[i][k] of K-slice v sits in lane i + 16 k = the D-layout place of element [4 v + k][i]):
//      nat(X, Y) := sum_v mfma(A = X.v, B = Y.v)  =  X^T Y          for two tiles in D layout, no conversion of either.
// Left-looking blocked Cholesky S = R^T R over tile rows (R upper; the factor's rows are R's columns):
//      acc_j = A_kj + sum_{p<k} nat(-R_pk, R_pj)                     j >= k
//      R_kk = chol(acc_k)^T, V = R_kk^-1                             (the only step off the matrix pipe, see below)
//      R_kj = R_kk^-T acc_j = nat(V, acc_j)                          j > k
// 4 MFMAs per tile product, 448 per chain at eight tiles; a finished tile stays in registers only while a later row needs it
// (<= 20 tiles alive, the next row's A tiles requested a step ahead: everything in the 256 VALU-addressable registers).  The diagonal tile goes through LDS into "lane = row" form: a
// right-looking 16 x 16 Cholesky whose updates are v_fmac_f64_dpp row_newbcast (the pivot row's entry one DPP read away), and in
// the same sweep W = L_kk^-1 by forward substitution on the identity (row-oriented, the finished rows broadcast the same way;
// backward stable - a Neumann product on the matrix pipe would square the tile's condition); both back to D layout through LDS.  Results go to the factor cache transposed through LDS, 128 bytes
// per row segment.  Values agree with the phase it replaces to rounding (the update sums in another order).
#include <type_traits>

#include "gpmpc_device.hpp"
#include "joint_args.hpp"

namespace gpmpc {

typedef double jc_d4 __attribute__((ext_vector_type(4)));
typedef double jc_d2 __attribute__((ext_vector_type(2)));
typedef unsigned jc_u2 __attribute__((ext_vector_type(2)));

constexpr int JC_LD = 18;                      // LDS row stride (doubles): rows 16-byte aligned, transposed reads conflict-free

template <int B, int E, class F>
__device__ __forceinline__ void jc_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        jc_for<B + 1, E>(f);
    }
}
__device__ __forceinline__ void jc_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__host__ __device__ constexpr int jc_idx(int k, int j, int ntl) { return k * ntl - k * (k - 1) / 2 + (j - k); }   // k <= j
// X^T Y for two tiles in D layout, accumulated onto C
__device__ __forceinline__ jc_d4 jc_nat(const jc_d4& X, const jc_d4& Y, jc_d4 C) {
#pragma unroll
    for (int v = 0; v < 4; ++v) C = __builtin_amdgcn_mfma_f64_16x16x4f64(X[v], Y[v], C, 0, 0, 0);
    return C;
}
// acc -= m * src@lane J of the DPP row (src is read through DPP: it must not have been written in the two instructions before)
// (EVERY statement carries its own two wait states: the register a DPP read names must not have been written by the VALU in the two
// instructions before, and hipcc is free to put a copy of an operand right in front of an asm statement - it did, at the last pivot of
// the eight-tile instance; tools/check_dpp_hazard.py, a CPU test, scans this file's ISA for exactly that)
template <int J, bool NOP = true>
__device__ __forceinline__ void jc_fnma_bcast(double& acc, double src, double m) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(J));
}

// acc -= m * acc@lane J of the DPP row (accumulator and broadcast source are the SAME register: as two asm operands hipcc copies it)
template <int J, bool NOP = true>
__device__ __forceinline__ void jc_fnma_self(double& acc, double m) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %0, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(m), "n"(J));
}

// GPMPC_PHASE_TIMERS: cycles of chain 0's wave per phase (loads | updates | diagonal tile | panel | stores), tools/debug/chol_phases.py
__device__ long long g_jc_phase[8];
#ifdef GPMPC_PHASE_TIMERS
#define JCPH(i) do { const long long n_ = __builtin_readcyclecounter(); jph[i] += n_ - jt; jt = n_; } while (0)
#else
#define JCPH(i)
#endif

// The diagonal tile A (D layout, symmetric) -> R = chol(A)^T and V = R^-1 (both D layout), 1 / diag of the lane's row (lane & 15);
// `bad` is set where a pivot is not positive.  blk / vt: two 16 x JC_LD LDS buffers of this wave.
__device__ __forceinline__ void jc_diag_tile(double* blk, double* vt, const jc_d4& A, int lr, int lc, bool& bad, double& mydinv_out,
                                             jc_d4& R, jc_d4& V) {
    jc_sync();                                            // the previous readers of blk / vt are done
#pragma unroll
    for (int v = 0; v < 4; ++v) blk[(4 * v + lr) * JC_LD + lc] = A[v];
    jc_sync();
    double r[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const jc_d2 t = *reinterpret_cast<const jc_d2*>(&blk[lc * JC_LD + 2 * q]);
        r[2 * q] = (2 * q <= lc) ? t.x : 0.0;
        r[2 * q + 1] = (2 * q + 1 <= lc) ? t.y : 0.0;
    }
    // right-looking: after pivot j every later column takes its update at once - 15 - j INDEPENDENT DPP FMAs (lane i:
    // A[i][q] -= L[i][j] L[q][j], L[q][j] = lane q's r[j]) instead of a chain of j dependent ones per pivot
    // (a column of L leaves for LDS the moment it is final and a column of W' is born at its own pivot: 16 doubles alive
    // between them, not 32)
    double mydinv = 0.0;
    double w[16];                                         // row lc of W' = D L^-1 (W'[i][:] = e_i - sum_k L[i][k] W[k][:])
    jc_sync();                                            // every lane has read its row of blk
    jc_for<0, 16>([&](auto jcn) {
        constexpr int j = decltype(jcn)::value;
        const double d = readlane_f64(r[j], j);
        if (!(d > 0.0)) bad = true;
        // v_rsq_f64 + ONE Newton step (3e-16 on 1 / sqrt, as rollout_one.hip) and sqrt = d / sqrt: six dependent operations on
        // the serial spine of the tile instead of the twelve of sqrt_rsqrt_fast
        double inv = __builtin_amdgcn_rsq(d);
        inv = fma(inv, fma(-0.5 * d * inv, inv, 0.5), inv);
        const double sd = d * inv;
        const double lj = (lc > j) ? r[j] * inv : 0.0;    // L[i][j] below the diagonal, zero elsewhere
        r[j] = (lc == j) ? sd : lj;
        mydinv = (lc == j) ? inv : mydinv;
        const double mj = lj * inv;                       // L[i][j] / L[j][j]: the multiplier of W'[j][:] for the rows below j
        w[j] = (lc == j) ? 1.0 : 0.0;
        if (lr == 0) blk[lc * JC_LD + j] = r[j];          // L_kk[lc][j] (zero above the diagonal)
        // (r[j] and mj are VALU results of this step, w[] of the previous one: the first DPP read of each group carries the wait)
        jc_for<j + 1, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            jc_fnma_bcast<q, q == j + 1>(r[q], r[j], lj);
        });
        jc_for<0, j + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            jc_fnma_self<j, c == 0>(w[c], mj);
        });
    });
    if (lr == 0) {                                        // W = D^-1 W', row-major (above the diagonal: zero)
#pragma unroll
        for (int q = 0; q < 8; ++q)
            *reinterpret_cast<jc_d2*>(&vt[lc * JC_LD + 2 * q]) = jc_d2{w[2 * q] * mydinv, w[2 * q + 1] * mydinv};
    }
    jc_sync();
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int rr = 4 * v + lr;
        V[v] = vt[lc * JC_LD + rr];                       // V[rr][lc] = W[lc][rr] (zero below the diagonal)
        R[v] = blk[lc * JC_LD + rr];   // R_kk[rr][lc] = L_kk[lc][rr] (zero below the diagonal)
    }
    mydinv_out = mydinv;
}

#ifndef GPMPC_JC_OCC8
#define GPMPC_JC_OCC8 1                        // waves per SIMD of the eight-tile instance (2: 256 registers, ~45 of them spilled: no faster)
#endif
template <int NTL>
__global__ __launch_bounds__(64, (NTL == 8) ? GPMPC_JC_OCC8 : 2) void joint_chol_mfma_kernel(const JointArgs a) {
    constexpr int NTT = NTL * (NTL + 1) / 2;
    constexpr int TS = 16 * JC_LD;                                // doubles of one tile buffer
    __shared__ __attribute__((aligned(16))) double blk[TS];       // the diagonal tile: A_kk, then L_kk (lower, row-major)
    __shared__ __attribute__((aligned(16))) double vt[NTL * TS];  // W = L_kk^-1 row-major; the output transposes of a column's tiles
    const int lane = threadIdx.x, lr = lane >> 4, lc = lane & 15;
    const int n_r = a.gp.n_r, n_c = a.n_c, n = a.n_ho - a.n_c, CS = a.fc_cs;
    const long mT = (long)a.m * a.gp.T;
    {
        const long chain = a.chain0 + blockIdx.x;                 // one workgroup (= one wave) per chain
        double* fc = a.fcache + (chain - a.fc_chain_base) * a.fc_stride;
        double* fdinv = fc + (long)a.fc_cap * CS;
        // the Schur complement (both triangles) through a buffer descriptor: rows beyond n are beyond the buffer's end and read as
        // zero without an address clamp; columns beyond n are masked on the value.  It sits in the chain's S buffer (leading dimension
        // n: JOINT_MFMA_FACTOR left it there) or - pend_use - in the diagonal block of the very cache rows this kernel writes (leading
        // dimension CS: the previous call's test-mode launch left S = K** - X^T X there; the likelihood noise of the rows' tasks goes
        // onto the diagonal here).  In place: a tile is read a whole step before the first store into its rows.
        const bool pend = a.pend_use != 0;
        const int ldS = pend ? CS : n;
        const double* Sbase = pend ? fc + (long)n_c * CS + n_r + n_c : a.Sall + chain * mT * mT;
        const __amdgpu_buffer_rsrc_t Sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Sbase), 0, ((n - 1) * ldS + n) * 8, 0x00020000);
        unsigned vo[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) vo[v] = (unsigned)(((4 * v + lr) * ldS + lc) * 8);
        // the chain's cache entry through a descriptor too: a lane that has nothing to store names an offset beyond its end - the
        // store is dropped without traffic and WITHOUT a branch (behind a store under an `if` hipcc's wait insertion cannot count the
        // vector-memory operations in flight any more and turns every later wait for a load into s_waitcnt vmcnt(0))
        const __amdgpu_buffer_rsrc_t Fr = __builtin_amdgcn_make_buffer_rsrc(fc, 0, (int)(a.fc_stride * 8), 0x00020000);
        unsigned so[4];                                           // row 4 v + lr of a 16-row group, column lc
#pragma unroll
        for (int v = 0; v < 4; ++v) so[v] = (unsigned)((((long)(n_c + 4 * v + lr)) * CS + n_r + n_c + lc) * 8);
        // tile (k, j), register v, this lane: element (16 k + 4 v + lr, 16 j + lc); the identity beyond n
        // (the launcher picks NTL with 16 (NTL - 2) < n <= 16 NTL: only tile indices NTL - 2 and NTL - 1 can reach beyond n - every
        // other tile needs no mask at all; hipcc hoists every lane mask of the unrolled kernel to its top and spills them)
        auto load_tile = [&](auto kc, auto jcn) -> jc_d4 {
            constexpr int k = decltype(kc)::value, j = decltype(jcn)::value;
            jc_d4 t;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                t[v] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Sr, vo[v], 128 * k * ldS + 128 * j, 0));
            if constexpr (k == j) {
                if (pend) {                                       // (uniform) + noise of the row's task on the diagonal
                    const int tk = a.h_slots[n_c + min(16 * k + lc, n - 1)] % 3;
                    const double nz = (tk == 0) ? a.gp.noise[0] : ((tk == 1) ? a.gp.noise[1] : a.gp.noise[2]);
#pragma unroll
                    for (int v = 0; v < 4; ++v) t[v] += (4 * v + lr == lc) ? nz : 0.0;
                }
            }
            if constexpr (j >= NTL - 2) {
                const bool cin = 16 * j + lc < n;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double val = cin ? t[v] : 0.0;
                    if constexpr (k == j) val = (4 * v + lr == lc && !cin) ? 1.0 : val;      // the identity beyond n
                    t[v] = val;
                }
            }
            return t;
        };
        // Left-looking over tile rows: row k is R_kj = V_k (A_kj - sum_{p<k} R_pk^T R_pj), j >= k.  Alive at step k: the finished tiles
        // R_pj with j >= k (column k's tiles leave for the cache at the end of the step) and the row in progress - at most
        // (k + 1)(NTL - k) <= 20 tiles = 160 registers: two waves per SIMD, one's diagonal step (VALU / LDS) under the other's MFMAs.
        jc_d4 U[NTT];
        jc_d4 An[NTL];                                            // the A tiles of the next row (requested behind the panel products)
        jc_for<0, NTL>([&](auto jcn) { An[decltype(jcn)::value] = load_tile(std::integral_constant<int, 0>{}, jcn); });
        bool bad = false;
#ifdef GPMPC_PHASE_TIMERS
        long long jph[5] = {0, 0, 0, 0, 0};
        long long jt = __builtin_readcyclecounter();
#endif
        jc_for<0, NTL>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            jc_d4 acc[NTL];
            jc_for<k, NTL>([&](auto jcn) { acc[decltype(jcn)::value] = An[decltype(jcn)::value]; });
            // the diagonal tile's products FIRST: the matrix pipe completes in order, so the VALU work on acc[k] below starts when ITS
            // products are done and runs beside the products of the row's other tiles
            jc_d4 Xn[k > 0 ? k : 1];
            jc_for<0, k>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                Xn[p] = -U[jc_idx(p, k, NTL)];
                acc[k] = jc_nat(Xn[p], U[jc_idx(p, k, NTL)], acc[k]);
            });
            jc_for<k + 1, NTL>([&](auto jcn) {
                constexpr int j = decltype(jcn)::value;
                jc_for<0, k>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    acc[j] = jc_nat(Xn[p], U[jc_idx(p, j, NTL)], acc[j]);
                });
            });
            JCPH(1);
            // ---- the diagonal tile in "lane = row" form (every DPP row of 16 lanes holds a copy and does the same work) ----------
            double mydinv;
            jc_d4 V;
            jc_diag_tile(blk, vt, acc[k], lr, lc, bad, mydinv, U[jc_idx(k, k, NTL)], V);
            // (1 / diag: an unconditional store as well; its slot sits behind the rows of the cache entry)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(jc_u2, mydinv), Fr,
                                                  (lr == 0 && 16 * k + lc < n) ? (unsigned)(((long)a.fc_cap * CS + n_c + 16 * k + lc) * 8) : 0x7ffff000u, 0, 0);
            JCPH(2);
            // ---- the rest of the row on the matrix pipe; the next row's A tiles are requested behind it ------------------------
            jc_for<k + 1, NTL>([&](auto jcn) {
                constexpr int j = decltype(jcn)::value;
                const jc_d4 zero = {0.0, 0.0, 0.0, 0.0};
                U[jc_idx(k, j, NTL)] = jc_nat(V, acc[j], zero);
            });
            if constexpr (k + 1 < NTL)
                jc_for<k + 1, NTL>([&](auto jcn) { An[decltype(jcn)::value] = load_tile(std::integral_constant<int, k + 1>{}, jcn); });
            JCPH(3);
            // ---- column k is final: the factor's rows are R's columns - its k + 1 tiles transposed through LDS (one buffer each, one
            // hand-over for all of them), 16 doubles = 128 bytes per row segment -------------------------------------------------------
            jc_sync();
            jc_for<0, k + 1>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
#pragma unroll
                for (int v = 0; v < 4; ++v) vt[p * TS + (4 * v + lr) * JC_LD + lc] = U[jc_idx(p, k, NTL)][v];
            });
            jc_sync();
            jc_for<0, k + 1>([&](auto pc) {
                constexpr int p = decltype(pc)::value;            // tile (p, k): L[16 k + cc][16 p + rr] = R_pk[rr][cc]
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double val = vt[p * TS + lc * JC_LD + 4 * v + lr];
                    bool ok = true;
                    if constexpr (k >= NTL - 2) ok = 16 * k + 4 * v + lr < n;
                    if constexpr (p == k) ok = ok && (lc <= 4 * v + lr);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(jc_u2, val), Fr, ok ? so[v] : 0x7ffff000u,
                                                          (16 * k * CS + 16 * p) * 8, 0);
                }
            });
            JCPH(4);
        });
#ifdef GPMPC_PHASE_TIMERS
        if (chain == a.chain0 && lane == 0)
            for (int i = 0; i < 5; ++i) g_jc_phase[i] = jph[i];
#endif
        const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) a.info[chain] = (a.info_in ? a.info[chain] : 0) | (any_bad ? GPMPC_INFO_TRAIN_CHOL_FAIL : 0);
    }
}

bool joint_chol_mfma_eligible(int n_new) { return n_new >= 1 && n_new <= 128; }

// the new hallucinated rows of the chains [a.chain0, a.chain1) against the new columns: Cholesky of the Schur complement in a.Sall
int joint_chol_mfma_launch(const JointArgs& a, hipStream_t st) {
    const int n = a.n_ho - a.n_c;
    if (!joint_chol_mfma_eligible(n)) return fail(GPMPC_E_UNSUPPORTED, "joint_chol_mfma_kernel: 1..128 new rows");
    const long nch = a.chain1 - a.chain0;
    const dim3 g((unsigned)nch), b(64);
    const int ntl = (n + 15) / 16;
    if (ntl <= 2) hipLaunchKernelGGL(joint_chol_mfma_kernel<2>, g, b, 0, st, a);
    else if (ntl <= 4) hipLaunchKernelGGL(joint_chol_mfma_kernel<4>, g, b, 0, st, a);
    else if (ntl <= 6) hipLaunchKernelGGL(joint_chol_mfma_kernel<6>, g, b, 0, st, a);
    else hipLaunchKernelGGL(joint_chol_mfma_kernel<8>, g, b, 0, st, a);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// joint_tail_mfma_kernel: the TAIL of a joint draw - the root of the posterior covariance with gpytorch's jitter-on-failure chain
// (reference src/agent.py:641), y = mean + R z, variance floor, variance-is-zero rule and beta clip (:646-708) - one
// WAVE per chain on the same tile machinery.  joint_kernel's TAIL phase walked S with one label row per thread and 16-column blocks
// behind workgroup barriers: ~130 us per launch at the configs[4] shard for an attempt that fails around pivot 50 (the car as
// shipped), ~120 us of the pendulum's 300 us draw for its one-and-a-half factorisations.
//   * an attempt is the left-looking tile Cholesky of joint_chol_mfma_kernel on S + jit I, abandoned at the first diagonal tile with a
//     non-positive pivot; a retry whose jitter would not change one diagonal entry of the columns walked is counted, not run (as
//     joint_kernel does: the shipped car's 1e-20 .. 1e-18 against variances of 1e-4 .. 1);
//   * R never leaves the registers: column k of R (tiles R_pk, p <= k) is final at the end of step k and contributes
//     (R^T z)[16 k + c] = sum_p sum_r R_pk[r][c] z[16 p + r] right there (four values per lane and tile, two cross-row exchanges per column).
// S may have only its lower triangle valid (joint_kernel's layout) or both (joint_test_mfma_kernel's): element (r, c) is read at
// min(r, c) * mT + max(r, c) - in the column-major lower storage that IS the valid copy.
template <int NTL>
__global__ __launch_bounds__(64, (NTL >= 6) ? 1 : 2) void joint_tail_mfma_kernel(const JointArgs a) {
    constexpr int NTT = NTL * (NTL + 1) / 2;
    constexpr int TS = 16 * JC_LD;
    constexpr int T = 3;
    __shared__ __attribute__((aligned(16))) double blk[TS];
    __shared__ __attribute__((aligned(16))) double vt[TS];
    __shared__ double zs[NTL * 16];                               // base samples
    __shared__ double ys[NTL * 16];                               // R^T z
    const GpParams& gp = a.gp;
    const int lane = threadIdx.x, lr = lane >> 4, lc = lane & 15;
    const int m = a.m, n = m * gp.T;
    const long chain = a.chain0 + blockIdx.x;
    // the covariance through its view (JointArgs::Sv*): the chain's S buffer, or the pending block of the factor cache (leading dimension CS)
    const int ld = a.Sv_ld;
    const double* Sm = a.Sv + (chain - a.Sv_chain_base) * a.Sv_cs;
    const __amdgpu_buffer_rsrc_t Sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Sm), 0, ((n - 1) * ld + n) * 8, 0x00020000);
    for (int t = lane; t < NTL * 16; t += 64) zs[t] = (t < n) ? a.z[chain * (long)n + t] : 0.0;
    // tile (k, j), register v, this lane: element (16 k + 4 v + lr, 16 j + lc) + jit on the diagonal; the identity beyond n
    auto load_tile = [&](auto kc, auto jcn, double jit) -> jc_d4 {
        constexpr int k = decltype(kc)::value, j = decltype(jcn)::value;
        jc_d4 t;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 16 * k + 4 * v + lr, c = 16 * j + lc;
            // (off-diagonal tiles: r < c always; diagonal tiles: the lane's entry may lie below the diagonal - its mirror image is read)
            const int lo = (k == j) ? min(r, c) : r, hi = (k == j) ? max(r, c) : c;
            t[v] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Sr, (unsigned)((lo * ld + hi) * 8), 0, 0));
        }
        if constexpr (j >= NTL - 2) {
            const bool cin = 16 * j + lc < n;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bool rin = (k >= NTL - 2) ? (16 * k + 4 * v + lr < n) : true;
                double val = (cin && rin) ? t[v] : 0.0;
                if constexpr (k == j) val = (4 * v + lr == lc && !cin) ? 1.0 : val;
                t[v] = val;
            }
        }
        if constexpr (k == j) {
#pragma unroll
            for (int v = 0; v < 4; ++v) t[v] += (4 * v + lr == lc && 16 * k + lc < n) ? jit : 0.0;
        }
        return t;
    };
    int level = 0, info_acc = 0;
    bool rooted = false, abandoned = false;
    double jit_total = 0.0;
    jc_sync();
#pragma unroll 1
    while (!rooted) {
        // abandon_root (GPMPC_ROOT_AUTO): once ANY chain of the batch has failed every retry the eigendecomposition root redraws the WHOLE
        // batch (A.7 step 4) - a chain that sees the flag at the head of an attempt skips it (its sample and info word are the eigh kernel's
        // either way; the variance / clip part below still runs).  On the shipped car every chain fails: the launch's first round of waves
        // raises the flag, the other rounds skip their attempts.
        if (a.abandon_root && __builtin_amdgcn_readfirstlane(*(volatile int*)a.any_fail) != 0) {
            abandoned = true;
            level = 3;                                            // what the batch reports after the redraw
            break;
        }
        // ---- one attempt ------------------------------------------------------------------------------------------------------------
        jc_d4 U[NTT];
        bool failed = false;                                      // uniform
        int c_fail = 0;
        // (an early-exit chain: step k + 1 stands INSIDE the success branch of step k - with the steps side by side under `if (!failed)`
        // every tile is alive across every join and the kernel spills a third of its registers)
        auto step = [&](auto self, auto kc) -> void {
            constexpr int k = decltype(kc)::value;
            jc_d4 acc[NTL];
            jc_for<k, NTL>([&](auto jcn) { acc[decltype(jcn)::value] = load_tile(kc, jcn, jit_total); });
            jc_d4 Xn[k > 0 ? k : 1];
            jc_for<0, k>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                Xn[p] = -U[jc_idx(p, k, NTL)];
                acc[k] = jc_nat(Xn[p], U[jc_idx(p, k, NTL)], acc[k]);
            });
            jc_for<k + 1, NTL>([&](auto jcn) {
                constexpr int j = decltype(jcn)::value;
                jc_for<0, k>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    acc[j] = jc_nat(Xn[p], U[jc_idx(p, j, NTL)], acc[j]);
                });
            });
            bool bad = false;
            double mydinv;
            jc_d4 V;
            jc_diag_tile(blk, vt, acc[k], lr, lc, bad, mydinv, U[jc_idx(k, k, NTL)], V);
            if (__builtin_amdgcn_ballot_w64(bad && (16 * k + lc < n)) != 0) {     // (uniform)
                failed = true;
                c_fail = min(16 * (k + 1), n);
                return;
            }
            jc_for<k + 1, NTL>([&](auto jcn) {
                constexpr int j = decltype(jcn)::value;
                const jc_d4 zero = {0.0, 0.0, 0.0, 0.0};
                U[jc_idx(k, j, NTL)] = jc_nat(V, acc[j], zero);
            });
            // column k of R is final: (R^T z)[16 k + lc] = sum_p sum_v sum_lr R_pk[4 v + lr][lc] z[16 p + 4 v + lr]
            double sacc = 0.0;
            jc_for<0, k + 1>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
#pragma unroll
                for (int v = 0; v < 4; ++v) sacc = fma(U[jc_idx(p, k, NTL)][v], zs[16 * p + 4 * v + lr], sacc);
            });
            sacc += __shfl_xor(sacc, 16, 64);
            sacc += __shfl_xor(sacc, 32, 64);
            if (lr == 0) ys[16 * k + lc] = sacc;
            if constexpr (k + 1 < NTL) self(self, std::integral_constant<int, k + 1>{});
        };
        step(step, std::integral_constant<int, 0>{});
        if (!failed) {
            rooted = true;
        } else {
            // a retry whose jitter does not change ONE diagonal entry of the columns walked repeats the failed attempt operation for
            // operation: it is counted, not run (total jitter after retry i is jitter * 10^i, accumulated incrementally like the library)
            bool identical = true;
            while (identical) {
                if (level == 3) break;
                const double jn = gp.jitter * ((level == 0) ? 1.0 : (level == 1) ? 10.0 : 100.0);
                const double jp = (level == 0) ? 0.0 : gp.jitter * ((level == 1) ? 1.0 : 10.0);
                const double jit_next = jit_total + (jn - jp);
                bool same = true;
                for (int t1 = lane; t1 < c_fail; t1 += 64) {
                    const double d = Sm[(long)t1 * ld + t1];
                    same = same && ((d + jit_next) == (d + jit_total));
                }
                identical = __builtin_amdgcn_ballot_w64(!same) == 0;
                jit_total = jit_next;
                ++level;
            }
            if (identical && level == 3) break;
        }
    }
    info_acc |= (level << 1);
    if (!rooted) {
        info_acc |= GPMPC_INFO_ROOT_FAIL;
        if (lane == 0 && !abandoned) atomicOr(a.any_fail, 1);
    }
    jc_sync();
    // ---- sample + post-processing (reference src/agent.py:641-708): one lane per test point --------------------------------------------
    for (int j = lane; j < m; j += 64) {
        double vv[T], mm[T], yy[T];
        bool all_zero = (a.var_zero_thr >= 0.0);
#pragma unroll
        for (int b = 0; b < T; ++b) {
            const int tau = j * T + b;
            double v = Sm[(long)tau * ld + tau];
            if (v < gp.var_floor) {
                v = gp.var_floor;
                info_acc |= GPMPC_INFO_VAR_CLAMPED;
            }
            vv[b] = v;
            mm[b] = a.mean[chain * (long)n + tau];
            all_zero = all_zero && (v <= a.var_zero_thr);
            yy[b] = rooted ? ys[tau] + mm[b] : __builtin_nan("");
        }
#pragma unroll
        for (int b = 0; b < T; ++b) {
            double yb = all_zero ? mm[b] : yy[b];
            if (a.apply_clip) {
                const double sd = a.beta * sqrt(vv[b]);
                yb = fmin(fmax(yb, mm[b] - sd), mm[b] + sd);
            }
            const long off = chain * (long)n + j * T + b;
            a.var[off] = vv[b];
            a.y[off] = yb;
        }
    }
    if (a.covar) {                                                // (debug / tests) the covariance, both triangles from the valid one
        double* Cv = a.covar + chain * (long)n * n;
        for (int e = lane; e < n * n; e += 64) {
            const int t2 = e / n, t1 = e - t2 * n;
            Cv[(long)t1 * n + t2] = Sm[(long)min(t1, t2) * ld + max(t1, t2)];
        }
    }
    // (the lanes' bits: OR over the wave)
    for (int off = 32; off >= 1; off >>= 1) info_acc |= __shfl_xor(info_acc, off, 64);
    if (lane == 0) a.info[chain] = (a.info_in ? a.info[chain] : 0) | info_acc;
}

bool joint_tail_mfma_eligible(int mT, int T) { return T == 3 && mT >= 2 && mT <= 128; }

int joint_tail_mfma_launch(const JointArgs& a, hipStream_t st) {
    const int n = a.m * a.gp.T;
    if (!joint_tail_mfma_eligible(n, a.gp.T)) return fail(GPMPC_E_UNSUPPORTED, "joint_tail_mfma_kernel: T = 3, 2..128 test slots");
    const long nch = a.chain1 - a.chain0;
    const dim3 g((unsigned)nch), b(64);
    const int ntl = (n + 15) / 16;
    if (ntl <= 2) hipLaunchKernelGGL(joint_tail_mfma_kernel<2>, g, b, 0, st, a);
    else if (ntl <= 4) hipLaunchKernelGGL(joint_tail_mfma_kernel<4>, g, b, 0, st, a);
    else if (ntl <= 6) hipLaunchKernelGGL(joint_tail_mfma_kernel<6>, g, b, 0, st, a);
    else hipLaunchKernelGGL(joint_tail_mfma_kernel<8>, g, b, 0, st, a);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// joint_real_mfma_kernel: columns conditioned on the REAL data alone, one WAVE per chain on the same tile machinery.  Two uses:
//   * JOINT_MFMA_FACTOR - the second SQP iteration of every MPC step: the hallucinated set was reset (reference src/agent.py:261-272),
//     the n_ho slots of the first iteration's points are all new and only meet the real columns.  X = L_rr^-1 K_r,new is their block of
//     the factor against the real columns (X^T into the cache rows 0 .. n_ho - 1) and S = K_nn - X^T X their Schur complement before the
//     likelihood noise (into the rows' diagonal block): exactly what a draw with GPMPC_PENDING_WRITE leaves behind, so the factor
//     extension is this launch + joint_chol_mfma_kernel in its pending form.  (joint_test_mfma_kernel's factor mode did that in 0.41
//     ms at the configs[4] shard: eight waves and 158 KB of LDS per chain, one chain per CU, for a substitution over three slot tiles.)
//   * JOINT_MFMA_TEST - a draw with NO hallucinated slot (the first SQP iteration of the first MPC step; reference src/agent.py:629-641
//     on the real-data model): mean = X^T w_r and S = K** - X^T X (both triangles, into a.Sall) for the test slots; the tail follows.
// The real block's factor is the plan's (shared by all chains of an output): blocked substitution over its 16 x 16 tiles - the A operands
// are tiles of L and of the plan's inverse as they lie in memory (nat(X, Y) = X^T Y) -, the kernel entries are formed in the D layout they
// are consumed in.
// dynamic LDS: the RBF values of every (column point, column point) and (real point, column point) pair - one exponential per pair
// of POINTS, formed in a pass of its own; the tiles' entries (nine per pair of full points) take theirs from here
template <int NTL, int NQ>
__global__ __launch_bounds__(64, 1) void joint_real_mfma_kernel(const JointArgs a) {
    constexpr int D = 2, T = 3, NC = NTL * 16, NR = NQ * 16;
    extern __shared__ __attribute__((aligned(16))) double jr_dyn[];
    __shared__ __attribute__((aligned(16))) double cx[NC][D];     // column slots: input point | one-hot of the tasks 1, 2 | point index
    __shared__ __attribute__((aligned(16))) double cb[NC][2];
    __shared__ int cp[NC];
    __shared__ __attribute__((aligned(16))) double rx[NR][D];     // real slots: the same, and w_r
    __shared__ __attribute__((aligned(16))) double ra[NR][4];     // a0 a1 a2 (all zero beyond n_r: the entry is zero) | w_r
    __shared__ int rp[NR];
    const GpParams& gp = a.gp;
    const int lane = threadIdx.x, lr = lane >> 4, lc = lane & 15;
    const bool test = a.mfma_mode == JOINT_MFMA_TEST;
    const int n_r = gp.n_r, Tr = gp.real_has_grad ? T : 1, CS = a.fc_cs;
    const int n = test ? a.m * T : a.n_ho;                        // columns
    const int P = test ? a.m : a.n_h, R = gp.N_r;                 // column points, real points
    double* knn = jr_dyn;                                         // [P][P]
    double* krn = knn + P * P;                                    // [R][P]
    const long chain = a.chain0 + blockIdx.x;
    const long s = chain / gp.g_ny;
    const int o = (int)(chain - s * gp.g_ny);
    const double il0 = gp.inv_l2[o][0], il1 = gp.inv_l2[o][1], os = gp.os[o];
    const double* LinvT = plan_LinvT(a.plan, gp, o);
    const double* w_r = plan_w(a.plan, gp, o);
    const double* pts = (test ? a.X_s : a.X_h) + chain * (long)P * D;
#ifdef GPMPC_PHASE_TIMERS
    long long jph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long jt = __builtin_readcyclecounter();
#endif
    // The real block by BLOCKED SUBSTITUTION, as joint_test_mfma_kernel does it (a product with the whole explicit inverse loses
    // cond(L_rr) eps where the substitution loses cond(diagonal tile) eps: 1e-9 against 1e-12 of the largest variance on the shipped car):
    //   X_q = Linv_qq (K_q - sum_{p < q} L_qp X_p),   Linv_qq = the diagonal tile of the plan's inverse (= the inverse of L's diagonal tile).
    // A operands, D layout, nat(A, B) = A^T B:  LT(q, q)[r][c] = Linv[16 q + c][16 q + r];  LT(p, q)[r][c] = - L[16 q + c][16 p + r], p < q
    // (zero beyond n_r); requested first: they arrive under the pair pass
    const double* Lrr = plan_L(a.plan, gp, o);
    jc_d4 LT[NQ * (NQ + 1) / 2];
    jc_for<0, NQ>([&](auto q2c) {
        constexpr int q2 = decltype(q2c)::value;
        jc_for<q2, NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = 16 * q2 + 4 * v + lr, c = 16 * q + lc;
                const long rc = min(r, n_r - 1), cc = min(c, n_r - 1);
                const double val = (q == q2) ? LinvT[rc * n_r + cc] : -Lrr[cc * n_r + rc];
                LT[jc_idx(q2, q, NQ)][v] = (r < n_r && c < n_r) ? val : 0.0;
            }
        });
    });
    for (int c = lane; c < NC; c += 64) {
        const int cc = min(c, n - 1);
        const int sl = test ? cc : a.h_slots[cc];
        const int pt = sl / T, tk = sl - pt * T;
        cx[c][0] = pts[pt * D];
        cx[c][1] = pts[pt * D + 1];
        cb[c][0] = (tk == 1) ? 1.0 : 0.0;
        cb[c][1] = (tk == 2) ? 1.0 : 0.0;
        cp[c] = pt;
    }
    for (int i = lane; i < NR; i += 64) {
        const int ii = min(i, n_r - 1), pi = ii / Tr, tk = ii - pi * Tr;
        const bool in = i < n_r;
        rx[i][0] = a.X_r[pi * D];
        rx[i][1] = a.X_r[pi * D + 1];
        ra[i][0] = (in && tk == 0) ? 1.0 : 0.0;
        ra[i][1] = (in && tk == 1) ? 1.0 : 0.0;
        ra[i][2] = (in && tk == 2) ? 1.0 : 0.0;
        ra[i][3] = in ? w_r[i] : 0.0;
        rp[i] = pi;
    }
    JCPH(5);
    // ---- one exponential per pair of points (the points themselves through LDS: [P] column points | [R] real points) ---------------------
    {
        jc_d2* ppt = reinterpret_cast<jc_d2*>(knn + (((P + R) * P + 1) & ~1));      // 16-byte aligned
        for (int i = lane; i < P + R; i += 64) ppt[i] = (i < P) ? jc_d2{pts[i * D], pts[i * D + 1]} : jc_d2{a.X_r[(i - P) * D], a.X_r[(i - P) * D + 1]};
        jc_sync();
        JCPH(6);
        const float invP = 1.0f / (float)P;
        const int tot = (P + R) * P;                              // knn and krn are one table of P + R rows
        // four exponentials in lockstep (expn_neg: a lone wave's dependent FP64 chain costs ~8 cycles per operation, four interleaved ~4)
        for (int e0 = 0; e0 < tot; e0 += 256) {
            double arg[4], ev[4];
            int ee[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + 64 * u + lane, tot - 1);
                int i = (int)(((float)e + 0.5f) * invP);
                i = (i * P > e) ? i - 1 : (((i + 1) * P <= e) ? i + 1 : i);
                const jc_d2 xa = ppt[i], xb = ppt[e - i * P];
                const double r0 = xa.x - xb.x, r1 = xa.y - xb.y;
                arg[u] = -0.5 * (r0 * r0 * il0 + r1 * r1 * il1);
                ee[u] = e;
            }
            expn_neg<4>(arg, ev);
#pragma unroll
            for (int u = 0; u < 4; ++u) knn[ee[u]] = os * ev[u];   // (clamped duplicates write the same value)
        }
    }
    JCPH(0);
    // cov(task a of x, task b of x') from the pair's RBF value k (tasks as one-hot weights, no selects):
    //   fa = a0 - a1 q0 - a2 q1, fb = b0 + b1 q0 + b2 q1, entry = k (fa fb + a1 b1 / l0^2 + a2 b2 / l1^2), q = (x - x') / l^2
    auto entry = [&](double k, double x0, double x1, double a0, double a1, double a2, double y0, double y1, double b0, double b1,
                     double b2) -> double {
        const double q0 = (x0 - y0) * il0, q1 = (x1 - y1) * il1;
        const double fa = fma(-a1, q0, fma(-a2, q1, a0));
        const double fb = fma(b1, q0, fma(b2, q1, b0));
        const double cd = fma(a1 * il0, b1, a2 * il1 * b2);
        return k * fma(fa, fb, cd);
    };
    jc_sync();
    JCPH(1);
    double* fc = a.fcache ? a.fcache + (chain - a.fc_chain_base) * a.fc_stride : nullptr;
    // every store below is unconditional: a lane with nothing to store names an offset beyond its descriptor's end (dropped without
    // traffic) - a store under an `if` is a branch, and the entries of the next tile no longer schedule under this tile's products
    double* Sm = test ? a.Sall + chain * (long)n * n : fc + n_r;  // S: leading dimension n / CS
    const int ldS = test ? n : CS;
    const __amdgpu_buffer_rsrc_t Srs = __builtin_amdgcn_make_buffer_rsrc(Sm, 0, ((n - 1) * ldS + n) * 8, 0x00020000);
    double* Xb = test ? a.mean + chain * (long)n : fc;            // the mean / the cache rows' real columns
    const __amdgpu_buffer_rsrc_t Xrs = __builtin_amdgcn_make_buffer_rsrc(Xb, 0, (test ? n : (n - 1) * CS + n_r) * 8, 0x00020000);
    // ---- X = L_rr^-1 K_r,cols ------------------------------------------------------------------------------------------------------------
    jc_d4 X[NQ][NTL];
    jc_for<0, NQ>([&](auto qc) {
        jc_for<0, NTL>([&](auto jcn) { X[decltype(qc)::value][decltype(jcn)::value] = jc_d4{0.0, 0.0, 0.0, 0.0}; });
    });
    jc_for<0, NQ>([&](auto q2c) {
        constexpr int q2 = decltype(q2c)::value;
        double x0[4], x1[4], a0[4], a1[4], a2[4];
        int kro[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = 16 * q2 + 4 * v + lr;
            const jc_d2 xx = *reinterpret_cast<const jc_d2*>(&rx[i][0]);
            const jc_d4 aa = *reinterpret_cast<const jc_d4*>(&ra[i][0]);
            x0[v] = xx.x, x1[v] = xx.y, a0[v] = aa.x, a1[v] = aa.y, a2[v] = aa.z;
            kro[v] = rp[i] * P;
        }
        jc_for<0, NTL>([&](auto jcn) {
            constexpr int j = decltype(jcn)::value;
            const jc_d2 yy = *reinterpret_cast<const jc_d2*>(&cx[16 * j + lc][0]);
            const jc_d2 bb = *reinterpret_cast<const jc_d2*>(&cb[16 * j + lc][0]);
            const double b0 = 1.0 - bb.x - bb.y;
            const int pc = cp[16 * j + lc];
            jc_d4 Kt;
#pragma unroll
            for (int v = 0; v < 4; ++v) Kt[v] = entry(krn[kro[v] + pc], x0[v], x1[v], a0[v], a1[v], a2[v], yy.x, yy.y, b0, bb.x, bb.y);
            // X[q2][j] holds - sum_{p < q2} L_{q2 p} X_p: the tile row is final, then it leaves its share in the rows below
            const jc_d4 zero = {0.0, 0.0, 0.0, 0.0};
            X[q2][j] = jc_nat(LT[jc_idx(q2, q2, NQ)], Kt + X[q2][j], zero);
            jc_for<q2 + 1, NQ>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                X[q][j] = jc_nat(LT[jc_idx(q2, q, NQ)], X[q2][j], X[q][j]);
            });
        });
    });
    JCPH(2);
    jc_for<0, NTL>([&](auto jcn) {
        constexpr int j = decltype(jcn)::value;
        const int c = 16 * j + lc;
        if (!test) {                                              // (uniform) X^T: cache row c, real columns
            jc_for<0, NQ>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                // (the components by name: __builtin_bit_cast of `X[q][j][v]`, a vector ELEMENT, compiled to component 0 for every v)
                const double xs[4] = {X[q][j].x, X[q][j].y, X[q][j].z, X[q][j].w};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * q + 4 * v + lr;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(jc_u2, xs[v]), Xrs,
                                                          (i < n_r && c < n) ? (unsigned)((c * CS + i) * 8) : 0x7ffff000u, 0, 0);
                }
            });
        } else {                                                  // mean = X^T w_r
            double part = 0.0;
            jc_for<0, NQ>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
#pragma unroll
                for (int v = 0; v < 4; ++v) part = fma(X[q][j][v], ra[16 * q + 4 * v + lr][3], part);
            });
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(jc_u2, part), Xrs, (lr == 0 && c < n) ? (unsigned)(c * 8) : 0x7ffff000u, 0, 0);
        }
    });
    JCPH(3);
    // ---- S = K_cc - X^T X, upper tiles (the diagonal ones whole) -------------------------------------------------------------------------
    jc_for<0, NTL>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        double x0[4], x1[4], a0[4], a1[4], a2[4];
        int kro[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int rr = 16 * k + 4 * v + lr;
            const jc_d2 xx = *reinterpret_cast<const jc_d2*>(&cx[rr][0]);
            const jc_d2 aa = *reinterpret_cast<const jc_d2*>(&cb[rr][0]);
            x0[v] = xx.x, x1[v] = xx.y, a1[v] = aa.x, a2[v] = aa.y, a0[v] = 1.0 - aa.x - aa.y;
            kro[v] = cp[rr] * P;
        }
        jc_for<k, NTL>([&](auto jcn) {
            constexpr int j = decltype(jcn)::value;
            const jc_d2 yy = *reinterpret_cast<const jc_d2*>(&cx[16 * j + lc][0]);
            const jc_d2 bb = *reinterpret_cast<const jc_d2*>(&cb[16 * j + lc][0]);
            const double b0 = 1.0 - bb.x - bb.y;
            const int pc = cp[16 * j + lc];
            jc_d4 acc;
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = -entry(knn[kro[v] + pc], x0[v], x1[v], a0[v], a1[v], a2[v], yy.x, yy.y, b0, bb.x, bb.y);
            jc_for<0, NQ>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                acc = jc_nat(X[q][k], X[q][j], acc);
            });
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int rr = 16 * k + 4 * v + lr, c = 16 * j + lc;
                const bool ok = rr < n && c < n;
                const jc_u2 val = __builtin_bit_cast(jc_u2, -acc[v]);
                __builtin_amdgcn_raw_buffer_store_b64(val, Srs, ok ? (unsigned)((rr * ldS + c) * 8) : 0x7ffff000u, 0, 0);
                if constexpr (k != j)                             // joint_eigh_kernel reads whole columns: the mirror image (test use)
                    __builtin_amdgcn_raw_buffer_store_b64(val, Srs, (ok && test) ? (unsigned)((c * ldS + rr) * 8) : 0x7ffff000u, 0, 0);
            }
        });
    });
    JCPH(4);
#ifdef GPMPC_PHASE_TIMERS
    if (chain == a.chain0 && lane == 0)
        for (int i = 0; i < 8; ++i) g_jc_phase[i] = jph[i];      // (read right behind this launch: tools/debug/real_phases.py)
#endif
    if (test && lane == 0) a.info[chain] = 0;
}

// dynamic LDS of joint_real_mfma_kernel: the pair tables of P column points against themselves and against the N_r real points | the points
static size_t joint_real_lds_bytes(int P, int N_r) { return ((size_t)P * P + (size_t)N_r * P + 1 + 2 * (size_t)(P + N_r)) * sizeof(double); }

// P: the points the columns' slots are spread over (test use: m; factor use: n_h).  Four waves per CU: 36 KB of pair tables each
bool joint_real_mfma_eligible(int n_r, int N_r, int ncols, int P, int T, int D) {
    return T == 3 && D == 2 && n_r >= 1 && n_r <= 64 && ncols >= 2 && ncols <= 128 && P >= 1 && joint_real_lds_bytes(P, N_r) <= 36 * 1024;
}

int joint_real_mfma_launch(const JointArgs& a, hipStream_t st) {
    const int n = (a.mfma_mode == JOINT_MFMA_TEST) ? a.m * a.gp.T : a.n_ho;
    const int P = (a.mfma_mode == JOINT_MFMA_TEST) ? a.m : a.n_h;
    if (!joint_real_mfma_eligible(a.gp.n_r, a.gp.N_r, n, P, a.gp.T, a.gp.D) || (a.mfma_mode != JOINT_MFMA_TEST && a.mfma_mode != JOINT_MFMA_FACTOR))
        return fail(GPMPC_E_UNSUPPORTED, "joint_real_mfma_kernel: T = 3, D = 2, <= 64 real slots, 2..128 columns, pair tables <= 36 KB");
    const size_t lds = joint_real_lds_bytes(P, a.gp.N_r);
    const long nch = a.chain1 - a.chain0;
    const dim3 g((unsigned)nch), b(64);
    const int ntl = (n + 15) / 16;
#define GPMPC_REAL_LAUNCH(NQ_)                                                                     \
    do {                                                                                           \
        if (ntl <= 2) hipLaunchKernelGGL((joint_real_mfma_kernel<2, NQ_>), g, b, lds, st, a);        \
        else if (ntl <= 4) hipLaunchKernelGGL((joint_real_mfma_kernel<4, NQ_>), g, b, lds, st, a);   \
        else if (ntl <= 6) hipLaunchKernelGGL((joint_real_mfma_kernel<6, NQ_>), g, b, lds, st, a);   \
        else hipLaunchKernelGGL((joint_real_mfma_kernel<8, NQ_>), g, b, lds, st, a);                 \
    } while (0)
    if (a.gp.n_r <= 48) GPMPC_REAL_LAUNCH(3);
    else GPMPC_REAL_LAUNCH(4);
#undef GPMPC_REAL_LAUNCH
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_joint_chol_phases(long long* out /*[host] 8*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_jc_phase), 8 * sizeof(long long)));
    return GPMPC_OK;
}
