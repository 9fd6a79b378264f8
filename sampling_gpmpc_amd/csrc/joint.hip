// gpmpc_joint_sample: joint posterior draw at m test points per (sample, output) chain (mode "J", gfx950).
//
// One workgroup per chain, one label row per thread, the workgroup just wide enough for the rows (128 / 256 / 512
// threads, 4 waves per SIMD; beyond 512 rows four rows per thread); chains are taken grid-stride so the HBM workspace
// is bounded by the grid.  The factorisation is BLOCKED (NB columns at a time): the pivot-block rows are staged in
// LDS, each thread keeps NB register accumulators per row it owns and streams its own row from HBM/L2 once per block
// (NB FMAs per 8 bytes).  The kernel sits at a register-pressure cliff: at 128 VGPRs any hoisting of the broadcast
// LDS reads spills, hence the data-dependent compiler fences in block_solve and the kernel-evaluation loops.
// Everything is ONE left-looking factorisation over a tall matrix M whose ROWS are label slots and whose COLUMNS
// are the conditioning slots (column-major, leading dimension = rows, so "thread = row" is coalesced):
//
//      rows   : [ hallucinated slots (n_ho) | w (1) | test slots (m*T) ]
//      columns: [ real slots (n_r) | hallucinated slots (n_ho) ]
//
//   * real columns:   M[row, :n_r] = L_rr^-1 k_r(row)          (blocked substitution against the plan's L_rr; w row = w_r)
//   * column n_r+c:   M[row, n_r+c] = (k(row, c) - sum_k M[row,k] M[c,k]) / L_cc   for every row below pivot c
//                     -> hallucinated rows become L_hh, the w row becomes w_h = L_hh^-1 (y_h - L_hr w_r),
//                        test rows become V^T = (L^-1 K_o*)^T             
//   * mean = V^T w, covariance S = K** - V^T V, variance = diag(S) floored (A.8)
//   * root = Cholesky of S with the jitter-on-failure chain (A.7), y = mean + R z, post-processing of sample_gp.
//   * S is left in a per-chain buffer: when any chain of the batch fails all retries, joint_eigh_kernel
//     (joint_eigh.hpp, launched right behind this kernel) redraws the whole batch with the eigendecomposition root.
#include <type_traits>
#include "gpmpc_host.hpp"
#include "joint_args.hpp"
#include "joint_eigh.hpp"

namespace gpmpc {


// Blocked left-looking step shared by the three phases.  For the column block whose pivot rows are
// prow0..prow0+nb-1 and for every row this thread owns (row = tid + rs*256, row >= rlo):
//     acc[rs][q] -= sum_{k < kdone} W[row][k] * W[prow0+q][k]
// The pivot rows are staged KC columns at a time in LDS (broadcast reads), the thread's own row streams from
// HBM/L2 once per block (coalesced: column-major, thread == row): NB FMAs per 8-byte load.
typedef double double2_j __attribute__((ext_vector_type(2)));

#ifndef GPMPC_JOINT_WPE
#define GPMPC_JOINT_WPE 4          // waves per SIMD the one-row-per-thread kernels are compiled for (128 VGPRs)
#endif
__device__ long long g_joint_phase[16];
#ifdef GPMPC_PHASE_TIMERS
#define JPH(idx) do { const long long _n = __builtin_readcyclecounter(); jph[idx] += _n - jt; jt = _n; } while (0)
#else
#define JPH(idx)
#endif

// pivot columns staged per chunk (0: 8 for the 128-thread workgroups, 16 beyond) and the row ring: GPMPC_JOINT_RD batches of
// GPMPC_JOINT_KU columns (0: the chunk in GPMPC_JOINT_RD batches).  tools/joint_sweep.sh, round 2 with the factor cache
// (car k=0 / k=2 / k=3, pendulum k=1; ms): 8,4,2: 2.47 / 7.03 / 11.69, 0.84 - 16,8,2: 2.63 / 6.87 / 10.97, 0.83 -
// 16,4,4: 2.70 / 6.94 / 11.12, 0.84 - 12,4,3: 2.99 / 8.44 / 13.13, 1.03
#ifndef GPMPC_JOINT_KC
#define GPMPC_JOINT_KC 0
#endif
#ifndef GPMPC_JOINT_KU
#define GPMPC_JOINT_KU 0
#endif
#ifndef GPMPC_JOINT_RD
#define GPMPC_JOINT_RD 2
#endif
#ifndef GPMPC_JOINT_WIDE_NB
#define GPMPC_JOINT_WIDE_NB 32     // pivot columns per block of the two-waves-per-SIMD variant for long conditioning sets
#endif
#ifndef GPMPC_JOINT_NB
#define GPMPC_JOINT_NB 16          // pivot columns per block of the launches of <= 256 rows (experiment knob; 24 at three waves per
                                   // SIMD measured: car k=0 / k=2 2.47 / 6.85 ms against 2.47 / 6.97, pendulum k=1 1.12 against 0.83)
#endif
#ifndef GPMPC_JOINT_S_MFMA_MIN
#define GPMPC_JOINT_S_MFMA_MIN 128       // conditioning slots from which S -= V V^T runs on the matrix pipe (1 << 30: never)
#endif
#ifndef GPMPC_JOINT_LDS_BCAST
#define GPMPC_JOINT_LDS_BCAST 0          // 1: the broadcast-ds_read_b128 update (comparison builds, tools/joint_sweep.sh)
#endif
// acc[q] -= w * p[q] for the 16 pivot-row entries of one column, p held ONE ENTRY PER LANE (lane i of every DPP row =
// entry i): v_fmac_f64_dpp row_newbcast:q broadcasts entry q without leaving the VALU.  The broadcast ds_read_b128 form
// (eight 1 KiB reads per 16 FMAs and wave) is bound by the LDS pipe the CU's waves share; this one needs a single
// 512-byte ds_read_b64 per column and wave.
__device__ __forceinline__ void fmac16_dpp(double (&acc)[16], double p, double w) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %16, -%17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %16, -%17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %16, -%17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %16, -%17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %16, -%17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %16, -%17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, %16, -%17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %7, %16, -%17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %8, %16, -%17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %9, %16, -%17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %10, %16, -%17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %11, %16, -%17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %12, %16, -%17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %13, %16, -%17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %14, %16, -%17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %15, %16, -%17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
          "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])
        : "v"(p), "v"(w));
}

// the same for 8 pivot-row entries (lanes 0..7 of every DPP row hold them): the tail of a 24-column block
__device__ __forceinline__ void fmac8_dpp(double (&acc)[8], double p, double w) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %8, -%9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %8, -%9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %8, -%9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %8, -%9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %8, -%9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %8, -%9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, %8, -%9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %7, %8, -%9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(p), "v"(w));
}

template <int NB, int RPT, int KC, int NT>
__device__ __forceinline__ void block_update(const double* __restrict__ W, int ld, int kdone,
                                             const double* __restrict__ P, long p_rs, long p_cs, int nb,
                                             int rlo, int nrow, double (&acc)[RPT][NB], double (*piv)[NB]) {
    // Pivot-row entry (q, k) lives at P[q * p_rs + k * p_cs] (rows of W itself, or rows of the plan's L_rr).
    // piv[kk][q]: the NB pivot-row entries of column k0+kk are contiguous -> NB/2 broadcast ds_read_b128 per column,
    // shared by all RPT rows of the thread (RPT*NB FMAs per NB/2 LDS reads and RPT 8-byte global loads).
    // Both HBM/L2 streams are software-pipelined (their ~1-2 us latency would otherwise be paid per batch):
    //   * the thread's own row entries: KU columns per batch, the next batch is in flight while this one is consumed;
    //   * the pivot rows of the NEXT KC-column chunk are fetched into registers while this chunk is consumed.
    const int tid = threadIdx.x;
    constexpr int KU = (RPT >= 4) ? 2 : (GPMPC_JOINT_KU ? GPMPC_JOINT_KU : KC / GPMPC_JOINT_RD);   // columns per row batch
    constexpr int PV = (NB * KC + NT - 1) / NT;
    constexpr int RD = GPMPC_JOINT_RD;                           // row batches in the ring
    static_assert(KC % (RD * KU) == 0, "the batch ring must turn a whole number of times per pivot chunk");
    bool own[RPT];
    const double* wr[RPT];
#pragma unroll
    for (int rs = 0; rs < RPT; ++rs) {
        const int row = tid + rs * NT;
        own[rs] = (row >= rlo && row < nrow);
        wr[rs] = W + (own[rs] ? row : rlo);
    }
    // waves none of whose rows take part (rows are contiguous per wave) only help staging the pivot rows: the column
    // blocks of the factorisation shrink the active row range, the S / root phases touch the m*T test rows only
    bool any_own = false;
#pragma unroll
    for (int rs = 0; rs < RPT; ++rs) {
        const int w0 = (tid & ~63) + rs * NT;
        any_own = any_own || (w0 + 63 >= rlo && w0 < nrow);
    }
    if (kdone > 0) {
        // ring of RD row batches (KU columns each): RD-1 batches = (RD-1)*KU*RPT 8-byte loads stay in flight per thread
        // while one batch is consumed; the ring runs across the pivot chunks (own-row loads do not depend on them)
        double pv[PV], ring[RD][KU][RPT];
        auto fetch_piv = [&](int k0) {
#pragma unroll
            for (int i = 0; i < PV; ++i) {
                const int e = tid + i * NT, kk = e / NB, q = e - kk * NB;
                pv[i] = (e < NB * KC && q < nb && k0 + kk < kdone) ? P[q * p_rs + (long)(k0 + kk) * p_cs] : 0.0;
            }
        };
        auto fetch_rows = [&](int kbase, double (&m)[KU][RPT]) {
#pragma unroll
            for (int j = 0; j < KU; ++j) {
                const int k = min(kbase + j, kdone - 1);                  // clamped: never consumed beyond kdone
#pragma unroll
                for (int rs = 0; rs < RPT; ++rs) m[j][rs] = wr[rs][(long)k * ld];
            }
        };
        fetch_piv(0);
        if (any_own) {
#pragma unroll
            for (int r = 0; r < RD - 1; ++r) fetch_rows(r * KU, ring[r]);
        }
        for (int k0 = 0; k0 < kdone; k0 += KC) {
            const int kc = min(KC, kdone - k0);
            __syncthreads();                                             // the previous chunk's piv has been consumed
#pragma unroll
            for (int i = 0; i < PV; ++i) {
                const int e = tid + i * NT, kk = e / NB, q = e - kk * NB;
                if (e < NB * KC) piv[kk][q] = pv[i];
            }
            __syncthreads();
            if (k0 + KC < kdone) fetch_piv(k0 + KC);
            for (int kk0 = 0; kk0 < kc && any_own; kk0 += RD * KU) {
#pragma unroll
                for (int r = 0; r < RD; ++r) {
                    const int kb = kk0 + r * KU;                         // this batch; uniform
                    fetch_rows(k0 + kb + (RD - 1) * KU, ring[(r + RD - 1) % RD]);
#pragma unroll
                    for (int j = 0; j < KU; ++j) {
                        if constexpr (NB == 16 && !GPMPC_JOINT_LDS_BCAST) {
                            if (kb + j < kc) {
                                const double pk = piv[kb + j][tid & 15];
#pragma unroll
                                for (int rs = 0; rs < RPT; ++rs) fmac16_dpp(acc[rs], pk, ring[r][j][rs]);
                            }
                        } else if constexpr ((NB == 24 || NB == 32 || NB == 48) && !GPMPC_JOINT_LDS_BCAST) {
                            if (kb + j < kc) {
#pragma unroll
                                for (int h = 0; h < NB / 16; ++h) {
                                    const double pkh = piv[kb + j][16 * h + (tid & 15)];
#pragma unroll
                                    for (int rs = 0; rs < RPT; ++rs)
                                        fmac16_dpp(*reinterpret_cast<double(*)[16]>(&acc[rs][16 * h]), pkh, ring[r][j][rs]);
                                }
                                if constexpr (NB % 16 == 8) {
                                    const double pkt = piv[kb + j][(NB & ~15) + (tid & 7)];
#pragma unroll
                                    for (int rs = 0; rs < RPT; ++rs)
                                        fmac8_dpp(*reinterpret_cast<double(*)[8]>(&acc[rs][NB & ~15]), pkt, ring[r][j][rs]);
                                }
                            }
                        } else
                        if (kb + j < kc) {
#pragma unroll
                            for (int qq = 0; qq < NB / 2; ++qq) {
                                const double2_j pp = *reinterpret_cast<const double2_j*>(&piv[kb + j][2 * qq]);
#pragma unroll
                                for (int rs = 0; rs < RPT; ++rs) {
                                    acc[rs][2 * qq] = fma(-ring[r][j][rs], pp.x, acc[rs][2 * qq]);
                                    acc[rs][2 * qq + 1] = fma(-ring[r][j][rs], pp.y, acc[rs][2 * qq + 1]);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int rs = 0; rs < RPT; ++rs) {
        if (!own[rs]) {
#pragma unroll
            for (int q = 0; q < NB; ++q) acc[rs][q] = 0.0;
        }
    }
}

// Factor the nb x nb diagonal block held (lower part) in blk; LAPACK failure rule; result, 1/diag and flag via LDS.
// Executed by ONE WAVE (lane = row, the row lives in registers, row j is broadcast with v_readlane): a single thread
// walking the block through LDS costs ~70k cycles per block (dependent ~100-cycle LDS reads), this ~5k.
// sacc -= r_k(lane j) * r_k(this lane): lane j's entry comes through the DPP (lane J of the reader's 16-lane row), no SGPR trip
// (s_nop 1: a VGPR written by the VALU may be read through the DPP two wait states later, and hipcc pads nothing inside asm)
template <int J>
__device__ __forceinline__ void fmac_neg_bcast(double& sacc, double rk) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(sacc) : "v"(rk), "n"(J));
}
template <int B, int E, class F>
__device__ __forceinline__ void jstatic_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        jstatic_for<B + 1, E>(f);
    }
}
template <int NB>
__device__ __forceinline__ void block_factor(double (*blk)[NB + 1], double* dinv, int nb, int* flag) {
    const int lane = threadIdx.x & 63;
    const int li = (lane < NB) ? lane : 0;
    double r[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) r[k] = (lane < nb && k <= lane) ? blk[li][k] : 0.0;
    bool bad = false;
    // Row j's finished entries r[0..j-1] live in lane j.  Rows 16 i .. 16 i + 15 share a DPP row, so for the rows below j in
    // j's own 16-row group they are one v_fmac_f64_dpp row_newbcast away (~8 cycles; the v_readlane pair + FMA they replace
    // ~25); only the rows of the NEXT group (NB = 32, j < 16) still need the SGPR broadcast.  Same operations in the same
    // order as before: bit-identical.
    jstatic_for<0, NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double sacc = r[j];
        if constexpr (NB == 16 || j >= 16) {
            jstatic_for<0, j>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                fmac_neg_bcast<(j & 15)>(sacc, r[k]);
            });
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k < j) sacc = fma(-r[k], readlane_f64(r[k], j), sacc);       // row j's finished entries, broadcast
        }
        const double d = readlane_f64(sacc, j);
        if (j < nb) {
            if (!(d > 0.0)) bad = true;
            // (v_rsq_f64 + two Newton steps + a Heron correction: sqrt to ~1 ulp, 1 / sqrt to ~1.5, eleven instructions on the
            // serial spine of the block instead of the ~60 of the IEEE sqrt and divide expansions)
            double inv;
            const double sd = sqrt_rsqrt_fast(d, inv);
            r[j] = (lane == j) ? sd : ((lane > j) ? sacc * inv : 0.0);
            if (lane == j) dinv[j] = inv;
        }
    });
    if (lane < nb) {
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (k <= lane) blk[li][k] = r[k];
    }
    if (lane == 0) *flag = bad ? 1 : 0;
}

// Row-wise forward substitution against the factorised diagonal block: x = (acc - x[:q] . blk[q][:q]) / blk[q][q] for the
// block's nb columns; columns beyond qmax (above the diagonal for a row inside the block) are zero.
template <int NB>
__device__ __forceinline__ void block_solve(const double (&acc)[NB], double (&x)[NB], double (*blk)[NB + 1],
                                            const double* dinv, int nb, int qmax) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double v = acc[q];
#pragma unroll
        for (int e = 0; e < NB; ++e)
            if (e < q) v = fma(-x[e], blk[q][e], v);
        x[q] = (q < nb && q <= qmax) ? v * dinv[q] : 0.0;
        // row q of blk is consumed before row q+1 is read: otherwise all NB(NB-1)/2 broadcast reads are hoisted (spills)
        asm volatile("" : "+v"(x[q]) : : "memory");
    }
}

// S -= V V^T on the FP64 matrix pipe (lower triangle; S column-major mT x mT, V[t][k] at Vt[k * ld + t], k < n_o).
// The VALU form streams every test row once per 16-column block of S with half-rate DPP broadcasts and keeps only the
// waves that own test rows busy; here the workgroup's waves split S into 16-row tile columns (column J together with
// column np-1-J: equal work), a wave holds the <= CH tiles of a column segment in 8 CH accumulator registers and reads the
// column's 16 rows once plus each tile's 16 rows per 4 conditioning slots: ~1.1 fragment loads (512 B) per
// v_mfma_f64_16x16x4_f64 (2048 FLOP).  Fragment layout as in eigh_gram: A[i][k], B[k][j] at lane (i | j) + 16 k,
// D[4 v + (lane >> 4)][lane & 15].  A = the column's rows, B = the tile's rows, so that D's lane index runs along the rows
// of S (coalesced update).
template <int NT, int CH, bool DBUF>
__device__ __forceinline__ void syrk_lower_mfma(const double* __restrict__ Vt, int ld, int n_o, int mT, double* __restrict__ Sm) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int nw = NT / 64;
    const int np = (mT + 15) >> 4;
    const int jl = lane & 15, kr = lane >> 4;
    for (int q = wave; 2 * q < np; q += nw) {
        for (int side = 0; side < 2; ++side) {
            const int J = side ? np - 1 - q : q;
            if (side && J == q) break;                            // odd np: the middle column once
            const int rj = J * 16 + jl;
            const bool vj = rj < mT;
            for (int I0 = J; I0 < np; I0 += CH) {
                // the accumulators START at K** and the products are subtracted from them (A holds -V): where the posterior is
                // nearly degenerate the running value shrinks towards S as it does in the sequential VALU form, instead of
                // forming V V^T ~ K** first and cancelling at the end
                double4_e acc[CH];
                const int nti = min(CH, np - I0);
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int row = (I0 + u) * 16 + jl;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int colS = J * 16 + kr + 4 * v;
                        acc[u][v] = (u < nti && row >= colS && row < mT && colS < mT) ? Sm[(long)colS * mT + row] : 0.0;
                    }
                }
                // fragments of 8 conditioning slots per step, double buffered: the next step's loads are in flight while this
                // step's MFMAs run (one wave per SIMD works here and M streams from HBM/L2: undivided, every step would pay
                // the full load latency)
                auto load_frags = [&](int k0, double (&af)[2], double (&bf)[2][CH]) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = k0 + 4 * h + kr;
                        const bool vk = k < n_o;
                        const double* col = Vt + (long)(vk ? k : 0) * ld;
                        af[h] = (vk && vj) ? -col[rj] : 0.0;
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int ri = (I0 + u) * 16 + jl;
                            bf[h][u] = (u < nti && vk && ri < mT) ? col[ri] : 0.0;
                        }
                    }
                };
                auto run_frags = [&](const double (&af)[2], const double (&bf)[2][CH]) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int u = 0; u < CH; ++u)
                            if (u < nti) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[h], bf[h][u], acc[u], 0, 0, 0);
                };
                if constexpr (DBUF) {
                    double af0[2], bf0[2][CH], af1[2], bf1[2][CH];
                    load_frags(0, af0, bf0);
                    for (int k0 = 0; k0 < n_o; k0 += 16) {
                        load_frags(k0 + 8, af1, bf1);             // beyond n_o: zeros
                        run_frags(af0, bf0);
                        load_frags(k0 + 16, af0, bf0);
                        run_frags(af1, bf1);
                    }
                } else {                                          // 128-register kernels: a second buffer spills
                    for (int k0 = 0; k0 < n_o; k0 += 8) {
                        double af0[2], bf0[2][CH];
                        load_frags(k0, af0, bf0);
                        run_frags(af0, bf0);
                    }
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (u < nti) {
                        const int row = (I0 + u) * 16 + jl;       // D's lane index: the row of S
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int colS = J * 16 + kr + 4 * v;
                            if (row >= colS && row < mT && colS < mT) {
                                Sm[(long)colS * mT + row] = acc[u][v];
                                Sm[(long)row * mT + colS] = acc[u][v];   // mirror: joint_eigh_kernel reads whole columns
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int T, int NB, int RPT, int NT, int WPE>
__global__ __launch_bounds__(NT, WPE) void joint_kernel(const JointArgs a) {
    constexpr int D = 2;
    constexpr int KC = GPMPC_JOINT_KC ? GPMPC_JOINT_KC : ((NT <= 128 || RPT > 1) ? 8 : 16);   // pivot columns staged per chunk
    __shared__ __attribute__((aligned(16))) double piv[KC][NB];
    __shared__ double blk[NB][NB + 1];
    __shared__ double dinv_s[NB];
    __shared__ int s_flag;
    __shared__ int s_abandon;
    __shared__ int s_info;
    __shared__ __attribute__((aligned(16))) double colx[NB][D];   // input point / task / label of the block's NB pivot slots
    __shared__ int colt[NB];
    __shared__ int colsame[NB];                                   // the slot's input point is the previous slot's (one exp for both)
    __shared__ double coly[NB];
    const GpParams& gp = a.gp;
    const int tid = threadIdx.x;
    constexpr int nt = NT;
    const int n_r = gp.n_r, Tr = gp.real_has_grad ? T : 1;
    const int n_ho = a.n_ho, m = a.m, mT = m * T;
    const int n_o = n_r + n_ho;
    const int ld = a.ld;
    // JOINT_PHASE_FACTOR: the hallucinated rows only (no w row, no test rows: joint_test_mfma_kernel forms them)
    // JOINT_PHASE_CHOL: the new rows against the new columns only, starting from the Schur complement in Sall
    const bool ph_chol = a.phase == JOINT_PHASE_CHOL;
    const bool ph_factor = a.phase != JOINT_PHASE_TAIL, ph_test = a.phase == JOINT_PHASE_ALL || a.phase == JOINT_PHASE_HEAD,
               ph_tail = a.phase == JOINT_PHASE_ALL || a.phase == JOINT_PHASE_TAIL;
    const int wrow = n_ho, trow0 = n_ho + 1, nrow = ph_test ? n_ho + 1 + mT : n_ho;

    double* M = a.ws + (long)blockIdx.x * a.ws_chain_stride;     // [n_o][ld]   column-major, thread == row
    double* Rm = M + (long)n_o * ld;                              // [mT][mT]    factor attempts
    double* muv = Rm + (long)mT * mT;                             // [mT]
    double* yv = muv + mT;                                        // [mT]   mean + R z

    for (long chain = a.chain0 + blockIdx.x; chain < a.chain1; chain += gridDim.x) {
        double* Sm = a.Sall + chain * (long)mT * mT;             // [mT][mT]    column-major, lower part valid
        double* fc = a.fcache ? a.fcache + (chain - a.fc_chain_base) * a.fc_stride : nullptr;      // cached factor rows of this chain
        double* fdinv = fc ? fc + (long)a.fc_cap * a.fc_cs : nullptr;
        const int n_c = fc ? a.n_c : 0, CS = a.fc_cs;             // rows < n_c: valid in the cache, not recomputed
        const int rb = n_c;                                       // thread 0 owns row rb: the workgroup spans the rows that are computed
        const bool fill = fc && n_ho <= a.fc_cap;                 // this call's hallucinated rows go into the cache
        const long s = chain / gp.g_ny;
        const int o = (int)(chain - s * gp.g_ny);
        const double* LinvT = plan_LinvT(a.plan, gp, o);
        const double* w_r = plan_w(a.plan, gp, o);
        const double* Xh = a.X_h ? a.X_h + chain * (long)a.n_h * D : nullptr;
        const double* Yh = a.Y_h ? a.Y_h + chain * (long)a.n_h * T : nullptr;
        const double* Xs = a.X_s + chain * (long)m * D;
        double il2[D];
#pragma unroll
        for (int d = 0; d < D; ++d) il2[d] = gp.inv_l2[o][d];
        const double os = gp.os[o];
        int info_acc = 0;
        if (tid == 0) s_info = a.info_in ? a.info[chain] : 0;     // (the bits of an earlier launch of the same call)
        __syncthreads();
#ifdef GPMPC_PHASE_TIMERS
        long long jph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long jroot_blocks = 0, jroot_attempts = 0;             // root phase: column blocks walked / attempts made
        long long jt = __builtin_readcyclecounter();
#endif

        // row descriptor: input point + task of label slot `row` (hallucinated or test; not the w row)
        auto row_point = [&](int row, const double*& xp, int& task) {
            if (row < n_ho) {
                const int sl = a.h_slots[row];
                const int j = sl / T;
                task = sl - j * T;
                xp = Xh + (long)j * D;
            } else {
                const int tau = row - trow0;
                const int j = tau / T;
                task = tau - j * T;
                xp = Xs + (long)j * D;
            }
        };

        // this thread's rows: input point and task, fixed for the whole chain
        double rowx[RPT][D];
        int rowt[RPT];
#pragma unroll
        for (int rs = 0; rs < RPT; ++rs) {
            const int row = rb + tid + rs * nt;
            rowt[rs] = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) rowx[rs][d] = 0.0;
            if (row < nrow && row != wrow) {
                const double* xp;
                row_point(row, xp, rowt[rs]);
#pragma unroll
                for (int d = 0; d < D; ++d) rowx[rs][d] = xp[d];
            }
        }

        // ---- real columns: M[row, :n_r] = L_rr^-1 k_r(row)  (w row = w_r) ------------------------------------------
        // the same blocked substitution as below, against the plan's factor L_rr (row-major) instead of rows of M
        if (ph_factor && !ph_chol) {
            const double* Lrr = plan_L(a.plan, gp, o);
            for (int cb = 0; cb < n_r; cb += NB) {
                const int nb = min(NB, n_r - cb);
                if (tid < NB) {
                    const int i = min(cb + tid, n_r - 1);
                    const int pi = i / Tr;
                    colx[tid][0] = a.X_r[pi * D];
                    colx[tid][1] = a.X_r[pi * D + 1];
                    colt[tid] = i - pi * Tr;
                    coly[tid] = w_r[i];
                    dinv_s[tid] = 1.0 / Lrr[(long)i * n_r + i];
                }
                for (int e = tid; e < NB * NB; e += nt) {
                    const int q = e / NB, c = e - q * NB;
                    if (c < q) blk[q][c] = (q < nb) ? Lrr[(long)(cb + q) * n_r + cb + c] : 0.0;
                }
                __syncthreads();
                double acc[RPT][NB];
#pragma unroll
                for (int rs = 0; rs < RPT; ++rs) {
                    const int row = rb + tid + rs * nt;
                    double qq[D] = {0.0, 0.0}, k = 0.0;                 // one exponential per real input point (its Tr task slots are adjacent)
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        acc[rs][q] = 0.0;
                        if (q < nb && row >= n_c && row < nrow && row != wrow) {
                            if (q == 0 || colt[q] == 0) k = kern_scalar<D>(colx[q], rowx[rs], il2, os, qq);    // r = x_real - x_row
                            acc[rs][q] = kern_entry<D>(qq, k, il2, colt[q], rowt[rs]);
                        }
                        if ((q & 3) == 3) asm volatile("" ::: "memory");
                    }
                }
                block_update<NB, RPT, KC, NT>(M + rb, ld, cb, Lrr + (long)cb * n_r, n_r, 1, nb, 0, nrow - rb, acc, piv);
#pragma unroll
                for (int rs = 0; rs < RPT; ++rs) {
                    const int row = rb + tid + rs * nt;
                    if (row >= n_c && row < nrow) {
                        double x[NB];
                        block_solve<NB>(acc[rs], x, blk, dinv_s, nb, NB);
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q < nb) M[(long)(cb + q) * ld + row] = (row == wrow) ? coly[q] : x[q];
                        if (fill && row < n_ho) {
#pragma unroll
                            for (int q = 0; q < NB; ++q)
                                if (q < nb) fc[(long)row * CS + cb + q] = x[q];
                        }
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        JPH(0);

        // ---- hallucinated columns, NB at a time -------------------------------------------------------------------
        // (the blocks are NB wide from slot 0 and again from slot n_c: a block never straddles the cached / new boundary,
        // so n_c needs no alignment - the cache holds plain factor entries, any partition can read them)
        const int koff = ph_chol ? n_r + n_c : 0;                 // JOINT_PHASE_CHOL: the old columns are already eliminated
        const int n_new = n_ho - n_c;
        for (int c0 = ph_chol ? n_c : 0, nb = 0; c0 < n_ho && ph_factor; c0 += nb) {
            const bool cached = c0 < n_c;                     // uniform: the block's pivot rows and its factorised diagonal block are in the cache
            nb = min(NB, (cached ? n_c : n_ho) - c0);
            if (tid < NB) {                                   // descriptors of the block's pivot slots, shared by all rows
                int tc = 0, same = 0;
                double x0 = 0.0, x1 = 0.0, yl = 0.0;
                if (tid < nb) {
                    const double* xc;
                    row_point(c0 + tid, xc, tc);
                    x0 = xc[0];
                    x1 = xc[1];
                    yl = Yh[a.h_slots[c0 + tid]];
                    same = (tid > 0 && a.h_slots[c0 + tid] / T == a.h_slots[c0 + tid - 1] / T) ? 1 : 0;
                }
                colx[tid][0] = x0;
                colx[tid][1] = x1;
                colt[tid] = tc;
                colsame[tid] = same;
                coly[tid] = yl;
            }
            __syncthreads();
            double acc[RPT][NB];
#pragma unroll
            for (int rs = 0; rs < RPT; ++rs) {
                const int row = rb + tid + rs * nt;
#pragma unroll
                for (int q = 0; q < NB; ++q) acc[rs][q] = 0.0;
                if (ph_chol) {
                    if (row >= c0 && row < nrow) {
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q < nb) acc[rs][q] = Sm[(long)(c0 + q - n_c) * n_new + (row - n_c)];
                    }
                } else if (row >= max(c0, n_c) && row < nrow) {
                    if (row == wrow) {
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q < nb) acc[rs][q] = coly[q];
                    } else {
                        // the T task slots of one input point share the exponential: it is evaluated when the point changes
                        double qq[D] = {0.0, 0.0}, k = 0.0;
#pragma unroll
                        for (int q = 0; q < NB; ++q) {
                            if (q < nb) {
                                const int tc = colt[q];
                                if (q == 0 || !colsame[q]) k = kern_scalar<D>(rowx[rs], colx[q], il2, os, qq);   // r = x_row - x_c
                                double kv = kern_entry<D>(qq, k, il2, rowt[rs], tc);
                                if (row == c0 + q) kv += gp.noise[tc];
                                acc[rs][q] = kv;
                            }
                            if ((q & 3) == 3) asm volatile("" ::: "memory");   // four kernel evaluations in flight (registers)
                        }
                    }
                }
            }
            JPH(1);
            block_update<NB, RPT, KC, NT>(M + rb + (long)koff * ld, ld, n_r + c0 - koff,
                                          cached ? fc + (long)c0 * CS : M + c0 + (long)koff * ld, cached ? CS : 1,
                                          cached ? 1 : ld, nb, cached ? 0 : c0 - rb, nrow - rb, acc, piv);
            __syncthreads();
            JPH(2);
            if (cached) {
                for (int e = tid; e < NB * NB; e += nt) {
                    const int q = e / NB, c = e - q * NB;
                    if (c <= q) blk[q][c] = (q < nb) ? fc[(long)(c0 + q) * CS + n_r + c0 + c] : 0.0;
                }
                if (tid < NB) dinv_s[tid] = (tid < nb) ? fdinv[c0 + tid] : 0.0;
                if (tid == 0) s_flag = 0;
            } else {
#pragma unroll
                for (int rs = 0; rs < RPT; ++rs) {
                    const int row = rb + tid + rs * nt;
                    if (row >= c0 && row < c0 + nb) {
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q <= row - c0) blk[row - c0][q] = acc[rs][q];
                    }
                }
                __syncthreads();
                if (tid < 64) block_factor<NB>(blk, dinv_s, nb, &s_flag);
            }
            __syncthreads();
            JPH(3);
            if (s_flag) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
            if (fill && !cached) {                            // the diagonal block as block_factor left it, and 1/diag
                for (int e = tid; e < NB * NB; e += nt) {
                    const int q = e / NB, c = e - q * NB;
                    if (c <= q && q < nb) fc[(long)(c0 + q) * CS + n_r + c0 + c] = blk[q][c];
                }
                if (tid < nb) fdinv[c0 + tid] = dinv_s[tid];
            }
#pragma unroll
            for (int rs = 0; rs < RPT; ++rs) {
                const int row = rb + tid + rs * nt;
                if (row >= max(c0, n_c) && row < nrow) {
                    double x[NB];
                    block_solve<NB>(acc[rs], x, blk, dinv_s, nb, (row < n_ho) ? row - c0 : NB);
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (q < nb) M[(long)(n_r + c0 + q) * ld + row] = x[q];
                    if (fill && row >= c0 + nb && row < n_ho) {   // rows below the block (its own rows: written above)
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q < nb) fc[(long)row * CS + n_r + c0 + q] = x[q];
                    }
                }
            }
            __syncthreads();
        }

        JPH(4);
        if (!ph_tail && !ph_test) {                               // JOINT_PHASE_FACTOR: the chain's rows are in M and in the cache
#ifdef GPMPC_PHASE_TIMERS
            if (blockIdx.x == 0 && tid == 0)                      // (slots 10..14: the last factor-only launch - FACTOR or CHOL)
                for (int i = 0; i < 5; ++i) g_joint_phase[10 + i] = jph[i];
#endif
            if (info_acc) atomicOr(&s_info, info_acc);
            __syncthreads();
            if (tid == 0) a.info[chain] = s_info;
            __syncthreads();
            continue;
        }
        // ---- posterior mean; covariance S = K** - V V^T (same blocked update, no factor step) ---------------------
        if (ph_test) {   // mu = V^T w.  A dot product per test row: the w row (n_o entries) goes through LDS in chunks, every test-row
            // thread streams its own row with MU_U loads in flight.  (It used to be the blocked update with the w row as the
            // only pivot: NB DPP FMAs per conditioning slot for one useful column - 159 k of 2.7 M cycles per chain at k = 3.)
            // Same operations in the same order as that update: acc = fma(-V[t][k], w[k], acc), k ascending; mu = -acc.
            constexpr int MU_U = (WPE < 4) ? 32 : 16;
            constexpr int MU_CH = KC * NB;                        // doubles of `piv`
            double* wl = &piv[0][0];
            const int tau = tid;                                  // RPT == 1 layout for the test rows: tau < mT <= 256 <= NT ... or strided
            double accm[RPT];
#pragma unroll
            for (int rs = 0; rs < RPT; ++rs) accm[rs] = 0.0;
            (void)tau;
            for (int kb = 0; kb < n_o; kb += MU_CH) {
                const int kn = min(MU_CH, n_o - kb);
                __syncthreads();
                for (int k = tid; k < kn; k += nt) wl[k] = M[(long)(kb + k) * ld + wrow];
                __syncthreads();
#pragma unroll
                for (int rs = 0; rs < RPT; ++rs) {
                    const int t1 = tid + rs * nt;
                    if (t1 < mT) {
                        const double* vr = M + trow0 + t1 + (long)kb * ld;
                        double a = accm[rs];
                        for (int k0 = 0; k0 < kn; k0 += MU_U) {
                            double v[MU_U];
#pragma unroll
                            for (int u = 0; u < MU_U; ++u) v[u] = vr[(long)min(k0 + u, kn - 1) * ld];
#pragma unroll
                            for (int u = 0; u < MU_U; ++u)
                                if (k0 + u < kn) a = fma(-v[u], wl[k0 + u], a);
                        }
                        accm[rs] = a;
                    }
                }
            }
#pragma unroll
            for (int rs = 0; rs < RPT; ++rs) {
                const int t1 = tid + rs * nt;
                if (t1 < mT) muv[t1] = -accm[rs];
            }
            __syncthreads();
        }
        // S = K** - V V^T.  Short conditioning sets: the blocked VALU update, one pass per column block (closed-loop k = 0:
        // 0.84 ms against 0.92 with the matrix-pipe form, whose set-up is per tile); from GPMPC_JOINT_S_MFMA_MIN slots on the
        // kernel entries only are formed here and V V^T comes off them on the matrix pipe (k = 3: 8.07 against 8.35 ms)
        // (the 128-thread kernel only ever sees <= 7 hallucinated slots: it keeps the VALU form and its register allocation)
        const bool s_mfma = NT >= 256 && n_o >= GPMPC_JOINT_S_MFMA_MIN;
        for (int c0 = 0; c0 < mT && ph_test; c0 += NB) {
            const int nb = min(NB, mT - c0);
            double acc[RPT][NB];
#pragma unroll
            for (int rs = 0; rs < RPT; ++rs) {
                const int row = trow0 + tid + rs * nt;                 // test rows only
#pragma unroll
                for (int q = 0; q < NB; ++q) acc[rs][q] = 0.0;
                const int t1 = tid + rs * nt;
                if (t1 >= c0 && t1 < mT) {
                    const int j1 = t1 / T, b1 = t1 - j1 * T;
                    double qq[D] = {0.0, 0.0}, k = 0.0;
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        if (q < nb) {
                            const int t2 = c0 + q, j2 = t2 / T, b2 = t2 - j2 * T;
                            if (q == 0 || b2 == 0) k = kern_scalar<D>(Xs + (long)j1 * D, Xs + (long)j2 * D, il2, os, qq);
                            acc[rs][q] = kern_entry<D>(qq, k, il2, b1, b2);
                        }
                        if ((q & 3) == 3) asm volatile("" ::: "memory");
                    }
                }
                (void)row;
            }
            // rows are offset by trow0 inside M: shift the base pointer so that "row" == test slot index
            if (!s_mfma) block_update<NB, RPT, KC, NT>(M + trow0, ld, n_o, M + trow0 + c0, 1, ld, nb, c0, mT, acc, piv);
#pragma unroll
            for (int rs = 0; rs < RPT; ++rs) {
                const int t1 = tid + rs * nt;
                if (t1 >= c0 && t1 < mT) {
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (q < nb) {
                            // the direct store covers the diagonal and below only; the mirror store is the ONE writer of the upper
                            // triangle (inside a diagonal block both used to write it: a last-bit asymmetric kern_entry would have
                            // made the eigh root's input depend on the race)
                            if (s_mfma || t1 >= c0 + q) Sm[(long)(c0 + q) * mT + t1] = acc[rs][q];
                            // mirror (the final values only): joint_eigh_kernel reads whole columns of S, coalesced
                            if (!s_mfma && t1 > c0 + q) Sm[(long)t1 * mT + c0 + q] = acc[rs][q];
                        }
                }
            }
        }
        __syncthreads();
        if (!ph_test) {                                           // JOINT_PHASE_TAIL: S is in Sm, the mean in a.mean
            for (int t1 = tid; t1 < mT; t1 += nt) muv[t1] = a.mean[chain * (long)mT + t1];
            __syncthreads();
        }
        if constexpr (NT >= 256) {
            if (s_mfma && ph_test) {                              // K** is in Sm: subtract V V^T, all waves of the workgroup
                syrk_lower_mfma<NT, (WPE >= 4) ? 4 : 8, (WPE < 4)>(M + trow0, ld, n_o, mT, Sm);
                __syncthreads();
            }
        }

        JPH(5);
        if (a.phase == JOINT_PHASE_HEAD) {                        // the tail is joint_tail_mfma_kernel's: hand over the mean (S is in Sall)
            for (int t1 = tid; t1 < mT; t1 += nt) a.mean[chain * (long)mT + t1] = muv[t1];
            if (info_acc) atomicOr(&s_info, info_acc);
            __syncthreads();
            if (tid == 0) a.info[chain] = s_info;
            __syncthreads();
            continue;
        }
        // ---- root: blocked Cholesky of S with the jitter-on-failure chain (A.7) -----------------------------------
        int level = 0;
        bool rooted = false;
        double jit_total = 0.0;
        if (mT == 1) {
            if (tid == 0) {
                Rm[0] = sqrt(Sm[0]);
                if (Sm[0] < 0.0) info_acc |= GPMPC_INFO_NEG_1x1;
            }
            rooted = true;
            __syncthreads();
        }
        bool abandoned = false;
        while (!rooted) {
            bool failed = false;
            int c_fail = 0;                                       // columns walked when the attempt failed
#ifdef GPMPC_PHASE_TIMERS
            ++jroot_attempts;
#endif
            for (int c0 = 0; c0 < mT && !failed; c0 += NB) {
                const int nb = min(NB, mT - c0);
                // Another chain of the batch may have failed for good (any_fail): thread 0 reads the flag NOW - a volatile load,
                // past the caches, whose latency hides behind the block's update - and publishes it with the block's own
                // factorisation flag, behind the barrier that is there anyway.  (A separate __syncthreads_or poll cost 70 us
                // per launch at k = 0: more than abandoning saves.)
                int seen = 0;
                if (a.abandon_root && tid == 0) seen = *(volatile int*)a.any_fail;
#ifdef GPMPC_PHASE_TIMERS
                ++jroot_blocks;
#endif
                double acc[RPT][NB];
#pragma unroll
                for (int rs = 0; rs < RPT; ++rs) {
                    const int t1 = tid + rs * nt;
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        acc[rs][q] = 0.0;
                        if (q < nb && t1 >= c0 + q && t1 < mT)
                            acc[rs][q] = Sm[(long)(c0 + q) * mT + t1] + ((t1 == c0 + q) ? jit_total : 0.0);
                    }
                }
                block_update<NB, RPT, KC, NT>(Rm, mT, c0, Rm + c0, 1, mT, nb, c0, mT, acc, piv);
                __syncthreads();
#pragma unroll
                for (int rs = 0; rs < RPT; ++rs) {
                    const int t1 = tid + rs * nt;
                    if (t1 >= c0 && t1 < c0 + nb) {
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q <= t1 - c0) blk[t1 - c0][q] = acc[rs][q];
                    }
                }
                __syncthreads();
                if (tid < 64) block_factor<NB>(blk, dinv_s, nb, &s_flag);
                if (tid == 0) s_abandon = seen;
                __syncthreads();
                if (s_abandon) {                       // uniform: the eigh kernel redraws the whole batch
                    abandoned = true;
                    break;
                }
                if (s_flag) {
                    failed = true;                     // uniform
                    c_fail = c0 + nb;
                } else {
#pragma unroll
                    for (int rs = 0; rs < RPT; ++rs) {
                        const int t1 = tid + rs * nt;
                        if (t1 >= c0 && t1 < mT) {
                            double x[NB];
                            block_solve<NB>(acc[rs], x, blk, dinv_s, nb, t1 - c0);
#pragma unroll
                            for (int q = 0; q < NB; ++q)
                                if (q < nb) Rm[(long)(c0 + q) * mT + t1] = x[q];
                        }
                    }
                }
                __syncthreads();
            }
            if (abandoned) {
                level = 3;                                        // what the batch reports after the redraw (joint_eigh_kernel)
                break;
            }
            if (!failed) {
                rooted = true;
            } else {
                // A retry whose jitter does not change ONE diagonal entry of the columns walked so far (shipped car
                // configuration: Dyn_gp_jitter 1e-20 against variances of 1e-4 .. 1: S_tt + 1e-18 == S_tt in FP64) repeats
                // the failed attempt operation for operation and fails at the same pivot: it is counted, not run.
                bool identical = true;
                while (identical) {
                    if (level == 3) break;
                    // total jitter after retry i is jitter*10^i, accumulated incrementally like the library does
                    const double jn = gp.jitter * ((level == 0) ? 1.0 : (level == 1) ? 10.0 : 100.0);
                    const double jp = (level == 0) ? 0.0 : gp.jitter * ((level == 1) ? 1.0 : 10.0);
                    const double jit_next = jit_total + (jn - jp);
                    bool same = true;
                    for (int t1 = tid; t1 < c_fail; t1 += nt) {
                        const double d = Sm[(long)t1 * mT + t1];
                        same = same && ((d + jit_next) == (d + jit_total));
                    }
                    identical = __syncthreads_and(same ? 1 : 0) != 0;
                    jit_total = jit_next;
                    ++level;
                }
                if (identical && level == 3) break;               // every remaining retry would have failed identically
            }
        }
        info_acc |= (level << 1);
        if (!rooted) {
            info_acc |= GPMPC_INFO_ROOT_FAIL;
            if (tid == 0) atomicOr(a.any_fail, 1);
        }
        __syncthreads();

        JPH(6);
        // ---- sample + post-processing (reference src/agent.py:641-708) --------------------------------------
        const double* zc = a.z + chain * (long)mT;
        // y_raw = mean + R z: thread = test slot, the column sweep in batches of 8 independent (coalesced) loads
        for (int tau = tid; tau < mT; tau += nt) {
            double acc = 0.0;
            if (rooted) {
                for (int cb = 0; cb <= tau; cb += 8) {
                    double r8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) r8[u] = Rm[(long)min(cb + u, mT - 1) * mT + tau];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (cb + u <= tau) acc += r8[u] * zc[cb + u];
                }
            } else {
                acc = __builtin_nan("");
            }
            yv[tau] = acc + muv[tau];
        }
        __syncthreads();
        for (int j = tid; j < m; j += nt) {
            double vv[T], mm[T], yy[T];
            bool all_zero = (a.var_zero_thr >= 0.0);
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const int tau = j * T + b;
                double v = Sm[(long)tau * mT + tau];
                if (v < gp.var_floor) {
                    v = gp.var_floor;
                    info_acc |= GPMPC_INFO_VAR_CLAMPED;
                }
                vv[b] = v;
                mm[b] = muv[tau];
                all_zero = all_zero && (v <= a.var_zero_thr);
                yy[b] = yv[tau];
            }
#pragma unroll
            for (int b = 0; b < T; ++b) {
                double yb = all_zero ? mm[b] : yy[b];
                if (a.apply_clip) {
                    const double sd = a.beta * sqrt(vv[b]);
                    yb = fmin(fmax(yb, mm[b] - sd), mm[b] + sd);
                }
                const long off = chain * (long)mT + j * T + b;
                a.mean[off] = mm[b];
                a.var[off] = vv[b];
                a.y[off] = yb;
            }
        }
        if (a.covar) {
            double* Cv = a.covar + chain * (long)mT * mT;
            for (int e = tid; e < mT * mT; e += nt) {
                const int t2 = e / mT, t1 = e - t2 * mT;
                const double v = (t1 >= t2) ? Sm[e] : Sm[(long)t1 * mT + t2];
                Cv[(long)t1 * mT + t2] = v;
            }
        }
        JPH(7);
#ifdef GPMPC_PHASE_TIMERS
        if (blockIdx.x == 0 && tid == 0) {
            for (int i = 0; i < 8; ++i) g_joint_phase[i] = jph[i];
            g_joint_phase[8] = jroot_blocks;
            g_joint_phase[9] = jroot_attempts;
        }
#endif
        if (info_acc) atomicOr(&s_info, info_acc);
        __syncthreads();
        if (tid == 0) a.info[chain] = s_info;
        __syncthreads();
    }
}

static long joint_chain_doubles(int n_r, int n_ho, int m, int T, int* ld_out) {
    const int mT = m * T;
    int ld = n_ho + 1 + mT;
    ld = (ld + 3) & ~3;
    if (ld_out) *ld_out = ld;
    return (long)(n_r + n_ho) * ld + 1L * mT * mT + 2L * mT + 4;
}

static long joint_grid(long nchains) {
    const long cap = 256L * 16;
    return nchains < cap ? nchains : cap;
}

// workspace layout (doubles): [joint slots | S of every chain | eigh slots | flags | temporary factor cache]
// The temporary factor cache serves the matrix-pipe path of a call WITHOUT a caller-owned cache (or with one that is too small
// for this call's rows): joint_test_mfma_kernel reads the hallucinated rows of the factor row-major, 16-byte aligned - the
// layout of the factor cache - so the factor phase fills tc_slots cache entries inside the workspace and the call proceeds in
// batches of tc_slots chains.
struct JointWs {
    long grid, stride, s_off, egrid, estride, e_off, f_off, total;
    long egrid_narrow, d_off;
    long tc_off, tc_slots, tc_stride;
    int tc_rows, tc_cs;
    long xt_off, xt_slots;                              // X tiles of the TOP launch (split conditioning sets): xt_slots chains
    int ld;
};

static int fc_row_stride(int n_r, int rows) { return (n_r + rows + 1) & ~1; }

// the launch sizes from which the matrix-pipe path (factor phase + joint_test_mfma_kernel + tail) is taken
static int joint_mfma_from() {
    // hallucinated slots from which it is used (0: never).  configs[4] shard (car, Ns = 1024, H = 40, closed-loop points), VALU path
    // against this one: k = 3 (360 slots) 7.15 / 4.53 ms, k = 2 (240) 4.45 / 3.15, k = 1 (120) 2.16 / 2.12 (Ns = 4096: 7.68 / 7.49) - the
    // fixed per-chain phases of joint_test_mfma_kernel (descriptors, tile inversion, kernel entries) only pay behind a
    // substitution of some length; below ~100 slots the one-launch VALU form wins
    static const char* env = getenv("GPMPC_JOINT_MFMA_FROM");
    return env ? atoi(env) : 100;
}
static int g_tail_kernel_force = -1;      // gpmpc_debug_joint_tail_kernel: -1 default (on), 0 / 1 forced
static int g_chol_kernel_force = -1;      // gpmpc_debug_joint_chol_kernel: -1 default (on), 0 / 1 forced (tests, A/B timing)
static int g_real_kernel_force = -1;      // gpmpc_debug_joint_real_kernel: -1 default (on), 0 / 1 forced (tests, A/B timing)
static int g_eigh_narrow_force = -1;      // gpmpc_debug_eigh_narrow: -1 heuristic, 0 / 1 forced (tests)
static int g_joint_path_pin = 0;          // gpmpc_joint_pin_path: 0 auto, 1 VALU path, 2 matrix-pipe path where instantiated
static int g_joint_last_path = 0;
static bool joint_mfma_wanted(int n_ho, int mT) {
    if (g_joint_path_pin == 1) return false;
    if (g_joint_path_pin == 2) return true;
    const int from = joint_mfma_from();
    if (from > 0 && n_ho >= from) return true;
    // Round 6 (the factor rows with nothing cached by joint_real_mfma_kernel, the Cholesky and the tail one wave per chain, S written once): with
    // a WIDE test block the matrix pipe also wins below 100 slots - Ns = 1024, scattered points, VALU / matrix pipe in ms: pendulum H = 30 at
    // 90 slots 0.547 / 0.458 (its closed loop's second SQP iteration: 0.489 -> 0.405), car H = 30 at 90 slots 2.44 / 2.30; with a narrow one it
    // does not (car H = 20 at 60 / 120 slots 1.44 / 1.61 and 1.89 / 2.33; pendulum H = 15 at 90 slots 0.38 / 0.52: the fixed per-chain parts
    // of joint_test_mfma_kernel do not shrink with the columns).  GPMPC_JOINT_MFMA_FROM, when set, is the whole rule.
    static const char* env = getenv("GPMPC_JOINT_MFMA_FROM");
    return !env && n_ho >= 48 && mT >= 84;
}
static bool joint_use_mfma(int n_r, int n_ho, int m, int T) {
    return n_ho >= 1 && joint_mfma_eligible(n_r, n_ho, m * T + 1, T) && joint_mfma_wanted(n_ho, m * T);
}
// conditioning sets beyond one launch of joint_test_mfma_kernel (the 45 + 480 slots of the k = 0 draw of MPC steps >= 1 at
// configs[4]): the test rows in two launches (JOINT_MFMA_TEST_TOP / _BOTTOM); needs a caller-owned factor cache with every row
static bool joint_use_mfma_split(int n_r, int n_ho, int m, int T) {
    static const char* env = getenv("GPMPC_JOINT_MFMA_SPLIT");         // 0: such draws stay on the vector pipe (A/B timing)
    if (env && atoi(env) == 0) return false;
    return joint_mfma_split_eligible(n_r, n_ho, m * T + 1, T) && joint_mfma_wanted(n_ho, m * T);
}

static long eigh_grid(long nchains) {
    static const char* env = getenv("GPMPC_EIGH_SLOTS_PER_CU");       // experiment knob (tools/eigh_sweep.sh)
    const long per_cu = env ? atol(env) : 10;            // one wave per chain, ~8 resident per CU (LDS / VGPR bound)
    const long cap = 256L * (per_cu > 0 ? per_cu : 10);
    return nchains < cap ? nchains : cap;
}

static JointWs joint_ws_layout(int n_r, int n_ho, int m, int T, long nchains) {
    JointWs w;
    const long mT = (long)m * T;
    w.grid = joint_grid(nchains);
    w.stride = joint_chain_doubles(n_r, n_ho, m, T, &w.ld);
    w.s_off = w.grid * w.stride;
    w.egrid = eigh_grid(nchains);
    w.estride = eigh_slot_doubles((int)mT);
    w.e_off = w.s_off + nchains * mT * mT;
    // (the narrow launch's slots - L for ranks <= 40, the log for ranks <= 32 - share the region: 4096 x 105 KB < 2560 x 1.15 MB at
    // m T = 120)
    w.egrid_narrow = nchains < 256L * 16 ? nchains : 256L * 16;
    long eregion = mT > 1 ? w.egrid * w.estride : 0;
    if (mT > 1 && w.egrid_narrow * eigh_narrow_slot_doubles((int)mT) > eregion) eregion = w.egrid_narrow * eigh_narrow_slot_doubles((int)mT);
    w.f_off = w.e_off + eregion;
    w.d_off = w.f_off + 32;                             // deferred chains' ids (ints)
    w.total = w.d_off + (nchains + 1) / 2 + 1;
    w.tc_off = w.tc_slots = w.tc_stride = 0;
    w.tc_rows = w.tc_cs = 0;
    if (n_ho >= 1 && joint_mfma_eligible(n_r, n_ho, m * (int)T + 1, T)) {       // (pin-independent: the workspace serves either path)
        w.tc_rows = (n_ho + 1) & ~1;
        w.tc_cs = fc_row_stride(n_r, w.tc_rows);
        w.tc_stride = (long)w.tc_rows * (w.tc_cs + 1);
        w.tc_slots = nchains < 1024 ? nchains : 1024;
        w.tc_off = (w.total + 1) & ~1L;
        w.total = w.tc_off + w.tc_slots * w.tc_stride;
    }
    w.xt_off = w.xt_slots = 0;
    if (joint_mfma_split_eligible(n_r, n_ho, m * (int)T + 1, T)) {                // (pin-independent, as above)
        w.xt_slots = nchains < 3072 ? nchains : 3072;          // 416 KB per chain: 1.3 GB at most
        w.xt_off = (w.total + 1) & ~1L;
        w.total = w.xt_off + w.xt_slots * JOINT_MFMA_XBUF_DOUBLES;
    }
    return w;
}

}  // namespace gpmpc

using namespace gpmpc;

extern "C" {

int gpmpc_debug_read_joint_phases(long long* out /*[host] 16*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_joint_phase), 16 * sizeof(long long)));
    return GPMPC_OK;
}

int gpmpc_debug_read_eigh_work(unsigned long long* out /*[host] 4*/, int reset) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eigh_work), 4 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero[4] = {0, 0, 0, 0};
        GPMPC_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_eigh_work), zero, sizeof(zero)));
    }
    return GPMPC_OK;
}

// tests / A-B timing: 0 = joint_kernel's own tail, 1 = joint_tail_mfma_kernel, -1 = default (1); returns the previous value
int gpmpc_debug_joint_tail_kernel(int mode) {
    const int prev = g_tail_kernel_force;
    g_tail_kernel_force = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}
// tests / A-B timing: 0 = the matrix-pipe path's Cholesky of the Schur complement by joint_kernel's CHOL phase, 1 = by
// joint_chol_mfma_kernel, -1 = default (1); returns the previous value
int gpmpc_debug_joint_chol_kernel(int mode) {
    const int prev = g_chol_kernel_force;
    g_chol_kernel_force = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}
// tests / A-B timing: joint_real_mfma_kernel (columns conditioned on the real data alone: the factor extension with nothing cached, the
// draw without hallucinated slots) off (0) / on (1), -1 = default (on; GPMPC_JOINT_REAL_KERNEL=0 turns it off); returns the previous value
int gpmpc_debug_joint_real_kernel(int mode) {
    const int prev = g_real_kernel_force;
    g_real_kernel_force = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}
// tests: force the two-launch form of the eigendecomposition root off (0) / on (1), -1 = the rank heuristic; returns the previous value
int gpmpc_debug_eigh_narrow(int mode) {
    const int prev = g_eigh_narrow_force;
    g_eigh_narrow_force = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}
// chains the narrow launches have deferred to the full instantiation since the last reset
long long gpmpc_debug_eigh_deferred(int reset) {
    unsigned long long v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_eigh_deferred), sizeof(v)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_eigh_deferred), &zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return (long long)v;
}

int gpmpc_debug_read_eigh_phases(long long* out /*[host] 8*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eigh_phase), 8 * sizeof(long long)));
    return GPMPC_OK;
}

// which joint path gpmpc_joint_sample takes: 0 = by size (GPMPC_JOINT_MFMA_FROM), 1 = the one-launch VALU path, 2 = the
// matrix-pipe path wherever it is instantiated (else the VALU path)
int gpmpc_joint_pin_path(int32_t path) {
    if (path < 0 || path > 2) return fail(GPMPC_E_ARG, "gpmpc_joint_pin_path: 0 (auto), 1 (VALU) or 2 (matrix pipe)");
    g_joint_path_pin = path;
    return GPMPC_OK;
}
int gpmpc_joint_last_path(void) { return g_joint_last_path; }

// occupancy of the eigh kernel as the runtime computes it (blocks of one wave per CU) for an m*T-slot covariance
int gpmpc_debug_eigh_occupancy(int mT, size_t extra_lds) {
    const int np = (mT + 1) & ~1;
    const int cap = np < EIGH_LDS_RANK ? np : EIGH_LDS_RANK;
    const size_t lds = (size_t)eigh_lds_doubles(mT, cap) * sizeof(double) + extra_lds;
    int nb = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, joint_eigh_kernel<3, 2, GPMPC_EIGH_WPE>, 64, lds) != hipSuccess) return -1;
    return nb;
}

// bytes of the caller-owned factor cache of gpmpc_joint_sample for up to cache_rows hallucinated label rows per chain
size_t gpmpc_joint_cache_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int32_t cache_rows) {
    if (check_gp(gp) != GPMPC_OK || cache_rows < 16 || (cache_rows & 1) || Ns < 1) return 0;
    const size_t cs = (size_t)fc_row_stride(observed_real_slots(gp), cache_rows);
    return align_up((size_t)Ns * gp->g_ny * cache_rows * (cs + 1) * sizeof(double), 256);
}

size_t gpmpc_joint_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int32_t n_ho, int32_t m) {
    if (check_gp(gp) != GPMPC_OK) return 0;
    const JointWs w = joint_ws_layout(observed_real_slots(gp), n_ho, m, gp->T, Ns * gp->g_ny);
    return align_up((size_t)w.total * sizeof(double), 256);
}

static int g_joint_pending_written = 0;    // gpmpc_joint_pending_written

int gpmpc_joint_pending_written(void) { return g_joint_pending_written; }

int gpmpc_joint_sample(const gpmpc_gp_desc_t* gp, const void* plan, const double* X_r, int64_t Ns, int32_t n_h,
                       const double* X_h, const double* Y_h, const int32_t* h_slots, int32_t n_ho, int32_t m,
                       const double* X_s, const double* z, double var_zero_thr, double beta, int32_t apply_clip,
                       double* mean, double* var, double* y, double* covar, double* root, int32_t root_mode,
                       int32_t* info, void* ws, size_t ws_bytes, void* stream, void* factor_cache, int32_t cache_rows,
                       int32_t n_cached) {
    return gpmpc_joint_sample_pending(gp, plan, X_r, Ns, n_h, X_h, Y_h, h_slots, n_ho, m, X_s, z, var_zero_thr, beta, apply_clip, mean,
                                      var, y, covar, root, root_mode, info, ws, ws_bytes, stream, factor_cache, cache_rows, n_cached, 0);
}

int gpmpc_joint_sample_pending(const gpmpc_gp_desc_t* gp, const void* plan, const double* X_r, int64_t Ns, int32_t n_h,
                               const double* X_h, const double* Y_h, const int32_t* h_slots, int32_t n_ho, int32_t m,
                               const double* X_s, const double* z, double var_zero_thr, double beta, int32_t apply_clip,
                               double* mean, double* var, double* y, double* covar, double* root, int32_t root_mode,
                               int32_t* info, void* ws, size_t ws_bytes, void* stream, void* factor_cache, int32_t cache_rows,
                               int32_t n_cached, int32_t pending) {
    g_joint_pending_written = 0;
    if (int rc = check_gp(gp)) return rc;
    if (!plan || !X_r || !X_s || !z || !mean || !var || !y || !info || !ws)
        return fail(GPMPC_E_ARG, "gpmpc_joint_sample: NULL pointer");
    if (Ns < 1 || m < 1 || n_ho < 0 || n_h < 0) return fail(GPMPC_E_ARG, "gpmpc_joint_sample: bad sizes");
    if (n_ho > 0 && (!X_h || !Y_h || !h_slots)) return fail(GPMPC_E_ARG, "gpmpc_joint_sample: hallucinated data missing");
    if (n_ho > n_h * gp->T) return fail(GPMPC_E_ARG, "gpmpc_joint_sample: n_ho > n_h*T");
    if (gp->D != 2) return fail(GPMPC_E_UNSUPPORTED, "only D = 2 is instantiated");
    const int mT = m * gp->T;
    if (mT > 256) return fail(GPMPC_E_UNSUPPORTED, "joint: m*T > 256");
    if (n_ho + 1 + mT > 2048) return fail(GPMPC_E_UNSUPPORTED, "joint: more than 2048 label rows per chain");
    if (root_mode < GPMPC_ROOT_AUTO || root_mode > GPMPC_ROOT_CHOLESKY)
        return fail(GPMPC_E_ARG, "gpmpc_joint_sample: root_mode must be GPMPC_ROOT_AUTO / _EIGH / _CHOLESKY");
    JointArgs a;
    a.gp = make_gp_params(gp);
    a.plan = (const double*)plan;
    a.X_r = X_r;
    a.Ns = Ns;
    a.n_h = n_h;
    a.X_h = X_h;
    a.Y_h = Y_h;
    a.h_slots = h_slots;
    a.n_ho = n_ho;
    a.m = m;
    a.X_s = X_s;
    a.z = z;
    a.var_zero_thr = var_zero_thr;
    a.beta = beta;
    a.apply_clip = apply_clip;
    a.mean = mean;
    a.var = var;
    a.y = y;
    a.covar = covar;
    a.info = (int*)info;
    a.ws = (double*)ws;
    const JointWs w = joint_ws_layout(a.gp.n_r, n_ho, m, gp->T, Ns * gp->g_ny);
    a.ws_chain_stride = w.stride;
    a.ld = w.ld;
    if (ws_bytes < (size_t)w.total * sizeof(double))
        return fail(GPMPC_E_WORKSPACE, "gpmpc_joint_sample: workspace too small");
    a.Sall = (double*)ws + w.s_off;
    a.Sv = a.Sall;
    a.Sv_ld = (int)mT;
    a.Sv_cs = (long)mT * mT;
    a.Sv_chain_base = 0;
    a.any_fail = (int*)((double*)ws + w.f_off);
    if (factor_cache) {
        // an EVEN row count and a 16-byte aligned base: the matrix-pipe path moves cache rows in 16-byte units, and with an odd
        // count it used to fall back to a temporary cache WITHOUT writing the caller's (the next call then read stale rows)
        if (cache_rows < 16 || (cache_rows & 1) || n_cached < 0 || n_cached > n_ho || n_cached > cache_rows ||
            (reinterpret_cast<uintptr_t>(factor_cache) & 15))
            return fail(GPMPC_E_ARG, "gpmpc_joint_sample: factor cache: rows >= 16 and even, 16-byte aligned, 0 <= n_cached <= min(n_ho, rows)");
        a.fcache = (double*)factor_cache;
        a.fc_cap = cache_rows;
        a.fc_cs = fc_row_stride(a.gp.n_r, cache_rows);
        a.fc_stride = (long)cache_rows * (a.fc_cs + 1);
        a.n_c = n_cached;
    } else {
        a.fcache = nullptr;
        a.fc_cap = a.fc_cs = a.n_c = 0;
        a.fc_stride = 0;
    }
    a.fc_chain_base = 0;
    a.pend_write = a.pend_use = 0;
    a.phase = JOINT_PHASE_ALL;
    a.info_in = 0;
    a.chain0 = 0;
    a.chain1 = Ns * gp->g_ny;
    a.mfma_mode = JOINT_MFMA_TEST;
    a.xbuf = nullptr;
    hipStream_t st = (hipStream_t)stream;
    GPMPC_HIP_CHECK(hipMemsetAsync(a.any_fail, 0, 4 * sizeof(int), st));      // any_fail, deferred chains, max rank (joint_eigh_kernel)
    const long nchains = Ns * gp->g_ny;
    // Abandoning pays when the launch needs at least two rounds of the chip (the chains of later rounds skip their root
    // phase): measured on the car's closed loop, Ns = 1024: k = 0 (1.5 rounds) +5 %, k = 1..3 and the 480-slot k = 0
    // (3-6 rounds) -1.5 ... -4.5 %; Ns = 4096: -3 ... -7 % at every k.  GPMPC_JOINT_ABANDON=0 / 1 forces it off / on.
    auto abandon_for = [&](int nrow) -> int {
        static const char* aenv = getenv("GPMPC_JOINT_ABANDON");
        const int nt_ = (nrow <= 128) ? 128 : ((nrow <= 256) ? 256 : ((nrow <= 512) ? 512 : 1024));
        const int wpe_ = (nrow > 128 && nrow <= 256 && n_ho >= 300) ? 2 : 4;
        const double rounds = (double)nchains * nt_ / (256.0 * 256.0 * wpe_);
        const bool on = aenv ? (atoi(aenv) != 0) : (rounds >= 2.0);
        return (root_mode == GPMPC_ROOT_AUTO && mT > 1 && on) ? 1 : 0;
    };
    // one label row per thread and a workgroup just wide enough for the rows (more chains per CU when they are short:
    // iteration 0 of config 5 has 121 rows; iteration 0 of every later MPC step conditions on the previous step's
    // whole hallucinated set - the reference's reset-after-build quirk - i.e. 601 rows at config 5: 1024 threads)
    static const char* penv = getenv("GPMPC_JOINT_LDS_PAD");           // experiment knob: dynamic LDS bytes per workgroup (caps the chains per CU)
    const size_t lds_pad = penv ? (size_t)atol(penv) : 0;
    // long conditioning sets on the 256-thread workgroups: 32-column blocks at two waves per SIMD (256 VGPRs) - half the
    // workspace re-reads; the draw is stream-bound and loses little with fewer chains per CU, but the wider blocks cost more
    // serial work per block, so only from ~300 hallucinated slots on (k=3 scattered points 9.8 against 10.8 ms; k=2 7.2 / 7.0)
    static const char* wenv = getenv("GPMPC_JOINT_WIDE_FROM");        // experiment knob: hallucinated slots from which it is used
    const bool wide = n_ho >= (wenv ? atoi(wenv) : 300);
    // launches joint_kernel for the chains [aa.chain0, aa.chain1) with `nrow` label rows per chain
    auto launch = [&](const JointArgs& aa, int nrow) {
        const long nch = aa.chain1 - aa.chain0;
        const dim3 g((unsigned)(nch < w.grid ? nch : w.grid));
#define GPMPC_JOINT_LAUNCH(TT)                                                                              \
    do {                                                                                                    \
        if (nrow <= 128) hipLaunchKernelGGL((joint_kernel<TT, GPMPC_JOINT_NB, 1, 128, GPMPC_JOINT_WPE>), g, dim3(128), lds_pad, st, aa);      \
        else if (nrow <= 256 && wide) hipLaunchKernelGGL((joint_kernel<TT, GPMPC_JOINT_WIDE_NB, 1, 256, 2>), g, dim3(256), lds_pad, st, aa); \
        else if (nrow <= 256) hipLaunchKernelGGL((joint_kernel<TT, GPMPC_JOINT_NB, 1, 256, GPMPC_JOINT_WPE>), g, dim3(256), lds_pad, st, aa); \
        else if (nrow <= 512) hipLaunchKernelGGL((joint_kernel<TT, 16, 1, 512, GPMPC_JOINT_WPE>), g, dim3(512), 0, st, aa); \
        else if (nrow <= 1024) hipLaunchKernelGGL((joint_kernel<TT, 16, 1, 1024, 4>), g, dim3(1024), 0, st, aa);            \
        else hipLaunchKernelGGL((joint_kernel<TT, 16, 2, 1024, 4>), g, dim3(1024), 0, st, aa);               \
    } while (0)
        if (gp->T == 1) GPMPC_JOINT_LAUNCH(1);
        else GPMPC_JOINT_LAUNCH(3);
#undef GPMPC_JOINT_LAUNCH
    };
    if (gp->T != 1 && gp->T != 3) return fail(GPMPC_E_UNSUPPORTED, "joint: only T = 1 and T = 3 (D = 2) are instantiated");
    // the tail (root with the jitter chain, sample, post-processing) as joint_tail_mfma_kernel (joint_chol.hip, round 6) wherever it is
    // instantiated: T = 3, 2 .. 128 test slots; GPMPC_JOINT_TAIL_KERNEL=0 / gpmpc_debug_joint_tail_kernel(0): joint_kernel's own tail
    static const char* tenv = getenv("GPMPC_JOINT_TAIL_KERNEL");
    const int tforce = g_tail_kernel_force >= 0 ? g_tail_kernel_force : (tenv ? atoi(tenv) : 1);
    const bool tail_kernel = tforce != 0 && joint_tail_mfma_eligible(mT, gp->T);
    // joint_tail_mfma_kernel abandons too (one flag read at the head of an attempt): it pays as soon as the launch has a second round of
    // waves (one wave per chain and SIMD at six tiles and more, two below); GPMPC_JOINT_ABANDON=0 / 1 forces it off / on
    static const char* taenv = getenv("GPMPC_JOINT_ABANDON");
    const int tail_abandon = (root_mode == GPMPC_ROOT_AUTO && mT > 1 &&
                              (taenv ? atoi(taenv) != 0 : nchains > (mT > 80 ? 1024 : 2048))) ? 1 : 0;
    JointArgs sv_args = a;                             // whose Sv* the eigh launch reads (redirected only with the caller's own cache: one batch)
    const bool split = !joint_use_mfma(a.gp.n_r, n_ho, m, gp->T) && joint_use_mfma_split(a.gp.n_r, n_ho, m, gp->T) &&
                       a.fcache && n_ho <= a.fc_cap && (a.fc_cap % 2) == 0;
    if (joint_use_mfma(a.gp.n_r, n_ho, m, gp->T) || split) {
        // The matrix-pipe path: (i) the factor is extended by the rows of the new hallucinated slots - their entries against the
        // old columns and the Schur complement on the matrix pipe (joint_test_mfma_kernel, JOINT_MFMA_FACTOR), the Schur
        // complement's blocked Cholesky by joint_kernel (JOINT_PHASE_CHOL); where that is not instantiated (more than 128 new
        // rows, or more new rows than test slots) joint_kernel's factor phase forms the rows on the vector pipe -, (ii)
        // joint_test_mfma_kernel forms the test rows, the mean and S, (iii) the tail draws.  Without a caller-owned cache that
        // can take this call's rows the factor rows go to a temporary cache inside the workspace, one batch of chains at a time.
        g_joint_last_path = 2;
        const bool own = a.fcache && n_ho <= a.fc_cap && (a.fc_cap % 2) == 0;
        JointArgs b = a;
        if (!own) {
            b.fcache = (double*)ws + w.tc_off;
            b.fc_cap = w.tc_rows;
            b.fc_cs = w.tc_cs;
            b.fc_stride = w.tc_stride;
            b.n_c = 0;
        }
        const int n_new = n_ho - b.n_c;
        static const char* fenv = getenv("GPMPC_JOINT_MFMA_FACTOR");      // 0: the factor phase stays on the vector pipe
        // (with nothing cached - the second SQP iteration - the new rows only meet the real columns: joint_kernel's factor phase does
        // rows and Cholesky in one launch, 1.98 against 2.15 ms per draw at configs[4], k = 1; from 120 cached slots on the matrix pipe
        // wins: 3.2 against 3.9 ms at k = 2)
        // (round 6: with the Schur complement's Cholesky on the matrix pipe too - joint_chol.hip - the factor extension wins with
        // nothing cached as well: 1.91 against 2.03 ms at configs[4], k = 1; GPMPC_JOINT_MFMA_FACTOR_FIRST=0 keeps that draw's factor
        // phase on the vector pipe)
        static const char* f0env = getenv("GPMPC_JOINT_MFMA_FACTOR_FIRST");
        const bool mfma_factor = n_new > 0 && n_new <= mT && (b.n_c > 0 || !(f0env && atoi(f0env) == 0)) &&
                                 joint_mfma_eligible(a.gp.n_r, b.n_c, n_new, gp->T) && !(fenv && atoi(fenv) == 0);
        // pending rows (see JointArgs): used when the caller says so and the shapes allow it; written by the one-launch test mode
        // into the caller's cache when the rows fit
        // (GPMPC_PENDING_USE is a permission: where the shapes do not allow it - or on the VALU path - the rows are recomputed)
        const bool pend_use = (pending & GPMPC_PENDING_USE) && own && b.n_c > 0 && gp->T == 3 && joint_chol_mfma_eligible(n_new);
        const bool pend_write = (pending & GPMPC_PENDING_WRITE) && own && !split && gp->T == 3 && n_ho + mT <= a.fc_cap &&
                                joint_chol_mfma_eligible(mT);
        static const char* renv = getenv("GPMPC_JOINT_REAL_KERNEL");
        const int rforce = g_real_kernel_force >= 0 ? g_real_kernel_force : (renv ? atoi(renv) : 1);
        const bool real_first = rforce != 0 && b.n_c == 0 && n_new == n_ho && joint_real_mfma_eligible(a.gp.n_r, a.gp.N_r, n_ho, n_h, gp->T, gp->D) &&
                                joint_chol_mfma_eligible(n_new) && !(fenv && atoi(fenv) == 0);
        const long step = split ? w.xt_slots : (own ? nchains : w.tc_slots);
        for (long c0 = 0; c0 < nchains; c0 += step) {
            b.chain0 = c0;
            b.chain1 = (c0 + step < nchains) ? c0 + step : nchains;
            b.fc_chain_base = own ? 0 : c0;
            b.info_in = 0;
            b.abandon_root = 0;
            if (n_new > 0 && pend_use) {
                // the caller vouches that the cache rows n_c .. n_ho - 1 hold the previous call's X^T and S (its test points are
                // this call's new slots): the factor extension is the Cholesky of (S + noise) in place, nothing else
                b.pend_use = 1;
                if (int rc = joint_chol_mfma_launch(b, st)) return rc;
                b.pend_use = 0;
                GPMPC_HIP_CHECK(hipGetLastError());
                b.info_in = 1;
            } else if (n_new > 0 && real_first) {
                // nothing cached - the second SQP iteration of an MPC step, right behind the reset - so the new slots only meet the real
                // columns: one wave per chain forms X^T and the Schur complement in the cache (what a draw with pending rows leaves
                // there), joint_chol_mfma_kernel factorises it in place (0.41 + 0.18 -> 0.1 + 0.18 ms at the configs[4] shard)
                b.mfma_mode = JOINT_MFMA_FACTOR;
                if (int rc = joint_real_mfma_launch(b, st)) return rc;
                b.pend_use = 1;
                if (int rc = joint_chol_mfma_launch(b, st)) return rc;
                b.pend_use = 0;
                b.info_in = 1;
            } else if (n_new > 0) {
                if (mfma_factor) {
                    b.mfma_mode = JOINT_MFMA_FACTOR;
                    if (int rc = joint_mfma_launch(b, st)) return rc;
                    b.phase = JOINT_PHASE_CHOL;
                } else {
                    b.phase = JOINT_PHASE_FACTOR;
                }
                // the Schur complement's Cholesky: one wave per chain on the matrix pipe (joint_chol.hip, round 6: 0.45 -> ~0.1 ms
                // per draw at the configs[4] shard); GPMPC_JOINT_CHOL_KERNEL=0 / gpmpc_debug_joint_chol_kernel(0): joint_kernel's CHOL phase
                static const char* cenv = getenv("GPMPC_JOINT_CHOL_KERNEL");
                const int cforce = g_chol_kernel_force >= 0 ? g_chol_kernel_force : (cenv ? atoi(cenv) : 1);
                if (mfma_factor && cforce != 0 && joint_chol_mfma_eligible(n_new)) {
                    if (int rc = joint_chol_mfma_launch(b, st)) return rc;
                } else {
                    launch(b, n_new);
                }
                GPMPC_HIP_CHECK(hipGetLastError());
                b.info_in = 1;
            }
            if (split) {
                b.xbuf = (double*)ws + w.xt_off;
                b.mfma_mode = JOINT_MFMA_TEST_TOP;
                if (int rc = joint_mfma_launch(b, st)) return rc;
                b.mfma_mode = JOINT_MFMA_TEST_BOTTOM;
            } else {
                b.mfma_mode = JOINT_MFMA_TEST;
                b.pend_write = pend_write ? 1 : 0;
                if (pend_write && tail_kernel) {       // S once: the pending block IS the covariance buffer of this draw's tail and eigh root
                    b.Sv = b.fcache + (long)n_ho * b.fc_cs + a.gp.n_r + n_ho;
                    b.Sv_ld = b.fc_cs;
                    b.Sv_cs = b.fc_stride;
                    b.Sv_chain_base = b.fc_chain_base;
                    sv_args = b;
                }
            }
            if (int rc = joint_mfma_launch(b, st)) return rc;
            b.pend_write = 0;
            if (tail_kernel) {
                b.abandon_root = tail_abandon;
                if (int rc = joint_tail_mfma_launch(b, st)) return rc;
            } else {
                b.phase = JOINT_PHASE_TAIL;
                b.abandon_root = abandon_for(mT);
                launch(b, mT);
            }
        }
        g_joint_pending_written = pend_write ? 1 : 0;
    } else {
        // (Measured and dropped: for conditioning sets beyond joint_test_mfma_kernel's 416 slots - k = 0 of the MPC steps after the first,
        // 45 + 480 slots at configs[4] - the factor extension alone on the matrix pipe and the test rows here with every
        // hallucinated row cached: 13.3 ms against 11.2 - the test rows' stream is the critical path of this kernel either way.)
        g_joint_last_path = 1;
        const int nrow = n_ho + 1 + mT - a.n_c;        // rows that are computed (the cached ones have no thread)
        static const char* renv = getenv("GPMPC_JOINT_REAL_KERNEL");
        const int rforce = g_real_kernel_force >= 0 ? g_real_kernel_force : (renv ? atoi(renv) : 1);
        if (n_ho == 0 && tail_kernel && rforce != 0 && g_joint_path_pin != 1 && joint_real_mfma_eligible(a.gp.n_r, a.gp.N_r, mT, m, gp->T, gp->D)) {
            // no hallucinated slot (the first SQP iteration of the first MPC step): the test columns only meet the real data, whose
            // inverse factor all chains of an output share - X = L_rr^-1 K_r*, mean and S one wave per chain on the matrix pipe
            g_joint_last_path = 2;
            a.mfma_mode = JOINT_MFMA_TEST;
            if (int rc = joint_real_mfma_launch(a, st)) return rc;
            a.info_in = 1;
            a.abandon_root = tail_abandon;
            if (int rc = joint_tail_mfma_launch(a, st)) return rc;
        } else if (tail_kernel) {                             // head (factor rows, test rows, mean, S) here, the tail one wave per chain
            a.phase = JOINT_PHASE_HEAD;
            a.abandon_root = 0;
            launch(a, nrow);
            GPMPC_HIP_CHECK(hipGetLastError());
            a.info_in = 1;
            a.abandon_root = tail_abandon;
            if (int rc = joint_tail_mfma_launch(a, st)) return rc;
        } else {
            a.abandon_root = abandon_for(nrow);
            launch(a, nrow);
        }
    }
    GPMPC_HIP_CHECK(hipGetLastError());
    // eigendecomposition root for the whole batch when a chain failed all jitter retries (or on request); the kernel
    // returns at once when the flag is clear
    if (mT > 1 && root_mode != GPMPC_ROOT_CHOLESKY) {
        EighArgs e;
        e.gp = a.gp;
        e.Ns = Ns;
        e.m = m;
        e.z = z;
        e.var_zero_thr = var_zero_thr;
        e.beta = beta;
        e.apply_clip = apply_clip;
        e.mean = mean;
        e.var = var;
        e.y = y;
        e.info = (int*)info;
        e.Sall = sv_args.Sv;
        e.S_ld = sv_args.Sv_ld;
        e.S_cs = sv_args.Sv_cs;
        e.any_fail = a.any_fail;
        e.force = (root_mode == GPMPC_ROOT_EIGH);
        e.ws = (double*)ws + w.e_off;
        e.ws_slot_stride = w.estride;
        e.root = root;
        const bool global_G = (getenv("GPMPC_EIGH_GLOBAL_G") != nullptr);     // test knob: Gram matrix in HBM/L2
        const int np = (mT + 1) & ~1;
        e.lds_cap = global_G ? 0 : (np < EIGH_LDS_RANK ? np : EIGH_LDS_RANK);
        e.tol_mult = 16.0;
        e.gg_off = (long)mT * mT;
        e.rlog_off = e.gg_off + 2 * eigh_packed(np);
        e.pass = EIGH_PASS_ALL;
        e.defer_rank = 0;
        int* flags = a.any_fail;                        // [0] any_fail, [1] deferred chains, [2] max rank of this call; zeroed above
        e.defer_count = flags + 1;
        e.rank_hint = flags + 2;
        e.defer_list = (int*)((double*)ws + w.d_off);
        // Batches of low rank (the closed loop's points: 6..16 of 120) first run the NARROW form - LDS for ranks <= 32, 128
        // registers: 16 chains resident per CU instead of 7 (the kernel is a chain of dependent LDS / L2 round trips per wave) - and
        // the chains it defers (rank > 32) the full form, in a second launch over their list.  Which form a chain takes does not
        // change its result.  The choice follows the largest rank of the LAST call whose counter has arrived (a heuristic for
        // speed only): scattered points (ranks ~50) skip the narrow launch.
        static int* hint_host = nullptr;
        if (!hint_host) {
            GPMPC_HIP_CHECK(hipHostMalloc((void**)&hint_host, sizeof(int), hipHostMallocDefault));
            *hint_host = 0;
        }
        static const char* nenv = getenv("GPMPC_EIGH_NARROW");            // 0 / 1: force the choice (A/B timing)
        const int forced = g_eigh_narrow_force >= 0 ? g_eigh_narrow_force : (nenv ? atoi(nenv) : -1);
        const int last_rank = *(volatile int*)hint_host;
        const bool narrow = !global_G && mT <= 128 && mT > EIGH_NARROW_RANK &&
                            (forced >= 0 ? forced != 0 : last_rank <= EIGH_NARROW_RANK);
        auto launch_eigh = [&](const EighArgs& ea, long grid, int wpe) {
            const size_t lds = (size_t)eigh_lds_doubles(mT, ea.lds_cap) * sizeof(double);
            const dim3 ge((unsigned)grid), be(64);
            if (gp->T == 1) {
                if (mT > 128) hipLaunchKernelGGL((joint_eigh_kernel<1, 4, GPMPC_EIGH_WPE>), ge, be, lds, st, ea);
                else if (wpe == 4) hipLaunchKernelGGL((joint_eigh_kernel<1, 2, GPMPC_EIGH_NARROW_WPE>), ge, be, lds, st, ea);
                else hipLaunchKernelGGL((joint_eigh_kernel<1, 2, GPMPC_EIGH_WPE>), ge, be, lds, st, ea);
            } else {
                if (mT > 128) hipLaunchKernelGGL((joint_eigh_kernel<3, 4, GPMPC_EIGH_WPE>), ge, be, lds, st, ea);
                else if (wpe == 4) hipLaunchKernelGGL((joint_eigh_kernel<3, 2, GPMPC_EIGH_NARROW_WPE>), ge, be, lds, st, ea);
                else hipLaunchKernelGGL((joint_eigh_kernel<3, 2, GPMPC_EIGH_WPE>), ge, be, lds, st, ea);
            }
        };
        if (narrow) {
            EighArgs en = e;
            en.pass = EIGH_PASS_NARROW;
            en.lds_cap = EIGH_NARROW_RANK;
            en.defer_rank = EIGH_NARROW_RANK;
            en.gg_off = 0;                              // (no chain of this launch takes the HBM/L2 form)
            en.rlog_off = (long)mT * (EIGH_NARROW_RANK + EIGH_PB);
            en.ws_slot_stride = eigh_narrow_slot_doubles((int)mT);
            launch_eigh(en, w.egrid_narrow, 4);           // (4: "the narrow instantiation", whatever its occupancy)
            GPMPC_HIP_CHECK(hipGetLastError());
            e.pass = EIGH_PASS_DEFERRED;
        }
        launch_eigh(e, w.egrid, GPMPC_EIGH_WPE);
        GPMPC_HIP_CHECK(hipGetLastError());
        GPMPC_HIP_CHECK(hipMemcpyAsync(hint_host, e.rank_hint, sizeof(int), hipMemcpyDeviceToHost, st));
        GPMPC_HIP_CHECK(hipGetLastError());
    }
    return GPMPC_OK;
}

}  // extern "C"
