// rollout_fast_kernel: the tuned re-conditioned rollout for the BASELINE shapes (T = 3 label slots, n_h <= 128
// appended slots, value-only real labels on an N_r = 36 / 45 grid).  gfx950, wave64.
//
// One wave per (sample, output) chain; a workgroup holds WAVES = G_NY * SPW chains (pendulum1D: 4 independent samples,
// car: the 3 outputs of one sample, which meet at one barrier per step).  At BASELINE configs[1] there is exactly ONE
// wave per SIMD, which issues in order: nothing overlaps unless it is adjacent in the instruction stream.  The kernel
// is therefore organised around (a) instruction count, (b) keeping every wave-wide broadcast on the VALU, and (c)
// running independent dependency chains in lockstep:
//
//   * lane == row of every triangular / rectangular factor block.  A pivot value is broadcast with
//     v_fmac_f64_dpp row_newbcast (a wave is four DPP rows of 16 lanes): the forward substitution works on 16-pivot
//     blocks = DPP rows (dpp_bank0_block), the two mat-vecs against k_r / v_r read their vector from registers that
//     ds_bpermute replicated to all four DPP rows (dpp_matvec).  No v_readlane -> SGPR -> v_fma round trip (~100 cycles
//     per pivot), no LDS broadcast reads (the LDS pipe is shared by the CU's four waves);
//   * registers: the lane's bank-0 row of L_hr (rows 0..63 of the chain's appended slots), the diagonal-block segments
//     of the lane's rows of L'' (dg0 / dg1), 1/L_pp and w_p of the lane's rows, the chain's base samples and the input
//     sequence (lane-indexed, v_readlane per step: no vector-memory load inside the step loop);
//   * LDS: L_rr^-1 rows (shared by the workgroup), the bank-1 rows of L_hr, a v_r buffer for the appending lanes and -
//     when it fits (LHH_LDS) - the chain's L_hh; otherwise L_hh lives in an HBM/L2 workspace;
//   * exponentials, the two 3x3 Choleskys and the nine wave reductions of a step advance in lockstep
//     (gpmpc_device.hpp: expn_neg, chol3_pair_fast, wave_sum9).
//
// L_hh is stored ROW-major (row r = its r entries, 16-byte aligned) and COLUMN-SCALED (L''[r][p] = L[r][p] / L[p][p]):
//   * lane == row reads its own row two pivots per 16-byte load with immediate offsets;
//   * the substitution rhs_r -= L''[r][p] * rhs_p needs no divide; v_p = rhs_p / L_pp is formed once after the loop;
//   * appending a row is one coalesced 8-byte store per lane (lane p owns the new row's entry in column p);
//   * the storage is zero-initialised, so not-yet-appended rows read as zero.
#include "gpmpc_host.hpp"
#include "rollout_args.hpp"

#include <type_traits>

namespace gpmpc {

__device__ long long g_fast_phase_cycles[16];

#ifdef GPMPC_PHASE_TIMERS
#define FPHASE_DECL                                   \
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};       \
    long long tph = __builtin_readcyclecounter()
#define FPHASE(idx)                                              \
    do {                                                         \
        const long long _n = __builtin_readcyclecounter();       \
        ph[idx] += _n - tph;                                     \
        tph = _n;                                                \
    } while (0)
#define FPHASE_STORE                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0)                     \
        for (int i = 0; i < 8; ++i) g_fast_phase_cycles[i] = ph[i]
#else
#define FPHASE_DECL
#define FPHASE(idx)
#define FPHASE_STORE
#endif

typedef double double2_t __attribute__((ext_vector_type(2)));

// row-major, every row start 16-byte aligned: row r holds r entries in a slot of r rounded up to even
__host__ __device__ __forceinline__ int lhh_rowofs(int r) {
    const int h = r >> 1;
    return (r & 1) ? 2 * h * (h + 1) : 2 * h * h;
}
// + slack after the last row: its first pair is the chain's zero pair / the sink of the branch-free append stores (the
// size formula is the one the workspace query has always used: rg = 4 in LDS, GPMPC_FAST_RING_GLOBAL in the workspace)
__host__ __device__ __forceinline__ long lhh_doubles(int nh_max, int rg) {
    const int r = nh_max - 1, cap = r + (r & 1), w = 2 * rg;
    int slack = ((nh_max + w - 1) / w) * w + w - cap;
    slack = (slack < 0) ? 0 : ((slack + 1) & ~1);
    return lhh_rowofs(nh_max) + slack;
}
constexpr int kZeroPage = 128;         // doubles of zeros at the head of the HBM/L2 factor workspace (one row's worth of pairs)
constexpr int kRingLds = 4;
#ifndef GPMPC_FAST_RING_GLOBAL
#define GPMPC_FAST_RING_GLOBAL 8
#endif
constexpr int kRingGlobal = GPMPC_FAST_RING_GLOBAL;

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------------------------
// Forward substitution on the DPP rows (LHH_LDS variant).  A wave is four DPP rows of 16 lanes and lane == factor row,
// so a 16-pivot block of the factor coincides with one DPP row:
//   diagonal block : v_fmac_f64_dpp row_newbcast:i with the running right-hand side ITSELF as the broadcast source and
//                    row_mask = the block's row - pivot i of the block reaches the 15 other lanes of its row without
//                    leaving the VALU (the v_readlane -> SGPR -> v_fma chain costs ~100 cycles per pivot, this ~22);
//   replicate      : the 16 solved values are copied to all four DPP rows with ds_bpermute (6 dwords);
//   off-diagonal   : v_fmac_f64_dpp row_newbcast:i from the replicated register, row_mask = the later rows (bank 0) or
//                    all rows (bank 1): three independent accumulation chains, issue bound.
// Every lane still reads only its own row of L'' (two pivots per ds_read_b128).  Entries at or right of the diagonal are
// replaced by exact zeros by pointing the read at a zero pair (one v_cmp + v_cndmask per PAIR instead of per pivot and
// right-hand side), so the arithmetic per row is the same sequence of FMAs as in the v_readlane form: results are
// bit-identical (tools/ubench/dppsubst.hip).  Hazard: a VGPR written by the VALU may be read through DPP only two wait
// states later; the compiler cannot see into the asm, hence the s_nop 1 at the head of every block of three.
// ---------------------------------------------------------------------------------------------------------------
template <int I, int RM>
__device__ __forceinline__ void fmac3_dpp_self(double (&v)[3], double la) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %0, -%3 row_newbcast:%4 row_mask:%5 bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, -%3 row_newbcast:%4 row_mask:%5 bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, -%3 row_newbcast:%4 row_mask:%5 bank_mask:0xf"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])
        : "v"(la), "n"(I), "n"(RM));
}
// acc[b] (+/-)= R[b]@(lane i of the DPP row) * l  for the three right-hand sides; R holds the same 16 values in every DPP
// row (dpp_replicate), so this is a wave-wide broadcast of pivot i without LDS traffic or SGPR round trip
template <int I, bool NEG>
__device__ __forceinline__ void fmac3_dpp_bcast(double (&acc)[3], const double (&R)[3], double l) {
    if constexpr (NEG) {
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %4, -%3 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %5, -%3 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %6, -%3 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2])
            : "v"(l), "v"(R[0]), "v"(R[1]), "v"(R[2]), "n"(I));
    } else {
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %4, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %5, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %6, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2])
            : "v"(l), "v"(R[0]), "v"(R[1]), "v"(R[2]), "n"(I));
    }
}
// four pivots (I0..I0+3) x three right-hand sides in ONE asm block: the DPP read-after-VALU-write hazard can only
// arise at the head of a block (nothing is scheduled inside it), so one s_nop 1 covers twelve fmacs
#define GPMPC_FMAC12(SRC0, SRC1, SRC2, SIGN, CTRL)                                                     \
    "s_nop 1\n\t"                                                                                      \
    "v_fmac_f64_dpp %0, " SRC0 ", " SIGN "%3 row_newbcast:%7 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %1, " SRC1 ", " SIGN "%3 row_newbcast:%7 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %2, " SRC2 ", " SIGN "%3 row_newbcast:%7 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %0, " SRC0 ", " SIGN "%4 row_newbcast:%8 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %1, " SRC1 ", " SIGN "%4 row_newbcast:%8 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %2, " SRC2 ", " SIGN "%4 row_newbcast:%8 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %0, " SRC0 ", " SIGN "%5 row_newbcast:%9 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %1, " SRC1 ", " SIGN "%5 row_newbcast:%9 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %2, " SRC2 ", " SIGN "%5 row_newbcast:%9 " CTRL "\n\t"                              \
    "v_fmac_f64_dpp %0, " SRC0 ", " SIGN "%6 row_newbcast:%10 " CTRL "\n\t"                             \
    "v_fmac_f64_dpp %1, " SRC1 ", " SIGN "%6 row_newbcast:%10 " CTRL "\n\t"                             \
    "v_fmac_f64_dpp %2, " SRC2 ", " SIGN "%6 row_newbcast:%10 " CTRL
// acc[b] (+/-)= R[b]@(lane I0+j of the DPP row) * l_j, j = 0..3, rows in RM
template <int I0, int RM, bool NEG>
__device__ __forceinline__ void fmac12_dpp_from(double (&acc)[3], const double (&R)[3], double l0, double l1, double l2,
                                                double l3) {
    if constexpr (NEG) {
        asm(GPMPC_FMAC12("%11", "%12", "%13", "-", "row_mask:%14 bank_mask:0xf")
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2])
            : "v"(l0), "v"(l1), "v"(l2), "v"(l3), "n"(I0), "n"(I0 + 1), "n"(I0 + 2), "n"(I0 + 3), "v"(R[0]), "v"(R[1]),
              "v"(R[2]), "n"(RM));
    } else {
        asm(GPMPC_FMAC12("%11", "%12", "%13", "", "row_mask:%14 bank_mask:0xf")
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2])
            : "v"(l0), "v"(l1), "v"(l2), "v"(l3), "n"(I0), "n"(I0 + 1), "n"(I0 + 2), "n"(I0 + 3), "v"(R[0]), "v"(R[1]),
              "v"(R[2]), "n"(RM));
    }
}
__device__ __forceinline__ double bpermute_f64(double v, int addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// acc0 += R0@(lane I of the DPP row) * l ; acc1 += R1@(lane I) * l   (the grid-root products, two vectors per axis)
template <int I>
__device__ __forceinline__ void fmac2_dpp_bcast(double& acc0, double& acc1, double R0, double R1, double l) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %3, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(acc0), "+v"(acc1)
        : "v"(l), "v"(R0), "v"(R1), "n"(I));
}
// P0[lane] = sum_j e0[lane OFS + STRIDE*j] * coef[j], P1 likewise with e1; the sources were replicated block-wise (Rb[blk][0/1])
template <int N, int STRIDE, int OFS, int J = 0>
__device__ __forceinline__ void grid_axis_product(double& P0, double& P1, const double (&Rb)[3][2], const double (&coef)[N]) {
    if constexpr (J < N) {
        constexpr int src = OFS + STRIDE * J;
        fmac2_dpp_bcast<src % 16>(P0, P1, Rb[src / 16][0], Rb[src / 16][1], coef[J]);
        grid_axis_product<N, STRIDE, OFS, J + 1>(P0, P1, Rb, coef);
    }
}

// mat-vec against a vector whose entries live one per lane (lanes 0..N-1): acc[b] (+/-)= sum_p M[lane][p] * x[b][p],
// p ascending (the same FMA sequence per row as a broadcast-from-LDS loop).  `entry(p)` returns this lane's M[lane][p].
template <int N, bool NEG, int P = 0, class Entry>
__device__ __forceinline__ void dpp_matvec(double (&acc)[3], const double (&R)[(N + 15) / 16][3], const Entry& entry) {
    if constexpr (P + 4 <= N) {                                   // P % 4 == 0: the four pivots share a 16-block
        fmac12_dpp_from<P % 16, 0xf, NEG>(acc, R[P / 16], entry(std::integral_constant<int, P>{}),
                                          entry(std::integral_constant<int, P + 1>{}), entry(std::integral_constant<int, P + 2>{}),
                                          entry(std::integral_constant<int, P + 3>{}));
        dpp_matvec<N, NEG, P + 4>(acc, R, entry);
    } else if constexpr (P < N) {
        fmac3_dpp_bcast<P % 16, NEG>(acc, R[P / 16], entry(std::integral_constant<int, P>{}));
        dpp_matvec<N, NEG, P + 1>(acc, R, entry);
    }
}
template <int N>
__device__ __forceinline__ void dpp_replicate_vec(const double (&x)[3], int bp_addr, double (&R)[(N + 15) / 16][3]) {
#pragma unroll
    for (int blk = 0; blk < (N + 15) / 16; ++blk) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int lo = __builtin_amdgcn_ds_bpermute(bp_addr + 64 * blk, __double2loint(x[b]));
            const int hi = __builtin_amdgcn_ds_bpermute(bp_addr + 64 * blk, __double2hiint(x[b]));
            R[blk][b] = __hiloint2double(hi, lo);
        }
    }
}
// The entries of a lane's rows INSIDE their own 16x16 diagonal block live in registers (dg[i] = L''[row][16*(row/16)+i],
// zero at and right of the diagonal; written once, by the owning lane, when the row is appended): the latency-bound
// diagonal phase needs no LDS read and no masking, and the off-diagonal phase reads entries left of the diagonal block
// only, which exist for every row the row_mask lets through - plain ds_read_b128 with immediate offsets.
template <int RM, int I0>
__device__ __forceinline__ void dpp_diag4(const double (&dg)[16], double (&v)[3]) {
    fmac3_dpp_self<I0 + 0, RM>(v, dg[I0 + 0]);
    fmac3_dpp_self<I0 + 1, RM>(v, dg[I0 + 1]);
    fmac3_dpp_self<I0 + 2, RM>(v, dg[I0 + 2]);
    fmac3_dpp_self<I0 + 3, RM>(v, dg[I0 + 3]);
}
template <int RM, int I0>
__device__ __forceinline__ void dpp_off4(const double2_t (&l)[8], const double (&R)[3], double (&v)[3]) {
    fmac12_dpp_from<I0, RM, true>(v, R, l[I0 / 2].x, l[I0 / 2].y, l[I0 / 2 + 1].x, l[I0 / 2 + 1].y);
}
// diagonal block in DPP row K; n_rem = pivots that exist from the block's first pivot on (uniform): whole groups of four
// beyond them are skipped
template <int K>
__device__ __forceinline__ void dpp_diag_block(const double (&dg)[16], double (&v)[3], int n_rem) {
    dpp_diag4<(1 << K), 0>(dg, v);
    if (n_rem > 4) {
        dpp_diag4<(1 << K), 4>(dg, v);
        if (n_rem > 8) {
            dpp_diag4<(1 << K), 8>(dg, v);
            if (n_rem > 12) dpp_diag4<(1 << K), 12>(dg, v);
        }
    }
}
template <int RM>
__device__ __forceinline__ void dpp_off_block(const double2_t (&l)[8], const double (&R)[3], double (&v)[3]) {
    if constexpr (RM != 0) {
        dpp_off4<RM, 0>(l, R, v);
        dpp_off4<RM, 4>(l, R, v);
        dpp_off4<RM, 8>(l, R, v);
        dpp_off4<RM, 12>(l, R, v);
    }
}
// the lane's own-row pairs 8*qb .. 8*qb+7
__device__ __forceinline__ void dpp_load_pairs(const double2_t* row, int qb, double2_t (&l)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) l[q] = row[8 * qb + q];
}
// copy DPP row K of v to all four rows
template <int K>
__device__ __forceinline__ void dpp_replicate(const double (&v)[3], int bp_addr, double (&R)[3]) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int lo = __builtin_amdgcn_ds_bpermute(bp_addr + 64 * K, __double2loint(v[b]));
        const int hi = __builtin_amdgcn_ds_bpermute(bp_addr + 64 * K, __double2hiint(v[b]));
        R[b] = __hiloint2double(hi, lo);
    }
}

// pivots 16K .. 16K+15 of bank 0 (rows 0..63 in v0, rows 64.. in v1 when TWO)
template <int K, bool TWO>
__device__ __forceinline__ void dpp_bank0_block(const double2_t* row0, const double2_t* row1, int n_h, int bp_addr,
                                                const double (&dg0)[16], double (&v0)[3], double (&v1)[3]) {
    if constexpr (K < 4) {
        if (!TWO && 16 * K >= n_h) return;
        const bool more = TWO || 16 * (K + 1) < n_h;              // rows below this block exist
        double2_t la[8], lb[8];
        if (more) {                                               // requested now, needed after the diagonal phase
            if constexpr (K < 3) dpp_load_pairs(row0, K, la);
            if constexpr (TWO) dpp_load_pairs(row1, K, lb);
        }
        dpp_diag_block<K>(dg0, v0, TWO ? 16 : n_h - 16 * K);
        if (more) {
            double R[3];
            dpp_replicate<K>(v0, bp_addr, R);
            if constexpr (K < 3) dpp_off_block<((0xf << (K + 1)) & 0xf)>(la, R, v0);
            if constexpr (TWO) dpp_off_block<0xf>(lb, R, v1);
            dpp_bank0_block<K + 1, TWO>(row0, row1, n_h, bp_addr, dg0, v0, v1);
        }
    }
}
// pivots 64+16K .. of bank 1
template <int K>
__device__ __forceinline__ void dpp_bank1_block(const double2_t* row1, int n_h, int bp_addr, const double (&dg1)[16],
                                                double (&v1)[3]) {
    if constexpr (K < 4) {
        if (kWave + 16 * K >= n_h) return;
        const bool more = kWave + 16 * (K + 1) < n_h;
        double2_t lb[8];
        if constexpr (K < 3) {
            if (more) dpp_load_pairs(row1, 4 + K, lb);
        }
        dpp_diag_block<K>(dg1, v1, n_h - kWave - 16 * K);
        if constexpr (K < 3) {
            if (more) {
                double R[3];
                dpp_replicate<K>(v1, bp_addr, R);
                dpp_off_block<((0xf << (K + 1)) & 0xf)>(lb, R, v1);
                dpp_bank1_block<K + 1>(row1, n_h, bp_addr, dg1, v1);
            }
        }
    }
}

// The same substitution with L'' in the HBM/L2 workspace (the chain's factor does not fit LDS): identical arithmetic, but the
// own-row pairs of the off-diagonal phase are global loads, requested one whole block (a diagonal phase + a replicate +
// an off-diagonal phase, ~800 cycles) before they are needed.
template <int K, bool TWO>
__device__ __forceinline__ void dppg_bank0_block(const double2_t* row0, const double2_t* row1, int n_h, int bp_addr,
                                                 const double (&dg0)[16], const double2_t (&la)[8], const double2_t (&lb)[8],
                                                 double (&v0)[3], double (&v1)[3]) {
    if constexpr (K < 4) {
        if (!TWO && 16 * K >= n_h) return;
        const bool more = TWO || 16 * (K + 1) < n_h;
        double2_t la_n[8], lb_n[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) la_n[q] = la[q], lb_n[q] = lb[q];
        if (more) {                                               // the next block's pairs
            if constexpr (K < 2) {
                if (TWO || 16 * (K + 2) < n_h) dpp_load_pairs(row0, K + 1, la_n);
            }
            if constexpr (TWO && K < 3) dpp_load_pairs(row1, K + 1, lb_n);
        }
        dpp_diag_block<K>(dg0, v0, TWO ? 16 : n_h - 16 * K);
        if (more) {
            double R[3];
            dpp_replicate<K>(v0, bp_addr, R);
            if constexpr (K < 3) dpp_off_block<((0xf << (K + 1)) & 0xf)>(la, R, v0);
            if constexpr (TWO) dpp_off_block<0xf>(lb, R, v1);
            dppg_bank0_block<K + 1, TWO>(row0, row1, n_h, bp_addr, dg0, la_n, lb_n, v0, v1);
        }
    }
}
template <int K>
__device__ __forceinline__ void dppg_bank1_block(const double2_t* row1, int n_h, int bp_addr, const double (&dg1)[16],
                                                 const double2_t (&lb)[8], double (&v1)[3]) {
    if constexpr (K < 4) {
        if (kWave + 16 * K >= n_h) return;
        const bool more = kWave + 16 * (K + 1) < n_h;
        double2_t lb_n[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) lb_n[q] = lb[q];
        if constexpr (K < 2) {
            if (kWave + 16 * (K + 2) < n_h) dpp_load_pairs(row1, 4 + K + 1, lb_n);
        }
        dpp_diag_block<K>(dg1, v1, n_h - kWave - 16 * K);
        if constexpr (K < 3) {
            if (more) {
                double R[3];
                dpp_replicate<K>(v1, bp_addr, R);
                dpp_off_block<((0xf << (K + 1)) & 0xf)>(lb, R, v1);
                dppg_bank1_block<K + 1>(row1, n_h, bp_addr, dg1, lb_n, v1);
            }
        }
    }
}

// GPMPC_FAST_MAXTHREADS=512 (build.py: GPMPC_EXTRA_DEFS) halves the register budget so that two workgroups share a CU:
// only for tools/occupancy_experiment.py
#ifndef GPMPC_FAST_MAXTHREADS
#define GPMPC_FAST_MAXTHREADS 256
#endif

template <int T, int NR, int G_NY, int ENV, bool LHH_LDS, bool GRID>
__global__ __launch_bounds__(GPMPC_FAST_MAXTHREADS, 1) void rollout_fast_kernel(const RolloutArgs a) {
    constexpr int D = 2;
    constexpr int N1 = 9, N0 = NR / N1;                           // GRID: the real inputs are meshgrid(axis0[N0], axis1[N1], "ij")
    static_assert(N0 * N1 == NR && N0 + N1 <= 16, "the BASELINE real-data grids have 9 points on axis 1");
    constexpr int NS = T * (T + 1) / 2;
    constexpr int NX = (ENV == GPMPC_ENV_PENDULUM1D) ? 2 : 4;
    constexpr int NU = (ENV == GPMPC_ENV_PENDULUM1D) ? 1 : 2;
    constexpr int NRP = (NR + 1) & ~1;                            // broadcast-buffer row length (even -> b128 aligned)
    constexpr int NRS = ((NRP / 2) & 1) ? NRP : NRP + 2;          // row stride of lane==row tables: 16-byte slots, odd count
    constexpr int NPAIR = NR / 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_info[4];

    const GpParams& gp = a.gp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int spw = (blockDim.x >> 6) / G_NY;
    const int sw = wave / G_NY, o = wave - sw * G_NY;
    const long s = (long)blockIdx.x * spw + sw;
    const bool valid = s < a.Ns;
    const int H = a.H, nh_max = a.nh_max;
    const int nb1 = (nh_max > kWave) ? nh_max - kWave : 0;        // rows kept in bank 1

    // ---- LDS carve (all offsets even => 16-byte aligned) -----------------------------------------------------------
    double* Linv_all = smem;                                      // [G_NY][NR][NRS]  L_rr^-1 rows, shared by the workgroup
    double* ybuf_all = Linv_all + (GRID ? 0 : G_NY * NR * NRS);   // [spw][2][G_NY] (+pad)
    double* wb = smem + a.lds_shared + (long)wave * a.lds_per_wave;
    double* kvs = wb;                                             // [T][NRP]  k_r, then (after phase B) v_r: one buffer
    double* Lhr1 = kvs + T * NRP;                                 // [nb1][NRS]
    double* Lhh = LHH_LDS ? (Lhr1 + nb1 * NRS) : (a.ws + kZeroPage + (s * G_NY + o) * a.ws_chain_stride);
    double* ybuf = ybuf_all + sw * 2 * G_NY;
    double* dgs = Lhr1 + nb1 * NRS;                               // [3][16] diagonal-segment scratch (global-factor variant only)

    if constexpr (!GRID) {                                        // the grid root needs no L_rr^-1 (nor its LDS)
        for (int e = threadIdx.x; e < G_NY * NR * NRS; e += blockDim.x) {
            const int oo = e / (NR * NRS), rem = e - oo * NR * NRS;
            const int i = rem / NRS, j = rem - i * NRS;
            Linv_all[e] = (j < NR) ? plan_LinvT(a.plan, gp, oo)[j * NR + i] : 0.0;
        }
    }
    if constexpr (G_NY > 1) {
        if (threadIdx.x < 4) s_info[threadIdx.x] = 0;
    }
    __syncthreads();
    if (!valid) return;                                           // whole workgroups are valid when G_NY > 1 (spw == 1)

    // The LDS factor is zero-initialised (rows that do not exist yet read as zero).  The HBM/L2 factor is NOT (that was
    // 56 KB per chain, 0.68 GB per launch at BASELINE configs[2]): a lane whose row does not exist yet reads the shared
    // zero page at the head of the workspace instead (row0e / row1e below), existing rows only ever read entries left
    // of their diagonal block, which were written when the row was appended.
    if constexpr (LHH_LDS) {
        for (long e = lane; e < a.ws_chain_stride; e += kWave) Lhh[e] = 0.0;
    }
    for (int e = lane; e < nb1 * NRS; e += kWave) Lhr1[e] = 0.0;

    double il2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) il2[d] = gp.inv_l2[o][d];
    const double os = gp.os[o];
    double xr[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xr[d] = (lane < NR) ? a.X_r[lane * D + d] : 0.0;
    const double w_lane = (lane < NR) ? (GRID ? plan_grid_w(a.plan, gp, o) : plan_w(a.plan, gp, o))[lane] : 0.0;
    // GRID: this lane's real point is (ga, gc) on the grid; columns ga of Qa and gc of Qb, and os / sqrt(D) of the point
    const int lr = (lane < NR) ? lane : 0, ga = lr / N1, gc = lr - ga * N1;
    double qa[N0], qb[N1], dsc = 0.0;
    if constexpr (GRID) {
#pragma unroll
        for (int j = 0; j < N0; ++j) qa[j] = plan_grid_Qa(a.plan, gp, o)[j * N0 + ga];
#pragma unroll
        for (int j = 0; j < N1; ++j) qb[j] = plan_grid_Qb(a.plan, gp, o)[j * N1 + gc];
        dsc = plan_grid_dsc(a.plan, gp, o)[lr];
    }
    // GRID: the N0 + N1 distinct axis factors of the real data's kernel row are evaluated in ONE chain: lane j < N0 holds
    // axis-0 point j (real point (j, 0) = N1 j), lane N0 + j axis-1 point j (real point (0, j) = j)
    const bool g_ax0 = lane < N0;
    const int g_pt = g_ax0 ? lane * N1 : ((lane < N0 + N1) ? lane - N0 : 0);
    const double g_x = a.X_r[g_pt * D + (g_ax0 ? 0 : 1)], g_il2 = g_ax0 ? il2[0] : il2[1];
    // lane == row of L_rr^-1 (lanes >= NR read row 0 and discard the result)
    const double2_t* linvrow = reinterpret_cast<const double2_t*>(Linv_all + ((long)o * NR + ((lane < NR) ? lane : 0)) * NRS);

    double x[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) x[d] = a.x0[(a.x0_per_sample ? s * NX : 0) + d];

    double Lhr0[NR];                                              // bank-0 row (`lane`) of L_hr
#pragma unroll
    for (int i = 0; i < NR; ++i) Lhr0[i] = 0.0;
    double xh0[D], xh1[D];                                        // GP input of the point this lane's rows belong to
#pragma unroll
    for (int d = 0; d < D; ++d) {
        xh0[d] = 0.0;
        xh1[d] = 0.0;
    }
    double dinv0 = 0.0, dinv1 = 0.0, wown0 = 0.0, wown1 = 0.0;
    double dg0[16], dg1[16];                                      // diagonal-block segments of this lane's rows (see dpp_diag4)
#pragma unroll
    for (int i = 0; i < 16; ++i) dg0[i] = 0.0, dg1[i] = 0.0;
    int info_acc = 0;
    int n_h = 0, ro_nh = 0;                                       // appended slots, lhh_rowofs(n_h)
    // this lane's L_hh rows (clamped into the allocation; non-existent rows are discarded by ex0 / ex1 below)
    const double2_t* row0 = reinterpret_cast<const double2_t*>(Lhh + lhh_rowofs(min(lane, nh_max - 1)));
    const double2_t* row1 = reinterpret_cast<const double2_t*>(Lhh + lhh_rowofs(min(lane + kWave, nh_max - 1)));
    const double2_t* lhr1row = reinterpret_cast<const double2_t*>(Lhr1 + (long)min(lane, max(nb1 - 1, 0)) * NRS);
    const double2_t* zero2 = reinterpret_cast<const double2_t*>(Lhh + lhh_rowofs(nh_max));    // first slack pair: never written
    const double2_t* zpage = reinterpret_cast<const double2_t*>(a.ws);                         // kZeroPage zeros (HBM/L2 factor)
    const int a0t = lane - (lane / T) * T, a1t = (lane + kWave) - ((lane + kWave) / T) * T;   // task of this lane's rows
    const double cA0[2] = {(a0t == 1) ? il2[0] : 0.0, (a0t == 2) ? il2[1] : 0.0};              // [a == b > 0] / l_a^2
    const double cA1[2] = {(a1t == 1) ? il2[0] : 0.0, (a1t == 2) ? il2[1] : 0.0};
    // The chain's base samples (H*T doubles) and the input sequence (H*NU) are fetched ONCE, one step per
    // lane, and handed out with v_readlane inside the loop: a global load inside the step loop of a single wave costs
    // its full latency (the three z loads were serialised behind s_waitcnt vmcnt(0), which also waits for the
    // trajectory stores of the step).  With L_hh in LDS the loop then contains no vector-memory load at all.
    // layout: register c, lane t = component c of step t (H <= 43 < 64), so a pick is two v_readlane, no register select
    constexpr bool TRAJ_REGS = (G_NY == 1);
    double xq[TRAJ_REGS ? NX : 1];                                // state of step `lane` (see the step loop)
#pragma unroll
    for (int d = 0; d < (TRAJ_REGS ? NX : 1); ++d) xq[d] = 0.0;
    double zq[T], uq[NU];
#pragma unroll
    for (int c = 0; c < T; ++c) zq[c] = (lane < H) ? a.z[(long)lane * a.z_step_stride + (s * G_NY + o) * T + c] : 0.0;
#pragma unroll
    for (int i = 0; i < NU; ++i) uq[i] = (lane < H) ? a.u_ff[lane * NU + i] : 0.0;
    FPHASE_DECL;

#pragma unroll 1
    for (int t = 0; t < H; ++t) {
        // ---- input, GP input -----------------------------------------------------------------------------------
        double u[NU], xi[D];
        {
            double uf[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) uf[i] = readlane_f64(uq[i], t);
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
        if constexpr (TRAJ_REGS) {
            // the trajectory is collected one step per lane (H + 1 <= 44 lanes) and written once, coalesced, after the
            // loop: no exec-masked store region and no address arithmetic inside the step.  (Pendulum only: the car
            // kernel has no registers left for it.)
#pragma unroll
            for (int d = 0; d < NX; ++d) xq[d] = (lane == t) ? x[d] : xq[d];
        }
#ifndef GPMPC_ABLATE_STORES
        if (lane == 0 && o == 0) {
            if constexpr (!TRAJ_REGS) {
#pragma unroll
                for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + t] = x[d];
            }
            if (a.Xi) {
#pragma unroll
                for (int d = 0; d < D; ++d) a.Xi[(s * H + t) * D + d] = xi[d];
            }
        }
#endif

        // ---- kernel entries: the row against the real data (lane = real point) and this lane's rows of k_h ----------
        // one basic block: the exponentials are independent dependency chains and interleave.  The lane's row of
        // L_rr^-1 is requested first so that its LDS latency hides behind them.
        const int bp_addr = (lane & 15) << 2;                     // ds_bpermute address of "my lane of DPP row 0"
        double2_t lrow[(NR + 1) / 2];
        if constexpr (!GRID) {
#pragma unroll
            for (int jp = 0; jp < (NR + 1) / 2; ++jp) lrow[jp] = linvrow[jp];     // NRS >= NR + 1: the odd tail reads a zero pad
        }
        const bool two = n_h > kWave;
        const bool ex0 = lane < n_h, ex1 = lane + kWave < n_h;
        double kr[T], v0[T], v1[T];                               // v0 / v1: rows lane / lane+64 of k_h, then rhs, then v_h
        double gq = 0.0;
        // cov(task_a(row point), task_b(test point)) = k * (A_a * B_b + [a == b > 0] / l_a^2) with A_0 = B_0 = 1,
        // A_a = -q_a, B_b = +q_b (kern_entry): the row's task enters through a per-lane select of A and a
        // per-lane constant, so the evaluation is branch-free (the task-dependent branches of kern_entry cost ~17
        // exec-mask regions per step here)
        {
            static_assert(D == 2 && T == 3, "branch-free kernel entries are written for D = 2, T = 3");
            // squared distances to the real point, the bank-0 row's point and the bank-1 row's point (the latter only
            // needed once n_h > 64)
            double q[D], q0[D], q1[D], arg[3], ee[3];
            double ea = 0.0;                                      // GRID: this lane's axis factor (see g_ax0)
            arg[0] = -0.5 * kern_sqdist<D>(xr, xi, il2, q);
            arg[1] = -0.5 * kern_sqdist<D>(xh0, xi, il2, q0);
            arg[2] = -0.5 * kern_sqdist<D>(xh1, xi, il2, q1);
            if constexpr (GRID) {
                // separable row: exp(-(s0 + s1)/2) = ea * eb; only the N0 + N1 axis factors are needed (one per lane)
                const double gr = g_x - (g_ax0 ? xi[0] : xi[1]);
                gq = gr * g_il2;
                const double a3[3] = {-0.5 * gr * gq, arg[1], arg[2]};
                double e3[3];
                if (two) {                                        // uniform
                    expn_neg<3>(a3, e3);
                } else {
                    const double a2[2] = {a3[0], a3[1]};
                    double e2[2];
                    expn_neg<2>(a2, e2);
                    e3[0] = e2[0], e3[1] = e2[1], e3[2] = 0.0;
                }
                ea = e3[0], ee[0] = 0.0, ee[1] = e3[1], ee[2] = e3[2];
            } else if (two) {                                     // uniform
                exp3_neg(arg, ee);
            } else {
                const double arg2[2] = {arg[0], arg[1]};
                double ee2[2];
                expn_neg<2>(arg2, ee2);
                ee[0] = ee2[0], ee[1] = ee2[1], ee[2] = 0.0;
            }
            const double k = os * ee[0];
            const double k0 = ex0 ? os * ee[1] : 0.0;
            const double k1 = ex1 ? os * ee[2] : 0.0;
            if constexpr (GRID) {
                kr[0] = ea, kr[1] = ea * gq, kr[2] = 0.0;         // the axis factor and its derivative factor travel in kr
            } else {
                kr[0] = k, kr[1] = k * q[0], kr[2] = k * q[1];    // lanes >= NR: never used as pivots
            }
            const double A0 = (a0t == 0) ? 1.0 : ((a0t == 1) ? -q0[0] : -q0[1]);
            v0[0] = k0 * A0;
            v0[1] = k0 * fma(A0, q0[0], cA0[0]);
            v0[2] = k0 * fma(A0, q0[1], cA0[1]);
            const double A1 = (a1t == 0) ? 1.0 : ((a1t == 1) ? -q1[0] : -q1[1]);
            v1[0] = k1 * A1;
            v1[1] = k1 * fma(A1, q1[0], cA1[0]);
            v1[2] = k1 * fma(A1, q1[1], cA1[1]);
        }
        FPHASE(0);

        // ---- v_r = L_rr^-1 k_r (lane = row; own row and the broadcast k_r both two entries per ds_read_b128) ----
        double vr[T];
#pragma unroll
        for (int b = 0; b < T; ++b) vr[b] = 0.0;
#ifdef GPMPC_ABLATE_VR
        vr[0] = kr[0];
#else
        if constexpr (GRID) {
            // v_r = W k_r with the grid root W = D^-1/2 (Qa (x) Qb)^T (plan, gpmpc_device.hpp): the three kernel rows are
            // os (ea (x) eb), os (ea q0 (x) eb), os (ea (x) eb q1), so W k_r needs Qa^T {ea, ea q0} (N0 pivots) and
            // Qb^T {eb, eb q1} (N1 pivots) instead of NR pivots x three right-hand sides.  Axis-0 factors sit in the lanes
            // 0..N0-1, axis-1 factors in the lanes N0..N0+N1-1 (g_ax0 above).
            static_assert(T == 3, "written for three right-hand sides");
            double Rg[3][2];                                      // DPP block 0 holds all N0 + N1 <= 16 axis factors
            Rg[0][0] = bpermute_f64(kr[0], bp_addr);
            Rg[0][1] = bpermute_f64(kr[1], bp_addr);
            double PA0 = 0.0, PA1 = 0.0, PB0 = 0.0, PB1 = 0.0;
            grid_axis_product<N0, 1, 0>(PA0, PA1, Rg, qa);
            grid_axis_product<N1, 1, N0>(PB0, PB1, Rg, qb);
            const double s0 = dsc * PB0;
            vr[0] = s0 * PA0;
            vr[1] = s0 * PA1;
            vr[2] = dsc * PA0 * PB1;
        } else {
            // k_r replicated to all four DPP rows (16 real points per register), then one v_fmac_f64_dpp per pivot and
            // right-hand side against the lane's own row of L_rr^-1 (two entries per ds_read_b128): no LDS broadcasts
            static_assert(T == 3, "dpp_matvec is written for three right-hand sides");
            double Rk[(NR + 15) / 16][3];
            dpp_replicate_vec<NR>(kr, bp_addr, Rk);
            dpp_matvec<NR, false>(vr, Rk, [&](auto pc) {
                constexpr int p = decltype(pc)::value;
                return (p & 1) ? lrow[p / 2].y : lrow[p / 2].x;
            });
        }
#endif
        if (lane >= NR) {
#pragma unroll
            for (int b = 0; b < T; ++b) vr[b] = 0.0;
        }
        double pm[T], pss[NS];
        {
            int e = 0;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                if (lane < NR) kvs[b * NRP + lane] = vr[b];
                pm[b] = vr[b] * w_lane;
#pragma unroll
                for (int c = 0; c <= b; ++c) pss[e++] = vr[b] * vr[c];
            }
        }
        wave_sync_lds();
        // The rows this step appends (base = n_h .. n_h + T - 1) get v_r^T as their L_hr rows, and v_r is in kvs from here
        // on: the owning lanes fetch it NOW - their rows do not exist yet (ex0 is false: whatever they compute below is
        // discarded, and it stays finite) - instead of at the end of the step, where the 18 LDS reads and their wait sat
        // on the serial spine between the sampling and the next step.
        if (n_h < kWave && t + 1 < H) {                                   // uniform: a new row lives in bank 0 (registers)
            const int a0e = lane - n_h;
            if (a0e >= 0 && a0e < T) {
                const double* src = kvs + a0e * NRP;
#pragma unroll
                for (int ip = 0; ip < NPAIR; ++ip) {
                    const double2_t vv = *reinterpret_cast<const double2_t*>(src + 2 * ip);
                    Lhr0[2 * ip] = vv.x;
                    Lhr0[2 * ip + 1] = vv.y;
                }
                if (NR & 1) Lhr0[NR - 1] = src[NR - 1];
            }
        }
        FPHASE(1);

        if (n_h > 0) {
            double2_t lg0[8], lg1[8];                             // global-factor variant: own-row pairs of pivot block 0
            const double2_t* row0e = (LHH_LDS || ex0) ? row0 : zpage;     // rows that do not exist yet: the zero page
            const double2_t* row1e = (LHH_LDS || ex1) ? row1 : zpage;
            if constexpr (!LHH_LDS) {
#pragma unroll
                for (int q = 0; q < 8; ++q) lg0[q] = double2_t{0.0, 0.0}, lg1[q] = double2_t{0.0, 0.0};
                if (n_h > 16) dpp_load_pairs(row0e, 0, lg0);
                if (two) dpp_load_pairs(row1e, 0, lg1);
            }
            // ---- rhs = k_h - L_hr v_r -----------------------------------------------------------------------------
#ifndef GPMPC_ABLATE_RHS
            {
                // v_r replicated to the DPP rows; bank-0 rows of L_hr come from registers, bank-1 rows from LDS
                double Rv[(NR + 15) / 16][3];
                dpp_replicate_vec<NR>(vr, bp_addr, Rv);
                dpp_matvec<NR, true>(v0, Rv, [&](auto pc) { return Lhr0[decltype(pc)::value]; });
                if (two) {
                    double2_t l1[(NR + 1) / 2];
#pragma unroll
                    for (int ip = 0; ip < (NR + 1) / 2; ++ip) l1[ip] = lhr1row[ip];
                    dpp_matvec<NR, true>(v1, Rv, [&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        return (p & 1) ? l1[p / 2].y : l1[p / 2].x;
                    });
                }
            }
#endif
            FPHASE(2);

            // ---- forward substitution (see the header comment) -----------------------------------------------------
            // ring of 4 row pairs per bank = 8 pivots of look-ahead; reads past a row / the allocation land in the
            // slack or in later rows and are masked (finished rows) or multiplied by a zero pivot value.  All T pivot
            // values are broadcast (v_readlane) before the FMAs that consume them, so the SGPR write latency overlaps.
#ifdef GPMPC_ABLATE_SUBST
            if (n_h >= 0) {
            } else
#endif
            if constexpr (LHH_LDS) {
                static_assert(T == 3, "the DPP substitution is written for three right-hand sides");
                if (!two) {
                    dpp_bank0_block<0, false>(row0, row1, n_h, bp_addr, dg0, v0, v1);
                } else {
                    dpp_bank0_block<0, true>(row0, row1, n_h, bp_addr, dg0, v0, v1);
                    dpp_bank1_block<0>(row1, n_h, bp_addr, dg1, v1);
                }
            } else {
                double2_t la[8], lb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) la[q] = lg0[q], lb[q] = lg1[q];     // requested before the L_hr product
                if (!two) {
                    dppg_bank0_block<0, false>(row0e, row1e, n_h, bp_addr, dg0, la, lb, v0, v1);
                } else {
                    dppg_bank0_block<0, true>(row0e, row1e, n_h, bp_addr, dg0, la, lb, v0, v1);
                    double2_t lc[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) lc[q] = la[q];
                    if (kWave + 16 < n_h) dpp_load_pairs(row1e, 4, lc);
                    dppg_bank1_block<0>(row1e, n_h, bp_addr, dg1, lc, v1);
                }
            }
            // v = rhs / L_pp (own rows; rows that do not exist yet are dropped); partial sums
            {
                int e = 0;
#pragma unroll
                for (int b = 0; b < T; ++b) {
                    v0[b] = ex0 ? v0[b] * dinv0 : 0.0;
                    v1[b] = ex1 ? v1[b] * dinv1 : 0.0;
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
            // nine wave sums: two lane-swap trees of four and one DPP ladder, in lockstep (wave_sum9)
            static_assert(T == 3, "the reduction grouping below is written for T = 3");
            double r[NS];
#ifdef GPMPC_ABLATE_REDUCE
            mu[0] = readlane_f64(pm[0], 0), mu[1] = readlane_f64(pm[1], 0), mu[2] = readlane_f64(pm[2], 0);
            for (int e2 = 0; e2 < NS; ++e2) r[e2] = readlane_f64(pss[e2], 0) * 1e-3;
#else
            {
                const double qa[4] = {pm[0], pm[1], pm[2], pss[0]}, qb[4] = {pss[1], pss[2], pss[3], pss[4]};
                double ra[4], rb[4];
                wave_sum9(qa, qb, pss[5], ra, rb, r[5]);
                mu[0] = ra[0], mu[1] = ra[1], mu[2] = ra[2], r[0] = ra[3];
                r[1] = rb[0], r[2] = rb[1], r[3] = rb[2], r[4] = rb[3];
            }
#endif
            int e = 0;
#pragma unroll
            for (int b = 0; b < T; ++b) {
#pragma unroll
                for (int c = 0; c <= b; ++c) {
                    const double kss = (b == c) ? ((b == 0) ? os : os * il2[b - 1]) : 0.0;
                    const double val = kss - r[e++];
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
        double C[T][T], cinv[T];
        bool c_ok;
        {
            double Sn[T][T];
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int c = 0; c < T; ++c) Sn[b][c] = S[b][c] + ((b == c) ? gp.noise[b] : 0.0);
#ifdef GPMPC_ABLATE_SAMPLE
            c_ok = true;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                cinv[b] = Sn[b][b];
#pragma unroll
                for (int c = 0; c < T; ++c) C[b][c] = Sn[b][c], R[b][c] = S[b][c];
            }
#else
            // chol(S + noise) and the root of S are independent: both factorisations advance in lockstep
            double rinv[T];
            bool r_ok;
            chol3_pair_fast(Sn, S, C, R, cinv, rinv, c_ok, r_ok);
            if (!r_ok) info_acc |= root_small_fast_retry<T>(S, gp.jitter, R);
#endif
        }
        double zt[T];
#pragma unroll
        for (int c = 0; c < T; ++c) zt[c] = readlane_f64(zq[c], t);
        double y[T];
#pragma unroll
        for (int b = 0; b < T; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c <= b; ++c) acc = fma(R[b][c], zt[c], acc);
            double yb = acc + mu[b];
            if (all_zero) yb = mu[b];
            // clip to mu +- beta sqrt(var): the square root is only needed when the clip is active
            const double dlt = yb - mu[b];
            if (dlt * dlt > a.beta * a.beta * var[b]) {
                const double sd = a.beta * sqrt(var[b]);
                yb = fmin(fmax(yb, mu[b] - sd), mu[b] + sd);
            }
            y[b] = yb;
        }
#ifdef GPMPC_ABLATE_STORES
        if (lane == 0 && a.Y && t < 0) {
#else
        if (lane == 0 && a.Y) {
#endif
#pragma unroll
            for (int b = 0; b < T; ++b) a.Y[((s * G_NY + o) * H + t) * T + b] = y[b];
        }
        FPHASE(5);

        // ---- append [v^T, chol(S + noise)] and w to the chain's factor (A.9) -------------------------------------
#ifdef GPMPC_ABLATE_APPEND
        if (t + 1 < H) n_h += T;
        if (t < 0) {
#else
        if (t + 1 < H) {
#endif
            double wn[T];
            if (!c_ok) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
#pragma unroll
            for (int b = 0; b < T; ++b) {
                double acc = y[b] - mu[b];
#pragma unroll
                for (int c = 0; c < b; ++c) acc = fma(-C[b][c], wn[c], acc);
                wn[b] = acc * cinv[b];
            }
            const int base = n_h;
            const int a0 = lane - base, a1 = lane + kWave - base;
            double dd0s[T] = {0.0, 0.0, 0.0}, dd1s[T] = {0.0, 0.0, 0.0};
            // (1)+(2) new rows base+c of L'': lane p owns the entry in column p:  L''[base+c][p] = v_p[c] / L_pp for
            // p < base, the column-scaled new diagonal block for base <= p < base+c.  Branch-free: every lane stores, lanes
            // at or right of the diagonal store 0.0 into slots of rows that do not exist yet (rows base+c+1.. are written
            // after row base+c) or, clamped, into the last slack slot of the chain; v_p[c] is exactly zero for p >= base.
            {
                const double d10 = C[1][0] * cinv[0], d20 = C[2][0] * cinv[0], d21 = C[2][1] * cinv[1];
                const int last = (int)a.ws_chain_stride - 1;
                double dd0[T], dd1[T];
                dd0[0] = 0.0, dd1[0] = 0.0;
                dd0[1] = (a0 == 0) ? d10 : 0.0, dd1[1] = (a1 == 0) ? d10 : 0.0;
                dd0[2] = (a0 == 0) ? d20 : ((a0 == 1) ? d21 : 0.0), dd1[2] = (a1 == 0) ? d20 : ((a1 == 1) ? d21 : 0.0);
                int ro = ro_nh;                                   // = lhh_rowofs(base), carried from step to step
#pragma unroll
                for (int c = 0; c < T; ++c) {
                    if (c > 0) ro += (base + c - 1) + ((base + c - 1) & 1);       // slot of row r: r rounded up to even
                    Lhh[min(ro + lane, last)] = fma(v0[c], dinv0, dd0[c]);
                    if (base + c > kWave) Lhh[min(ro + lane + kWave, last)] = fma(v1[c], dinv1, dd1[c]);      // uniform
                    dd0s[c] = dd0[c], dd1s[c] = dd1[c];
                }
                ro_nh = ro + (base + T - 1) + ((base + T - 1) & 1);
            }
            // (3) owners of the new rows: 1/L_pp, w_p, the point's GP input, and the L_hr row (= v_r^T)
            {
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
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    xh0[d] = new0 ? xi[d] : xh0[d];
                    xh1[d] = new1 ? xi[d] : xh1[d];
                }
                if (base < kWave) {                                       // uniform: a new row lives in bank 0 (registers)
                    // only the three owning lanes execute the loads (exec-masked): no select against the old row, and
                    // the loads can land directly in the registers that hold the row
                    if (new0) {                                           // (the L_hr row itself was fetched after phase 1)
                        if constexpr (LHH_LDS) {
                            // ... and read the diagonal-block segment of their new row back (entries at / right of the
                            // diagonal: the zero pair); same wave, LDS operations complete in order
                            const int r = lane, qb = 8 * (r >> 4);
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const double2_t pr = *((2 * (qb + q) < r) ? (row0 + qb + q) : zero2);
                                dg0[2 * q] = pr.x;
                                dg0[2 * q + 1] = pr.y;
                            }
                        }
                    }
                }
                if (base + T > kWave) {                                   // uniform: a new row lives in bank 1 (LDS)
#pragma unroll
                    for (int c = 0; c < T; ++c) {
                        if (base + c >= kWave && lane < NR) Lhr1[(long)(base + c - kWave) * NRS + lane] = vr[c];
                    }
                }
                if constexpr (LHH_LDS) {
                    if (base + T > kWave) {
                        if (new1) {
                            const int r = lane + kWave, qb = 8 * (r >> 4);
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const double2_t pr = *((2 * (qb + q) < r) ? (row1 + qb + q) : zero2);
                                dg1[2 * q] = pr.x;
                                dg1[2 * q + 1] = pr.y;
                            }
                        }
                    }
                } else {
                    // factor in the HBM/L2 workspace: the new rows' diagonal-block segments travel through a 3 x 16
                    // LDS scratch (lane p of the row's own 16-block deposits column p; zero at / right of the diagonal
                    // by construction of the stored values), no global read-back
#pragma unroll
                    for (int c = 0; c < T; ++c) {
                        const int blk = (base + c) >> 4;
                        if (blk < 4) {
                            if ((lane >> 4) == blk) dgs[c * 16 + (lane & 15)] = fma(v0[c], dinv0, (c == 0) ? 0.0 : ((c == 1) ? dd0s[1] : dd0s[2]));
                        } else {
                            if ((lane >> 4) == blk - 4) dgs[c * 16 + (lane & 15)] = fma(v1[c], dinv1, (c == 0) ? 0.0 : ((c == 1) ? dd1s[1] : dd1s[2]));
                        }
                    }
                    wave_sync_lds();
                    if (new0) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const double2_t pr = *reinterpret_cast<const double2_t*>(dgs + a0 * 16 + 2 * q);
                            dg0[2 * q] = pr.x;
                            dg0[2 * q + 1] = pr.y;
                        }
                    }
                    if (base + T > kWave) {
                        if (new1) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const double2_t pr = *reinterpret_cast<const double2_t*>(dgs + a1 * 16 + 2 * q);
                                dg1[2 * q] = pr.x;
                                dg1[2 * q + 1] = pr.y;
                            }
                        }
                    }
                }
            }
            n_h += T;
        }
        FPHASE(6);

        // ---- state hand-over ---------------------------------------------------------------------------------------
        double g[G_NY];
        if (G_NY == 1) {
            g[0] = y[0];
            wave_sync_lds();                       // the appended rows are read by other lanes in the next step
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

    if constexpr (TRAJ_REGS) {
        if (o == 0 && lane <= H) {
#pragma unroll
            for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + lane] = (lane == H) ? x[d] : xq[d];
        }
    } else {
        if (lane == 0 && o == 0) {
#pragma unroll
            for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + H] = x[d];
        }
    }
    if constexpr (G_NY == 1) {
        if (lane == 0) a.info[s] = info_acc;
    } else {
        if (info_acc && lane == 0) atomicOr(&s_info[sw], info_acc);
        __syncthreads();
        if (o == 0 && lane == 0) a.info[s] = s_info[sw];
    }
    FPHASE_STORE;
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
    if (g_rollout_pin != GPMPC_KERNEL_AUTO) return g_rollout_pin != GPMPC_KERNEL_FAST;
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

static bool fast_grid_root(const gpmpc_gp_desc_t* gp) {
    const char* eg = std::getenv("GPMPC_DISABLE_GRID_ROOT");
    return !(eg && eg[0] == '1') && gp->grid_n1 == 9 && gp->grid_n0 * 9 == gp->N_r &&
           plan_has_grid_root(gp->grid_n0, gp->grid_n1, gp->real_has_grad);
}

static void fast_plan(const gpmpc_gp_desc_t* gp, int nx, int H, bool force_global, FastPlan* fp) {
    const int NR = gp->N_r, T = 3, G = gp->g_ny;
    (void)nx;
    (void)H;
    const int NRP = (NR + 1) & ~1;
    const int NRS = ((NRP / 2) & 1) ? NRP : NRP + 2;
    const int nh_max = 3 * (H - 1);
    const int nb1 = nh_max > 64 ? nh_max - 64 : 0;
    fp->chain_doubles = lhh_doubles(nh_max, kRingLds);
    const long vec = (long)T * NRP + (long)nb1 * NRS;
    const size_t budget = 160 * 1024 - 64;
    const int max_spw = (G == 1) ? 4 : 1;
    const bool grid = fast_grid_root(gp);                         // the grid root keeps no L_rr^-1 in LDS
    auto shared_doubles = [&](int spw) { return ((grid ? 0 : G * NR * NRS) + ((G > 1) ? spw * 2 * G : 0) + 1) & ~1; };
    // L_hh stays in LDS only if that still leaves one wave on every SIMD (max_spw samples per workgroup): with fewer
    // resident samples the HBM/L2-workspace variant at full occupancy is faster (tools/horizon_sweep.py, Ns=4096:
    // H=31 1.07 vs 1.25 ms, H=43 1.9 vs 5.3 ms; at H<=30, where 4 samples fit, LDS wins 0.82 vs 0.95 ms)
    fp->lhh_lds = false;
    fp->spw = max_spw;
    if (!force_global) {
        const long per = vec + fp->chain_doubles;
        const size_t bytes = ((size_t)shared_doubles(max_spw) + (size_t)max_spw * G * ((per + 1) & ~1L)) * sizeof(double);
        fp->lhh_lds = bytes <= budget;
    }
    if (!fp->lhh_lds) fp->chain_doubles = lhh_doubles(nh_max, kRingGlobal);
    fp->waves = fp->spw * G;
    fp->lds_shared = shared_doubles(fp->spw);
    const long per = vec + (fp->lhh_lds ? fp->chain_doubles : 48);          // 48: diagonal-segment scratch (dgs)
    fp->lds_per_wave = (int)((per + 1) & ~1L);
    fp->lds_bytes = ((size_t)fp->lds_shared + (size_t)fp->waves * fp->lds_per_wave) * sizeof(double);
}

size_t rollout_fast_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int H) {
    return ((size_t)kZeroPage + (size_t)Ns * gp->g_ny * (size_t)lhh_doubles(3 * (H - 1), kRingGlobal)) * sizeof(double);
}

template <int NR, int G_NY, int ENV, bool GRID>
static int launch_fast(RolloutArgs& args, const FastPlan& fp, hipStream_t st) {
    const long nblk = (args.Ns + fp.spw - 1) / fp.spw;
    const dim3 grid((unsigned)nblk), block(64 * fp.waves);
    if (fp.lhh_lds) {
        auto k = rollout_fast_kernel<3, NR, G_NY, ENV, true, GRID>;
        GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes));
        hipLaunchKernelGGL(k, grid, block, fp.lds_bytes, st, args);
    } else {
        auto k = rollout_fast_kernel<3, NR, G_NY, ENV, false, GRID>;
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
    args.linv_in_lds = 1;
    args.ws_chain_stride = fp.chain_doubles;
    if (!fp.lhh_lds) {
        if (!ws || ws_bytes < rollout_fast_workspace_bytes(gp, args.Ns, args.H))
            return fail(GPMPC_E_WORKSPACE, "gpmpc_rollout: workspace too small");
        GPMPC_HIP_CHECK(hipMemsetAsync(ws, 0, kZeroPage * sizeof(double), st));      // the zero page (1 KB); the factor itself is not cleared
    }
    // the grid root of the plan (separable real-data kernel row) when the real inputs are the reference's tensor grid
    const bool grid = fast_grid_root(gp);
    if (env->env_id == GPMPC_ENV_PENDULUM1D)
        return grid ? launch_fast<36, 1, GPMPC_ENV_PENDULUM1D, true>(args, fp, st)
                    : launch_fast<36, 1, GPMPC_ENV_PENDULUM1D, false>(args, fp, st);
    return grid ? launch_fast<45, 3, GPMPC_ENV_CAR_RESIDUAL, true>(args, fp, st)
                : launch_fast<45, 3, GPMPC_ENV_CAR_RESIDUAL, false>(args, fp, st);
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_fast_phases(long long* out /*[host] 16*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_fast_phase_cycles), 16 * sizeof(long long)));
    return GPMPC_OK;
}
