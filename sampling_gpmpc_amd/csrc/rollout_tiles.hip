// rollout_tiles_kernel: the THROUGHPUT form of the re-conditioned rollout (mode R, T = 3 label slots per point, value-only
// real labels on a tensor grid).  gfx950, wave64, one wave per workgroup, one workgroup per SIMD (512 registers).
//
// The tuned kernel of rollout_fast.hip gives a chain (sample, output) a whole wave: at BASELINE configs[1] that is what
// fills the chip, but beyond Ns = 1024 chains it costs Ns / 1024 rounds, 18 % of its lanes do algorithmic work and the car
// streams 60-97x its algorithmic traffic.  Here a wave carries FOUR chains (pendulum: four samples; car: the three outputs
// of one sample, the fourth is a copy of the third) and the forward substitution runs on the matrix pipe:
//
//   * v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4x4 blocks, block b = lanes {16 k + 4 b + x} (the b-th quad of
//     every DPP row).  Lane maps (tools/ubench/mfma64_layout.hip, measured): A[i][k] at lane 16 k + 4 b + i, B[k][j] at
//     16 k + 4 b + j, D[i][j] at 16 i + 4 b + j.  With block == chain, a 4-row TILE of a chain's right-hand sides
//     (rows x {3 right-hand sides + one spare column}) is one FP64 register, the chain's factor is a lower-triangular
//     matrix of 4x4 tiles (one register per tile), and  acc_r -= L_rp V_p  is ONE instruction for all four chains, at
//     17 cycles, 15 of the SIMD's 16 FP64 FMA per clock (a lone wave issues v_fma_f64 every 6.8 cycles: 9.5 per clock).
//     A register holding a matrix X in the "natural" layout (X[r][c] at lane 16 r + 4 b + c = the B / D map) acts as X^T
//     when used as the A operand:  mfma(X, Y) = X^T Y.  Factor tiles are therefore kept as (-L_rp)^T in natural layout;
//     the diagonal tiles as (L_rr^-1)^T, so that the whole solve is a left-looking sequence of MFMAs
//         acc = R_r + sum_{p<r} (-L_rp) V_p ;  V_r = L_rr^-1 acc,
//     two accumulators alternating (a dependent FP64 MFMA needs 4 software wait states = an MFMA in between).
//   * the spare column carries the whitened labels: R[:, y] = y_h - mu_real(x_h), hence V[:, y] = w_h, and ONE more MFMA
//     per tile, S' += V_r^T V_r, yields v^T v (the posterior covariance's subtrahend) and v^T w (the mean) together.  The
//     real-data block enters the same way: 12 pseudo-tiles of phi_e = dsc_e A_a B_c (the grid root, gpmpc_device.hpp) and
//     w_E.  No cross-lane reduction ladder anywhere.
//   * the first NRA tile rows of the factor live in AGPRs (MFMA reads its A operand from there at no cost; VALU cannot
//     address them), the rest streams from an HBM/L2 workspace, one coalesced 512-byte load per tile and wave, a tile row
//     ahead.  New rows are V itself: in step t the right-hand side of task c is put in column (n_h + c) mod 4, so the lane
//     that holds v_p[c] IS the lane of the new row's entry in the A operand - appending is one masked store per tile.
//   * everything that is not a triangular solve runs on the VALU with chain == DPP row and lane == conditioning POINT
//     (16 points per pass): kernel entries, and the real-data correction  L_hr v_r  in the grid root's Kronecker form
//     (a point keeps Qa^T{ea, ea q0} and Qb^T{eb, eb q1}: 28 doubles in LDS instead of three 45-vectors), per-chain
//     operands broadcast with v_fmac_f64_dpp row_newbcast.  A 1.5 KB LDS scratch converts between the two lane maps.
#include "gpmpc_host.hpp"
#include "rollout_args.hpp"

#include <type_traits>
#include <utility>

namespace gpmpc {

typedef double double2_v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_v __attribute__((ext_vector_type(4)));

// GPMPC_TILES_DEBUG (build.py: GPMPC_EXTRA_DEFS=-DGPMPC_TILES_DEBUG): wave 0 dumps its solved tiles and the tiles it appends
// at step GPMPC_TILES_DEBUG_STEP into g_tiles_dbg (tools/debug/tiles_dbg.py compares them with a dense numpy factor)
__device__ double g_tiles_dbg[64 * 64];
#ifndef GPMPC_TILES_DEBUG_STEP
#define GPMPC_TILES_DEBUG_STEP 2
#endif
#ifdef GPMPC_TILES_DEBUG
#define TDBG(slot, val) do { if (blockIdx.x == 0 && t == GPMPC_TILES_DEBUG_STEP) g_tiles_dbg[(slot) * 64 + lane] = (val); } while (0)
#else
#define TDBG(slot, val)
#endif

// GPMPC_TILES_PHASES: per-phase s_memtime totals of wave 0 (lane 0) in g_tiles_dbg[63 * 64 + phase] (tools/debug/tiles_phases.py)
#ifdef GPMPC_TILES_PHASES
#define TPH_DECL long long tph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tpht_ = __builtin_readcyclecounter(); long long tps_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tpst_ = tpht_
#define TPH(i) do { const long long n_ = __builtin_readcyclecounter(); tph_[i] += n_ - tpht_; tpht_ = n_; tpst_ = n_; } while (0)
// sub-phases (slots 8 .. 23 of the record): cycles since the last TPH / TPS mark
#define TPS(i) do { const long long n_ = __builtin_readcyclecounter(); tps_[i] += n_ - tpst_; tpst_ = n_; } while (0)
#define TPH_STORE do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) g_tiles_dbg[63 * 64 + i_] = (double)tph_[i_]; for (int i_ = 0; i_ < 16; ++i_) g_tiles_dbg[63 * 64 + 8 + i_] = (double)tps_[i_]; } } while (0)
#else
#define TPH_DECL
#define TPH(i)
#define TPS(i)
#define TPH_STORE
#endif

// Tile rows of the factor kept in AGPRs: 12 (78 tiles = 156 registers) where the parked state and constants leave room,
// else 11 (66 tiles).  The wave's other AGPRs: the parked doubles (state, test point, input, P0 / P1, U, 1/diag, five chain
// constants, the lane's point of each pass), the columns of Qa / Qb and three constants per grid-entry register.
__host__ __device__ constexpr int tiles_nra(int n0, int n1, int nx, int nt) {
    const int nps = (nt * 4 / 3 + 15) / 16, ne = (n0 * n1 + 15) / 16;
    const int other = 2 * (nx + 12 + 2 * nps) + 2 * (n0 + n1) + 6 * ne;
    return (156 + other <= 246) ? 12 : 11;                        // hipcc starts spilling AGPRs a few registers short of 256
}
__host__ __device__ constexpr int tri(int r) { return r * (r + 1) / 2; }
__device__ __forceinline__ void tiles_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------------------------
// FP64 MFMA, issued from inline asm so that the operand register classes are ours (A from an AGPR where the factor is
// resident).  hipcc pads nothing inside or around asm, and on gfx950 (tools/ubench/mfma64_hazard2.hip, mfma64_chain.hip)
//   MFMA D -> MFMA SrcC: 4 wait states; MFMA D -> MFMA SrcA/B or ANY VALU reader (also a spill of the result): 6;
//   VALU write -> MFMA source: 2; overwriting an MFMA's source right behind it: safe.
// Hence: chains of MFMAs are ONE statement each (rollout_tiles_mfma.inc, generated), every statement opens with s_nop 1
// and closes with s_nop 5.
// ---------------------------------------------------------------------------------------------------------------
#include "rollout_tiles_mfma.inc"
// D = A B (C = 0)
__device__ __forceinline__ double mfma_zero_v(double a, double b) {
    double d;
    asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0\n\ts_nop 5" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double mfma_zero_a(double a, double b) {
    double d;
    asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0\n\ts_nop 5" : "=&v"(d) : "a"(a), "v"(b));
    return d;
}
// acc[p & 1] += A[p] B[p], p = 0 .. N-1, in statements of at most 12 MFMAs
template <int N, bool AGPR, int P0 = 0>
__device__ __forceinline__ void mfma_rowsum(double (&acc)[2], const double* A, const double* B) {
    if constexpr (P0 < N) {
        constexpr int K = (N - P0 < 12) ? N - P0 : 12;
        if constexpr (AGPR) mfma_chain_a<K>(acc[0], acc[1], A + P0, B + P0);
        else mfma_chain_v<K>(acc[0], acc[1], A + P0, B + P0);
        mfma_rowsum<N, AGPR, P0 + K>(acc, A, B);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// DPP-row broadcasts (chain == DPP row on the VALU side).  A VGPR written by the VALU may be read through DPP two wait
// states later: the grid factors (written right before) carry an s_nop; E0 / E1 / P0 / P1 are written a phase before they
// are broadcast, and tools/check_dpp_hazard.py (a CPU test) scans the ISA for a copy or reload hipcc might put in front.
// ---------------------------------------------------------------------------------------------------------------
template <int LN>
__device__ __forceinline__ void tl_fmac2_bcast(double& acc0, double& acc1, double r0, double r1, double l) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %3, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %2 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(acc0), "+v"(acc1)
        : "v"(l), "v"(r0), "v"(r1), "n"(LN));
}
// s00 += E0@LN y0, s01 += E1@LN y0, s10 += E0@LN y1, s11 += E1@LN y1
template <int LN>
__device__ __forceinline__ void tl_fmac4_bcast(double& s00, double& s01, double& s10, double& s11, double e0, double e1, double y0,
                                               double y1) {
    asm("v_fmac_f64_dpp %0, %4, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %5, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %7 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %5, %7 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
        : "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11)
        : "v"(e0), "v"(e1), "v"(y0), "v"(y1), "n"(LN));
}
// o0 += P0@LN xa, o1 += P1@LN xa, o2 += P0@LN xb
template <int LN>
__device__ __forceinline__ void tl_fmac3_bcast(double& o0, double& o1, double& o2, double p0, double p1, double xa, double xb) {
    asm("v_fmac_f64_dpp %0, %3, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
        : "+v"(o0), "+v"(o1), "+v"(o2)
        : "v"(p0), "v"(p1), "v"(xa), "v"(xb), "n"(LN));
}

template <int N, int J = 0>
__device__ __forceinline__ void tl_axis_product(double& P0, double& P1, double k0, double k1, const double (&cf)[N]) {
    if constexpr (J < N) {
        tl_fmac2_bcast<J>(P0, P1, k0, k1, cf[J]);
        tl_axis_product<N, J + 1>(P0, P1, k0, k1, cf);
    }
}
// the four D-weighted inner products of a point: s[y][m][a] = sum_c E_m[a][c] Y_y[c]; entry (a, c) of E sits in register
// (a N1 + c) / 16 at lane (a N1 + c) % 16 of the chain's DPP row
template <int N0, int N1, int NE, int A = 0, int C = 0>
__device__ __forceinline__ void tl_inner(double (&s00)[N0], double (&s01)[N0], double (&s10)[N0], double (&s11)[N0],
                                         const double (&E0)[NE], const double (&E1)[NE], const double (&Y0)[N1], const double (&Y1)[N1]) {
    if constexpr (A < N0) {
        constexpr int e = A * N1 + C;
        tl_fmac4_bcast<e % 16>(s00[A], s01[A], s10[A], s11[A], E0[e / 16], E1[e / 16], Y0[C], Y1[C]);
        if constexpr (C + 1 < N1) tl_inner<N0, N1, NE, A, C + 1>(s00, s01, s10, s11, E0, E1, Y0, Y1);
        else tl_inner<N0, N1, NE, A + 1, 0>(s00, s01, s10, s11, E0, E1, Y0, Y1);
    }
}
template <int N0, int A = 0>
__device__ __forceinline__ void tl_outer(double& o0, double& o1, double& o2, double P0, double P1, const double (&xa)[N0],
                                         const double (&xb)[N0]) {
    if constexpr (A < N0) {
        tl_fmac3_bcast<A>(o0, o1, o2, P0, P1, xa[A], xb[A]);
        tl_outer<N0, A + 1>(o0, o1, o2, P0, P1, xa, xb);
    }
}

// A double kept in two AGPRs between its uses.  A lone wave has 256 VALU-addressable registers; the solve phase needs 64
// (V) + 128 (the ring of streamed tiles) of them, so everything that merely has to SURVIVE the solve (state, test point,
// the chain's constants, the incomplete diagonal tile ...) is parked by hand: left to hipcc, it is the freshly loaded
// ring entries that get spilled - to scratch, one s_waitcnt vmcnt(0) each.
struct Parked {
    int lo, hi;
    __device__ __forceinline__ void put(double v) {
        asm volatile("v_accvgpr_write_b32 %0, %2\n\tv_accvgpr_write_b32 %1, %3"
                     : "=a"(lo), "=a"(hi)
                     : "v"(__double2loint(v)), "v"(__double2hiint(v)));
    }
    __device__ __forceinline__ double get() const {
        int vlo, vhi;
        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(vlo), "=v"(vhi) : "a"(lo), "a"(hi));
        return __hiloint2double(vhi, vlo);
    }
};

// compile-time loop with an integral_constant index (register arrays need static indices)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS map of a workgroup (= one wave), doubles.  GL chains are stored (car: 3, the fourth chain aliases the third).
// ---------------------------------------------------------------------------------------------------------------
template <int N0, int N1>
struct TilesLds {
    static constexpr int XH = (N0 + 1) & ~1;                      // padded length of PA0 / PA1
    static constexpr int YH = (N1 + 1) & ~1;
    // record stride of XF in doubles: six consecutive points (a 16-lane group reads six) must start in six different 16-byte
    // slots mod 16, as must their PA1 halves: 12 doubles (6 slots: 0 6 12 2 8 14 | +3) do, 8 doubles (4 slots) do not -> 10
    static constexpr int XS = (XH == 4) ? 10 : 2 * XH;
    static constexpr int YS = 2 * YH;
    static constexpr int SXN = 96;                                // per chain: new-point record (2 XH + 2 YH <= 32), S' exchange (32), scalars (16)
    // per chain: rows x 4 columns of the lane-map converter: 48 rows (16 points x 3 tasks) or the padded grid (16 per register)
    static constexpr int SCRN = 4 * ((16 * ((N0 * N1 + 15) / 16) > 48) ? 16 * ((N0 * N1 + 15) / 16) : 48);
    __host__ __device__ static constexpr int per_chain(int npt) { return npt * (XS + YS + 4) + SCRN + SXN; }
};

// GPMPC_TILES_PF: live KB of the NEXT solve's streamed tiles that are pulled into the XCD's L2 while the right-hand sides
// (phase B, VALU + LDS only) are formed - see `prefetch` in the kernel.  0 disables it.
// Measured (car Ns = 4096, H = 40; profiles/r6_car_prefetch_sweep.txt): 0 KB 1.643 ms, 8: 1.609, 12: 1.595, 16: 1.592, 20: 1.613, 32:
// 1.700, 64: 1.99 - the XCD's 4 MB of L2 are shared by 128 waves (32 KB each, and the demand stream passes through it too):
// beyond ~16 KB per wave the prefetched sectors are evicted before they are used and everything is fetched twice.
#ifndef GPMPC_TILES_PF
#define GPMPC_TILES_PF 12
#endif
// GPMPC_TILES_KEEP: live KB at the head of a wave's streamed tiles whose loads may stay in the XCD's L2 (sc1); the loads
// beyond are streaming (sc1 + nt: evicted first).  A wave re-reads its streamed tiles once per step, in the same order: under
// plain LRU a cyclic pass over more bytes than the wave's share of the L2 (4 MB / 128 waves) hits NOTHING - the tail evicts
// the head just before it is wanted again.  With a non-temporal tail the head survives from step to step.  0: everything sc1.
// MEASURED AND NOT KEPT (profiles/r6_car_prefetch_sweep.txt): every nt load is slower than the L2 miss it avoids for others - 1.84 /
// 1.79 / 1.75 / 1.68 / 1.66 ms at 16 / 24 / 32 / 48 / 64 KB kept against 1.643 with no nt at all (all loads nt: 1.81): the stream
// is served by the Infinity Cache (the in-flight factors, ~150 MB, fit its 256 MB), which nt loads do not allocate in.
#ifndef GPMPC_TILES_KEEP
#define GPMPC_TILES_KEEP 0
#endif

// SEED: the call has conditioning-only passes and / or value-only points (gpmpc_rollout_seeded, hall_tasks = 1); the plain
// rollout is compiled without their selects and branches (3 % at configs[2])
template <int N0, int N1, int ENV, int NT, bool SEED>
__global__ __launch_bounds__(64, 1) void rollout_tiles_kernel(const RolloutArgs a) {
    constexpr int D = 2, T = 3;
    constexpr int G_NY = (ENV == GPMPC_ENV_PENDULUM1D) ? 1 : 3;
    constexpr int NX = (ENV == GPMPC_ENV_PENDULUM1D) ? 2 : 4;
    constexpr int NU = (ENV == GPMPC_ENV_PENDULUM1D) ? 1 : 2;
    constexpr int GL = (G_NY == 1) ? 4 : 3;                       // chains with their own LDS / results
    constexpr int NE = (N0 * N1 + 15) / 16;                       // registers holding one value per grid entry (a, c)
    constexpr int NRT = (N0 * N1 + 3) / 4;                        // pseudo-tiles of the real block
    constexpr int NPS = (NT * 4 / 3 + 15) / 16;                   // passes of 16 conditioning points
    constexpr int NRA = tiles_nra(N0, N1, NX, NT);
#ifdef GPMPC_TILES_NRV_EXTRA                                      // experiment knob (tools/tiles_try.sh)
    constexpr int NRV = NRA + GPMPC_TILES_NRV_EXTRA;
#else
    constexpr int NRV = NRA;                                      // + tile rows kept in VGPRs: 2 rows cost 60-90 spilled registers in phases A-C
#endif
    constexpr int NBT = tri(NRV) - tri(NRA);
#ifdef GPMPC_TILES_RC                                             // experiment knob (tools/tiles_try.sh)
    constexpr int RC = GPMPC_TILES_RC;
#else
    // register slots of the streamed-tile ring.  A DEEPER ring is slower (car Ns = 4096, H = 40: 1.75 ms at 24, 1.77 at 32,
    // 1.81 at 48, 1.88 at 60; tools/ubench/stream_tiles.hip shows the same for bare loads): a CU's four waves keep ~200
    // 128-byte lines in flight at 16 tiles each, beyond that requests only queue
    constexpr int RC = (NT <= 32) ? 24 : 32;
#endif
    using L = TilesLds<N0, N1>;
    static_assert(N0 + N1 <= 16 && NE <= 4 && NT % 8 == 0 && NT > NRA, "grid / tile limits");
    static_assert(RC >= 16 && RC % 2 == 0, "the ring holds a statement's 12 tiles + the diagonal tile, requested in pairs");
    extern __shared__ __attribute__((aligned(16))) double smem[];

    const GpParams& gp = a.gp;
    const int lane = threadIdx.x;
    const int cv = lane >> 4, l = lane & 15;                      // VALU side: chain = DPP row, lane l of the row
    const int kq = lane >> 4, bm = (lane >> 2) & 3, jq = lane & 3;    // MFMA side: chain = quad bm; row (A: column) kq, column (A: row) jq
    // conditioning-only passes in front of step 0 (gpmpc_rollout_seeded): n_h0 seed points observed with all T tasks, then
    // n_v0 observed like the rollout's own draws.  hall_tasks == 1 (value-only appended labels, reference src/agent.py:399-405):
    // a value-only point KEEPS its three row slots, the two derivative rows are dead - identity rows of the factor with a
    // zero right-hand side, which contribute nothing to any product - so that every index of the tile layout stays as it is
    const int n_pre = SEED ? a.n_h0 + a.n_v0 : 0;
    const bool th1 = SEED && a.hall_tasks == 1;
    const int H = a.H, npt_cap = max(n_pre + H - 1, 1);
    const long blk = blockIdx.x;
    // chain -> (sample, output)
    auto chain_sample = [&](int c) -> long { return (G_NY == 1) ? min(4 * blk + c, a.Ns - 1) : blk; };
    auto chain_live = [&](int c) -> bool { return (G_NY == 1) ? (4 * blk + c < a.Ns) : (c < 3); };
    const long s = chain_sample(cv);
    const int o = (G_NY == 1) ? 0 : min(cv, 2);
    const bool live = chain_live(cv);
    const int lc = min(cv, GL - 1), lcm = min(bm, GL - 1);       // LDS chain slots of the two lane maps

    const int pc = L::per_chain(npt_cap);
    double* XF = smem + lc * pc;                                  // [npt][XS]   PA0 | PA1
    double* YF = XF + npt_cap * L::XS;                            // [npt][YS]   PB0 | PB1
    double* YT = YF + npt_cap * L::YS;                            // [npt][4]    whitened-label residual y - mu_real of the point's rows
    double* SCR = YT + npt_cap * 4;                               // [48][4]     lane-map converter
    double* SX = SCR + L::SCRN;                                   // new-point record | S' exchange | chain scalars
    double* SCRm = smem + lcm * pc + npt_cap * (L::XS + L::YS + 4);
    double* SXm = SCRm + L::SCRN;

    // ---- per-chain / per-lane constants (VALU side) ----------------------------------------------------------------
    // axis lanes: lane l < N0 carries axis-0 point l, lane N0 + j axis-1 point j
    const bool ax0 = l < N0, ax1 = (l >= N0) && (l < N0 + N1);
    Parked c_il0, c_il1, c_os, c_gx, c_gil;
    {
        const double il0 = gp.inv_l2[o][0], il1 = gp.inv_l2[o][1];
        c_il0.put(il0);
        c_il1.put(il1);
        c_os.put(gp.os[o]);
        c_gx.put(ax0 ? a.X_r[(l * N1) * D] : (ax1 ? a.X_r[(l - N0) * D + 1] : 0.0));
        c_gil.put(ax0 ? il0 : (ax1 ? il1 : 0.0));
    }
    // grid entry e = 16 q + l = a N1 + c (three registers cover the N0 N1 entries); the plan's constants per entry and the
    // columns of Qa / Qb live ...
    // ... in AGPRs: 46 registers the VALU cannot address but v_accvgpr_read fetches in one instruction each, where a
    // global load of the plan (L2) costs its latency at the head of every step
    int cfA[2 * (N0 + N1)], geA[6 * NE];
    {
        const double* Qa = plan_grid_Qa(a.plan, gp, o);
        const double* Qb = plan_grid_Qb(a.plan, gp, o);
        auto park = [](int& dst, int v) { asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(v)); };
#pragma unroll
        for (int j = 0; j < N0 + N1; ++j) {
            const double c = (j < N0) ? (ax0 ? Qa[j * N0 + l] : 0.0) : (ax1 ? Qb[(j - N0) * N1 + (l - N0)] : 0.0);
            park(cfA[2 * j], __double2loint(c));
            park(cfA[2 * j + 1], __double2hiint(c));
        }
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = 16 * q + l;
            const bool ev = e < N0 * N1;
            const int ee = ev ? e : 0;
            const double v3[3] = {ev ? plan_grid_m2(a.plan, gp, o)[ee] : 0.0, ev ? plan_grid_dsc(a.plan, gp, o)[ee] : 0.0,
                                  ev ? plan_grid_w(a.plan, gp, o)[ee] : 0.0};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                park(geA[6 * q + 2 * k], __double2loint(v3[k]));
                park(geA[6 * q + 2 * k + 1], __double2hiint(v3[k]));
            }
        }
    }
    auto unpark = [](int lo, int hi) -> double {
        int vlo, vhi;
        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(vlo), "=v"(vhi) : "a"(lo), "a"(hi));
        return __hiloint2double(vhi, vlo);
    };
    // MFMA side constants: the identity in natural layout, and this lane's slot in the wave's factor workspace
    const double Inat = (kq == jq) ? 1.0 : 0.0;
    // Tiles of the wave's factor are stored in PAIRS: tiles 2q and 2q + 1 share 1024 bytes at wsu + 1024 q in which lane
    // (bm, kq, jq) owns 16 bytes (tile 2q's double, then tile 2q + 1's): the ring of streamed tiles fetches a pair with ONE
    // 16-byte load per lane (tools/ubench/stream_tiles.hip: 0.80 ms against 0.93 ms for this launch's loads at 8 bytes).
    // Accessed through a buffer descriptor (uniform base, scalar tile offset, 32-bit lane offset): with per-lane 64-bit
    // addresses hipcc hoists the address arithmetic of all 528 tiles out of the step loop and spills it.
    char* wsu = reinterpret_cast<char*>(a.ws + blk * a.ws_chain_stride);
    // lanes of a chain that carries no sample (the car's fourth quad, the tail of the last pendulum wave) address past the
    // descriptor's end: their loads return zero and their stores are dropped without memory traffic
    const bool live_m = (G_NY == 1) ? (4 * blk + bm < a.Ns) : (bm < 3);
    // inside a pair the 128 doubles are ordered chain-major (chain bm owns two 128-byte lines): the lines of a chain without
    // a sample are never fetched - in lane order every line would carry 32 dead bytes and all of them would move
    const unsigned lane16 = live_m ? (unsigned)(bm * 16 + kq * 4 + jq) * 16u : 0x7ffff000u;
    const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(wsu, 0, (int)(a.ws_chain_stride * 8), 0x00020000);
    // L2 prefetch (see the step loop): lane i < 48 names 32-byte sector i % 24 of the live 768 bytes of pair i / 24 (the dead
    // chain's quarter of a pair is never touched); lanes 48 .. 63 repeat lanes 0 .. 15 (same lines: no extra traffic)
    const unsigned pf_voff = (unsigned)(((lane % 48) / 24) * 1024 + ((lane % 48) % 24) * 32);
    const u32x4_v pf_rsrc = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(reinterpret_cast<uintptr_t>(wsu))),
                             (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((reinterpret_cast<uintptr_t>(wsu) >> 32) & 0xffffu)),
                             (unsigned)__builtin_amdgcn_readfirstlane((int)(a.ws_chain_stride * 8)), 0x00020000u};
    // landing slot: chain 0's S' exchange (32 doubles = 64 dwords) - written in phase F and read right behind it, dead from
    // there to the next phase F; the solve in between opens with s_waitcnt vmcnt(0)
    const unsigned pf_m0 = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) double*)(smem + max(n_pre + a.H - 1, 1) * (L::XS + L::YS + 4) + L::SCRN + 32));
    auto tile_off = [](int e) -> int { return (e >> 1) * 1024 + (e & 1) * 8; };
    auto tile_load = [&](unsigned voff, int e) -> double {
        // sc1 (aux bit 4): served by L2.  A tile row is re-read after this wave has stored into it (rows arrive three at a
        // time, the diagonal tile is rewritten); the CU's L1 keeps the line it fetched BEFORE the store (measured: stale
        // tiles at the third step), and streaming 4x the L1's size per step gains nothing from L1 anyway
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wsr, voff, tile_off(e), 16));
    };
    auto tile_store = [&](unsigned voff, int e, double v) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_v, v), wsr, voff, tile_off(e), 0);
    };

    Parked xP[NX], xiP[D], uP, pP[2], xptP[NPS][D], ucP, dcP;     // state; test point; last input; P0 / P1; the lane's points; U, 1/diag
#pragma unroll
    for (int d = 0; d < NX; ++d) xP[d].put(a.x0[(a.x0_per_sample ? s * NX : 0) + d]);
#pragma unroll
    for (int q = 0; q < NPS; ++q) xptP[q][0].put(0.0), xptP[q][1].put(0.0);
    ucP.put((kq == jq) ? 1.0 : 0.0);                             // the incomplete diagonal tile: L^T and 1/diag (natural layout)
    dcP.put((kq == jq) ? 1.0 : 0.0);
    // AGPR-resident tile rows 0 .. NRA-1 (natural layout, see header).  Every element is DEFINED by an asm load with an
    // "=a" output and only ever read through "a" operands, so the values are of the AGPR class from birth (a C++ array
    // that merely feeds "a" operands is allocated to VGPRs and spilled).  Row r is loaded when it completes and not read
    // before (rows >= rs stream), hence no initialisation.
    double At[tri(NRA)];
    double Bt[NBT > 0 ? NBT : 1];                                 // tile rows NRA .. NRV-1: ordinary registers, loaded when the row completes
#pragma unroll
    for (int i = 0; i < (NBT > 0 ? NBT : 1); ++i) Bt[i] = 0.0;
    int info_acc = 0;
    int n_pts = 0;                                                // appended points; n_h = 3 n_pts label rows

    // the feedback law's constants once, not one scalar round trip per step and use (they sit in the kernel-argument segment)
    double fbK[NU][NX], fbg[NX];
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) fbK[i][j] = a.env.K[i][j];
#pragma unroll
    for (int j = 0; j < NX; ++j) fbg[j] = a.env.x_goal[j];
    const bool use_fb = a.env.use_feedback != 0;
    const double env_dt = a.env.dt;
    double uf_next[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) uf_next[i] = a.u_ff[i];         // (step 0's input; a conditioning-only pass uses the seed's point)
    TPH_DECL;
#pragma unroll 1
    for (int tt = -n_pre; tt < H; ++tt) {
        const bool seeding = SEED && tt < 0;                      // uniform: condition on a given point, draw nothing
        const int t = seeding ? 0 : tt;
        const bool seed_full = seeding && (tt + n_pre) < a.n_h0;
        const bool vo_new = th1 && !seed_full;                    // this pass's point is observed in its value only
        const int n_h = 3 * n_pts, i0 = n_h & 3, nt = (n_h + 3) >> 2, nfull = n_h >> 2;
        const int ycol = (i0 + 3) & 3;
        // ---- input, GP input ---------------------------------------------------------------------------------------
        double xi[D];
        {
            double x[NX], u[NU];
#pragma unroll
            for (int d = 0; d < NX; ++d) x[d] = xP[d].get();
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const double uf = uf_next[i];                     // requested a step ago (a scalar round trip per step otherwise)
                if (use_fb) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; ++j) acc += (fbg[j] - x[j]) * fbK[i][j];
                    u[i] = -acc + uf;
                } else {
                    u[i] = uf;
                }
            }
            {
                const int tn1 = (tt + 1 < 0) ? 0 : min(tt + 1, H - 1);   // the next step's input: requested now, used a step later
#pragma unroll
                for (int i = 0; i < NU; ++i) uf_next[i] = a.u_ff[tn1 * NU + i];
            }
            xi[0] = x[(ENV == GPMPC_ENV_PENDULUM1D) ? 0 : (NX > 2 ? 2 : 0)];
            xi[1] = u[0];
            if (seeding) {                                        // the given point of this chain
                const int pt = tt + n_pre;
                const double* xs = seed_full ? a.X_h0 + ((s * G_NY + o) * (long)a.n_h0 + pt) * D
                                             : a.X_v0 + ((s * G_NY + o) * (long)a.n_v0 + (pt - a.n_h0)) * D;
                xi[0] = xs[0];
                xi[1] = xs[1];
            }
            xiP[0].put(xi[0]);
            xiP[1].put(xi[1]);
            uP.put(u[NU - 1]);
            if (l == 0 && live && o == 0 && !seeding) {
#pragma unroll
                for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + t] = x[d];
                if (a.Xi) {
#pragma unroll
                    for (int d = 0; d < D; ++d) a.Xi[(s * H + t) * D + d] = xi[d];
                }
            }
        }

        TPS(0);                                                   // A1: state, feedback law, GP input, trajectory stores
        // ---- phase A: the real block at the test point (grid root) ---------------------------------------------------
        double P0 = 0.0, P1 = 0.0;                                // lanes < N0: PA0', PA1'; lanes N0 .. N0+N1-1: PB0', PB1'
        double E0[NE], E1[NE];
        double Sr;                                                // S' of the real block (natural layout)
        {
            double cf[N0 + N1];                                   // column l of Qa (lanes < N0) / column l - N0 of Qb
#pragma unroll
            for (int j = 0; j < N0 + N1; ++j) cf[j] = unpark(cfA[2 * j], cfA[2 * j + 1]);
            double m2e[NE], dsce[NE], wEe[NE];
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                m2e[q] = unpark(geA[6 * q], geA[6 * q + 1]);
                dsce[q] = unpark(geA[6 * q + 2], geA[6 * q + 3]);
                wEe[q] = unpark(geA[6 * q + 4], geA[6 * q + 5]);
            }
            int ea_[NE], ec_[NE];
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int e = 16 * q + l, ee = (e < N0 * N1) ? e : 0;
                ea_[q] = ee / N1;
                ec_[q] = ee - ea_[q] * N1;
            }
            const double g_x = c_gx.get(), g_il = c_gil.get();
            const double gr = g_x - (ax0 ? xi[0] : xi[1]);
            const double gq = gr * g_il;
            const double ea = exp(-0.5 * gr * gq);
            const double k0 = ea, k1 = ea * gq;
            tl_axis_product<N0 + N1>(P0, P1, k0, k1, cf);
            // the new point's record (also what is appended): PA0 | PA1 | PB0 | PB1
            if (ax0) {
                SX[l] = P0;
                SX[L::XH + l] = P1;
            } else if (ax1) {
                SX[2 * L::XH + (l - N0)] = P0;
                SX[2 * L::XH + L::YH + (l - N0)] = P1;
            }
            tiles_sync_lds();
            TPS(1);                                               // A2: unpark, exponential, axis products, record to LDS
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const double pa0 = SX[ea_[q]], pa1 = SX[L::XH + ea_[q]];
                const double pb0 = SX[2 * L::XH + ec_[q]], pb1 = SX[2 * L::XH + L::YH + ec_[q]];
                E0[q] = m2e[q] * pb0;
                E1[q] = m2e[q] * pb1;
                const double d0 = dsce[q] * pb0;
                double* dst = SCR + (16 * q + l) * 4;             // column of task b: (i0 + b) & 3 (dynamic, uniform)
                dst[i0] = d0 * pa0;                               // (PA0, PB0): value
                dst[(i0 + 1) & 3] = d0 * pa1;                     // (PA1, PB0): d/dx0
                dst[(i0 + 2) & 3] = dsce[q] * pb1 * pa0;          // (PA0, PB1): d/dx1
                dst[ycol] = wEe[q];
            }
            tiles_sync_lds();
            TPS(2);                                               // A3: grid entries, pseudo-tiles to LDS
            double Sq[2] = {0.0, 0.0};
            double rt[NRT];
#pragma unroll
            for (int q = 0; q < NRT; ++q) rt[q] = SCRm[(4 * q + kq) * 4 + jq];
            mfma_rowsum<NRT, false>(Sq, rt, rt);
            const double Sa = Sq[0], Sb = Sq[1];
            Sr = Sa + Sb;
            tiles_sync_lds();                                     // SCR is reused by phase B
            TPS(3);                                               // A4: the real block's Gram on the matrix pipe
        }

        TPH(0);
        // ---- L2 prefetch of this step's streamed tiles ---------------------------------------------------------------------
        // The solve below re-reads tile rows NRV .. nt-1 from the workspace (the wave's own stores of earlier steps; by now
        // they sit in HBM / the Infinity Cache: 1024 waves x ~150 KB cycle through 32 MB of L2).  A lone wave cannot overlap
        // that stream with its VALU phases through registers (no room for a ring that survives phases A-C) nor through LDS
        // (full: the chains' point records) - but the XCD's L2 is a buffer nobody has to allocate: one dword per 32-byte
        // sector of the first PF live KB is requested here, 4.2 k cycles of VALU / LDS work ahead of the first use, as LDS-DMA
        // loads (`buffer_load_dword ... lds`: no destination register; all of them land in one 256-byte slot of LDS that is
        // dead at this point of the step).  Inline asm on purpose: hipcc's wait insertion would put a vmcnt(0) in front of the next LDS read for
        // a DMA it cannot prove disjoint.  The requests are older than every load of the solve, so the solve's counted waits
        // are unaffected, and the `s_waitcnt vmcnt(0)` at its head (after phase B) finds them landed.
        if constexpr (GPMPC_TILES_PF > 0) {
            if (nfull >= NRV) {                                   // (uniform) regime (ii): rows NRV .. nt-1 stream
                const int b0 = tri(NRV) * 512, b1 = tri(nt) * 512;      // byte range of the streamed tiles (pairs of 1024 B)
                int nins = (b1 - b0 + 2047) >> 11;                // one instruction touches the 48 live sectors of two pairs
                nins = min(nins, (GPMPC_TILES_PF * 1024 + 1535) / 1536);
                int so = b0;
#pragma unroll 1
                for (int j = 0; j < nins; ++j, so += 2048)
                    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                                 :: "s"(pf_m0), "v"(pf_voff), "s"(pf_rsrc), "s"(so) : "m0", "memory");
            }
        }
        // ---- phase B: right-hand sides of the hallucinated rows, 16 points per pass; phase C: into tile registers ------
        double V[NT];
#pragma unroll
        for (int r = 0; r < NT; ++r) V[r] = 0.0;
        static_for<0, NPS>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if (16 * q < n_pts) {                                 // uniform
                const double il0 = c_il0.get(), il1 = c_il1.get(), os = c_os.get();
                const double xpt0 = xptP[q][0].get(), xpt1 = xptP[q][1].get();
                const int jp = 16 * q + l;
                const bool ex = jp < n_pts;
                const bool exd = ex && !(th1 && jp >= a.n_h0);    // the point's derivative rows exist (not value-only)
                const int jr = min(jp, npt_cap - 1);
                // kernel entries against the test point: cov(task a of the point, task b of the test point)
                const double d0 = xpt0 - xi[0], d1 = xpt1 - xi[1];
                const double q0 = d0 * il0, q1 = d1 * il1;
                const double kk = ex ? os * exp(-0.5 * (d0 * q0 + d1 * q1)) : 0.0;
                double out[T][T];
                {
                    const double Aa[T] = {1.0, -q0, -q1}, Bb[T] = {1.0, q0, q1}, cd[T] = {0.0, il0, il1};
#pragma unroll
                    for (int aa = 0; aa < T; ++aa)
#pragma unroll
                        for (int b = 0; b < T; ++b) out[aa][b] = kk * fma(Aa[aa], Bb[b], (aa == b) ? cd[aa] : 0.0);
                }
                TPS(4);                                           // B1: kernel entries against the test point (one exponential)
                // minus the real-data correction  v_r(point, task a) . v_r(test point, task b)  in Kronecker form
                {
                    double PAj[2][N0], PBj[2][N1];
                    const double* xr = XF + jr * L::XS;
                    const double* yr = YF + jr * L::YS;
#pragma unroll
                    for (int i = 0; i < L::XH / 2; ++i) {
                        const double2_v u0 = *reinterpret_cast<const double2_v*>(xr + 2 * i);
                        const double2_v u1 = *reinterpret_cast<const double2_v*>(xr + L::XH + 2 * i);
                        if (2 * i < N0) PAj[0][2 * i] = u0.x, PAj[1][2 * i] = u1.x;
                        if (2 * i + 1 < N0) PAj[0][2 * i + 1] = u0.y, PAj[1][2 * i + 1] = u1.y;
                    }
#pragma unroll
                    for (int i = 0; i < L::YH / 2; ++i) {
                        const double2_v u0 = *reinterpret_cast<const double2_v*>(yr + 2 * i);
                        const double2_v u1 = *reinterpret_cast<const double2_v*>(yr + L::YH + 2 * i);
                        if (2 * i < N1) PBj[0][2 * i] = u0.x, PBj[1][2 * i] = u1.x;
                        if (2 * i + 1 < N1) PBj[0][2 * i + 1] = u0.y, PBj[1][2 * i + 1] = u1.y;
                    }
                    double s00[N0], s01[N0], s10[N0], s11[N0];    // s[y][m]: Y = PB_y of the point, E_m of the test point
#pragma unroll
                    for (int i = 0; i < N0; ++i) s00[i] = s01[i] = s10[i] = s11[i] = 0.0;
                    tl_inner<N0, N1, NE>(s00, s01, s10, s11, E0, E1, PBj[0], PBj[1]);
                    // rows a: (X, y) = (PA0, 0), (PA1, 0), (PA0, 1); right-hand sides b: (P', m) = (P0, 0), (P1, 0), (P0, 1)
                    double xa[N0], xb[N0], c0, c1, c2;
#pragma unroll
                    for (int i = 0; i < N0; ++i) xa[i] = PAj[0][i] * s00[i], xb[i] = PAj[0][i] * s01[i];
                    c0 = c1 = c2 = 0.0;
                    tl_outer<N0>(c0, c1, c2, P0, P1, xa, xb);
                    out[0][0] -= c0, out[0][1] -= c1, out[0][2] -= c2;
#pragma unroll
                    for (int i = 0; i < N0; ++i) xa[i] = PAj[1][i] * s00[i], xb[i] = PAj[1][i] * s01[i];
                    c0 = c1 = c2 = 0.0;
                    tl_outer<N0>(c0, c1, c2, P0, P1, xa, xb);
                    out[1][0] -= c0, out[1][1] -= c1, out[1][2] -= c2;
#pragma unroll
                    for (int i = 0; i < N0; ++i) xa[i] = PAj[0][i] * s10[i], xb[i] = PAj[0][i] * s11[i];
                    c0 = c1 = c2 = 0.0;
                    tl_outer<N0>(c0, c1, c2, P0, P1, xa, xb);
                    out[2][0] -= c0, out[2][1] -= c1, out[2][2] -= c2;
                }
                TPS(5);                                           // B2: the Kronecker correction (records from LDS, DPP products)
                const double2_v y01 = *reinterpret_cast<const double2_v*>(YT + jr * 4);
                const double y2 = YT[jr * 4 + 2];
                const double yt[T] = {y01.x, y01.y, y2};
#pragma unroll
                for (int aa = 0; aa < T; ++aa) {
                    double* dst = SCR + (3 * l + aa) * 4;
                    const bool er = (aa == 0) ? ex : exd;
                    dst[i0] = er ? out[aa][0] : 0.0;
                    dst[(i0 + 1) & 3] = er ? out[aa][1] : 0.0;
                    dst[(i0 + 2) & 3] = er ? out[aa][2] : 0.0;
                    dst[ycol] = er ? yt[aa] : 0.0;
                }
                tiles_sync_lds();
                static_for<0, 12>([&](auto wc) {                  // (rows of points that do not exist yet were written as zeros)
                    constexpr int w = decltype(wc)::value, tg = 12 * q + w;
                    if constexpr (tg < NT) V[tg] = SCRm[(4 * w + kq) * 4 + jq];
                });
                tiles_sync_lds();
                TPS(6);                                           // B3 / C: through the lane-map converter into tile registers
            }
        });

        pP[0].put(P0);
        pP[1].put(P1);
        TPH(1);
        TDBG(0, V[0]);
        TDBG(1, V[1]);
        TDBG(2, V[2]);
        TDBG(3, V[3]);
        // ---- phase D: forward substitution, left-looking over tile rows; phase E: S' += V_r^T V_r -----------------------
        // Tile rows below `rs` are complete and AGPR-resident; rows rs .. nt-1 stream from the workspace through a ring of
        // 64 registers (sched_barrier keeps hipcc from hoisting all the loads to the top, which spills hundreds of registers).
        double Sh[2] = {0.0, 0.0};
        double Vlast = 0.0;                                      // V of the last tile row solved (phase H wants the incomplete one)
        if (nt > 0) {
            // the previous step's tile stores precede this step's tile loads (same lane, same addresses); they were
            // issued thousands of cycles ago, the wait is a formality
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const bool part = (n_h & 3) != 0;                     // the last tile row is incomplete: rows >= n_h are masked
            // The incomplete tile row: its rows >= n_h do not exist yet and their entries in the workspace are whatever the
            // last launch left there.  Row i of a product (-L_rp) V_p depends on row i of L_rp alone, so nothing has to be
            // masked but the accumulated row block itself (D layout: row index kq), once.
            const bool accex = !part || (4 * nfull + kq) < n_h;
            // RC register slots hold the streamed tiles in flight, tile e = tri(r) + p in slot e mod RC.  Tiles are requested
            // in the order they are consumed, always RC tiles ahead: in front of every statement of <= 12 MFMAs the slots the
            // previous statements emptied are re-requested (a lead of RC x ~21 cycles of MFMAs ~ 1300 cycles, above the L2
            // latency under load, whatever the row length).
            // hipcc's s_waitcnt insertion counts loads exactly only along straight-line code: a load under an `if` that
            // rejoins makes every later wait a vmcnt(0), which waits for the prefetch just issued.  Hence two regimes:
            //   (i)  fewer than NRA complete tile rows: all of them resident, only the incomplete row streams;
            //   (ii) rows 0 .. NRA-1 resident (straight line), rows NRA .. stream with UNCONDITIONAL requests - up to RC
            //        tiles beyond row nt-1 are requested and never used (inside the wave's workspace) - and early exits only.
            constexpr int TOT = tri(NT), E0 = tri(NRV);
            double ring[RC];
            auto request = [&](auto loc, auto hic) {              // tile PAIRS of [lo & ~1, hi & ~1), clamped to the workspace
                constexpr int lo = decltype(loc)::value & ~1, hi = (decltype(hic)::value < TOT ? decltype(hic)::value : TOT) & ~1;
                static_for<lo / 2, hi / 2>([&](auto qc) {
                    constexpr int e = 2 * decltype(qc)::value;
                    constexpr int aux = (GPMPC_TILES_KEEP > 0 && (e - E0) * 384 >= GPMPC_TILES_KEEP * 1024) ? 18 : 16;
                    const u32x4_v v = __builtin_amdgcn_raw_buffer_load_b128(wsr, lane16, (e >> 1) * 1024, aux);
                    ring[e % RC] = __builtin_bit_cast(double, u32x2_v{v.x, v.y});
                    ring[(e + 1) % RC] = __builtin_bit_cast(double, u32x2_v{v.z, v.w});
                });
            };
            auto streamed_row = [&](auto rc) {
                constexpr int r = decltype(rc)::value, base = tri(r), NCH = (r + 11) / 12;
                double ac[2] = {V[r], 0.0};
                static_for<0, NCH>([&](auto cc) {
                    constexpr int c = decltype(cc)::value, p0 = 12 * c, K = (r - p0 < 12) ? r - p0 : 12, e0 = base + p0;
                    // what the previous statement (of this row, of the row above, or the resident rows) left requested
                    constexpr int prev = (c > 0) ? e0 - 12 : ((r == NRV) ? E0 : tri(r - 1) + 12 * ((r - 1 + 11) / 12 - 1));
                    request(std::integral_constant<int, prev + RC>{}, std::integral_constant<int, e0 + RC>{});
                    __builtin_amdgcn_sched_barrier(0);
                    double cur[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) cur[k] = ring[(e0 + k) % RC];
                    mfma_chain_v<K>(ac[0], ac[1], cur, V + p0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                const double acc = (r < nt - 1 || accex) ? ac[0] + ac[1] : 0.0;
                V[r] = mfma_zero_v(ring[(base + r) % RC], acc);
                Vlast = V[r];
            };
#ifdef GPMPC_TILES_NO_AGPR
            const bool resident = false;
#else
            const bool resident = nfull >= NRV;
#endif
            if (!resident) {                                      // regime (i)
#ifdef GPMPC_TILES_NO_AGPR
                static_for<0, NT>([&](auto rc) {                  // debug build: every row from the workspace, no prefetch
                    constexpr int r = decltype(rc)::value;
                    if (r < nt) {
                        double cur[r + 1];
#pragma unroll
                        for (int p = 0; p <= r; ++p) cur[p] = tile_load(lane16, tri(r) + p);
                        double ac[2] = {V[r], 0.0};
                        mfma_rowsum<r, false>(ac, cur, V);
                        const double acc = (r < nt - 1 || accex) ? ac[0] + ac[1] : 0.0;
                        V[r] = mfma_zero_v(cur[r], acc);
                        Vlast = V[r];
                    }
                });
#else
                // the complete rows 0 .. nfull-1 are resident (each was loaded when it completed); only the incomplete row
                // nfull streams (its own registers, requested when its turn comes: one exposed L2 latency per step while the
                // factor is this small - requesting it ahead through the ring costs more in spills than it hides).
                // (A chain with early exit: nfull + 1 uniform branches instead of two per possible row - a taken branch of a
                // lone wave costs an instruction fetch, ~30 cycles)
                auto small_from = [&](auto self, auto rc) -> void {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (r <= NRV) {
                        if (r < nfull) {
                            if constexpr (r < NRA) {
                                double ac[2] = {V[r], 0.0};
                                mfma_rowsum<r, true>(ac, At + tri(r), V);
                                const double acc = ac[0] + ac[1];
                                V[r] = mfma_zero_a(At[tri(r) + r], acc);
                                Vlast = V[r];
                            } else if constexpr (r < NRV) {
                                double ac[2] = {V[r], 0.0};
                                mfma_rowsum<r, false>(ac, Bt + (tri(r) - tri(NRA)), V);
                                const double acc = ac[0] + ac[1];
                                V[r] = mfma_zero_v(Bt[tri(r) - tri(NRA) + r], acc);
                                Vlast = V[r];
                            }
                            self(self, std::integral_constant<int, r + 1>{});
                        } else if (part) {                        // r == nfull: the incomplete row
                            double cur[r + 1];
#pragma unroll
                            for (int p = 0; p <= r; ++p) cur[p] = tile_load(lane16, tri(r) + p);
                            double ac[2] = {V[r], 0.0};
                            mfma_rowsum<r, false>(ac, cur, V);
                            const double acc = accex ? ac[0] + ac[1] : 0.0;
                            V[r] = mfma_zero_v(cur[r], acc);
                            Vlast = V[r];
                        }
                    }
                };
                small_from(small_from, std::integral_constant<int, 0>{});
#endif
            } else {                                              // regime (ii)
                // the first RC streamed tiles are requested behind the resident rows, a few tiles in front of each row's
                // MFMAs: 50+ loads issued in one burst by the CU's four waves keep the vector-memory issue port busy for
                // thousands of cycles before the first MFMA
                constexpr int PER = (RC + NRA - 1) / NRA;
                static_for<0, NRA>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    constexpr int e0 = E0 + r * PER, e1 = (e0 + PER < E0 + RC) ? e0 + PER : E0 + RC;
                    request(std::integral_constant<int, e0>{}, std::integral_constant<int, e1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    double ac[2] = {V[r], 0.0};
                    mfma_rowsum<r, true>(ac, At + tri(r), V);
                    const double acc = ac[0] + ac[1];
                    V[r] = mfma_zero_a(At[tri(r) + r], acc);
                    Vlast = V[r];
                    __builtin_amdgcn_sched_barrier(0);
                });
                static_for<NRA, NRV>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    double ac[2] = {V[r], 0.0};
                    mfma_rowsum<r, false>(ac, Bt + (tri(r) - tri(NRA)), V);
                    const double acc = ac[0] + ac[1];
                    V[r] = mfma_zero_v(Bt[tri(r) - tri(NRA) + r], acc);
                    Vlast = V[r];
                    __builtin_amdgcn_sched_barrier(0);
                });
                __builtin_amdgcn_sched_barrier(0);
                TPH(6);
                auto rest = [&](auto self, auto rc) -> void {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (r < NT) {
                        if (r < nt) {                             // uniform; the only way out is the end of the phase
                            streamed_row(rc);
                            self(self, std::integral_constant<int, r + 1>{});
                        }
                    }
                };
                rest(rest, std::integral_constant<int, NRV>{});
            }
            TPH(2);
            // S' += V_r^T V_r, eight tiles per statement (tiles >= nt are zero)
            auto gram_from = [&](auto self, auto gc) -> void {    // early exit (see store_from)
                constexpr int g = decltype(gc)::value;
                if constexpr (g < NT / 8) {
                    if (8 * g < nt) {
                        mfma_rowsum<8, false>(Sh, V + 8 * g, V + 8 * g);
                        self(self, std::integral_constant<int, g + 1>{});
                    }
                }
            };
            gram_from(gram_from, std::integral_constant<int, 0>{});
        }
        TDBG(4, V[0]);
        TDBG(5, V[1]);
        TDBG(6, V[2]);
        TDBG(7, V[3]);
        double zt[T];                                             // base samples of this step: requested here, used after phase F
#pragma unroll
        for (int c = 0; c < T; ++c) zt[c] = a.z[(long)t * a.z_step_stride + (s * G_NY + o) * T + c];
        // ---- phase F: S' to the chains' VALU lanes ---------------------------------------------------------------------
        const double Stot = Sr + (Sh[0] + Sh[1]);
        TDBG(20, Stot);
        TDBG(21, Sr);
        if (live_m) {                                             // a chain without a sample shares its LDS slot with the last live one
            SXm[32 + kq * 4 + jq] = Stot;
            SXm[48 + kq * 4 + jq] = Sr;
        }
        tiles_sync_lds();
        double mu[T], mur[T], S[T][T];
        {
            const int cb[T] = {i0, (i0 + 1) & 3, (i0 + 2) & 3};
            const double il2[D] = {c_il0.get(), c_il1.get()}, os = c_os.get();
#pragma unroll
            for (int b = 0; b < T; ++b) {
                mu[b] = SX[32 + cb[b] * 4 + ycol];
                mur[b] = SX[48 + cb[b] * 4 + ycol];
#pragma unroll
                for (int c = 0; c <= b; ++c) {
                    const double kss = (b == c) ? ((b == 0) ? os : os * il2[b - 1]) : 0.0;
                    const double val = kss - SX[32 + cb[b] * 4 + cb[c]];
                    S[b][c] = val;
                    S[c][b] = val;
                }
            }
        }
        TPH(3);
        // ---- phase G: variance floor, roots, sample (as sample_gp, src/agent.py:629-708) ----------------------------------
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
        double R[T][T], C[T][T], cinv[T];
        bool c_ok;
        {
            double Sn[T][T], rinv[T];
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int c = 0; c < T; ++c) Sn[b][c] = S[b][c] + ((b == c) ? gp.noise[b] : 0.0);
            bool r_ok;
            chol3_pair_fast(Sn, S, C, R, cinv, rinv, c_ok, r_ok);
            if (!r_ok && !seeding) info_acc |= root_small_fast_retry<T>(S, gp.jitter, R);
            if (vo_new) {                                         // value-only label: the 1 x 1 factor, two identity rows
                c_ok = Sn[0][0] > 0.0;
                C[1][0] = C[2][0] = C[2][1] = 0.0;
                C[1][1] = C[2][2] = 1.0;
                cinv[1] = cinv[2] = 1.0;
            }
        }
        double y[T];
#pragma unroll
        for (int b = 0; b < T; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c <= b; ++c) acc = fma(R[b][c], zt[c], acc);
            double yb = acc + mu[b];
            if (all_zero) yb = mu[b];
            const double dlt = yb - mu[b];
            if (dlt * dlt > a.beta * a.beta * var[b]) {
                const double sd = a.beta * sqrt(var[b]);
                yb = fmin(fmax(yb, mu[b] - sd), mu[b] + sd);
            }
            y[b] = yb;
        }
        if (seeding) {                                            // the given labels of the seed point
            const int pt = tt + n_pre;
            const double* ys = seed_full ? a.Y_h0 + ((s * G_NY + o) * (long)a.n_h0 + pt) * T
                                         : a.Y_v0 + ((s * G_NY + o) * (long)a.n_v0 + (pt - a.n_h0)) * T;
#pragma unroll
            for (int b = 0; b < T; ++b) y[b] = ys[b];
        }
        if (l == 0 && live && a.Y && !seeding) {
#pragma unroll
            for (int b = 0; b < T; ++b) a.Y[((s * G_NY + o) * H + t) * T + b] = y[b];
        }

        TPH(4);
        // ---- phase H: append the point (A.9): three rows of the factor, the point's record, its label residuals -------------
        if (seeding || t + 1 < H) {
            if (!c_ok) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
            const int jn = n_pts, tn = n_h >> 2;
            // VALU side: the record, the residuals y - mu_real, the point itself, and the chain's scalars for the MFMA side
            const double P0 = pP[0].get(), P1 = pP[1].get();
            if (ax0) {
                XF[jn * L::XS + l] = P0;
                XF[jn * L::XS + L::XH + l] = P1;
            } else if (ax1) {
                YF[jn * L::YS + (l - N0)] = P0;
                YF[jn * L::YS + L::YH + (l - N0)] = P1;
            }
            if (l < T) YT[jn * 4 + l] = ((l == 0) ? y[0] : ((l == 1) ? y[1] : y[2])) - ((l == 0) ? mur[0] : ((l == 1) ? mur[1] : mur[2]));
            static_for<0, NPS>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if ((jn >> 4) == q) {                             // uniform: the pass the new point belongs to
                    const bool mine = (16 * q + l) == jn;
                    const double o0 = xptP[q][0].get(), o1 = xptP[q][1].get();
                    xptP[q][0].put(mine ? xiP[0].get() : o0);
                    xptP[q][1].put(mine ? xiP[1].get() : o1);
                }
            });
            if (l == 0) {                                         // C (lower, row-major 3x3) and 1/diag
#pragma unroll
                for (int b = 0; b < T; ++b) {
#pragma unroll
                    for (int c = 0; c < T; ++c) SX[64 + 3 * b + c] = (c <= b) ? C[b][c] : 0.0;
                    SX[73 + b] = cinv[b];
                }
            }
            tiles_sync_lds();
            TPS(7);                                               // H1: the point's record, labels, point, C to LDS
            // MFMA side.  Lane (kq, jq) of tile pair (Rt, Pt) is the entry row gi = 4 Rt + jq, column gk = 4 Pt + kq of L.
            // New rows are n_h .. n_h+2; their entry against an OLD column is v of that column = V[Pt] in this very lane.
            const int base = n_h;
            // (the entries of C and 1 / diag this lane will want are read HERE, in front of the store chain below: their LDS latency -
            // seven dependent round trips where the diagonal tiles use them - passes under the stores)
            auto c_read = [&](int Rt, int Pt) -> double {
                const int gi = 4 * Rt + jq, gk = 4 * Pt + kq;
                const int ci = min(max(gi - base, 0), 2), ck = min(max(gk - base, 0), 2);
                return SXm[64 + 3 * ci + ck];
            };
            auto d_read = [&](int Rt, bool rowwise) -> double {
                const int g = 4 * Rt + (rowwise ? kq : jq);
                return SXm[73 + min(max(g - base, 0), 2)];
            };
            const int tn_ = n_h >> 2;
            const double cv_tt = c_read(tn_, tn_), cv_ut = c_read(tn_ + 1, tn_), cv_uu = c_read(tn_ + 1, tn_ + 1);
            const double dr_t = d_read(tn_, true), dc_t = d_read(tn_, false), dr_u = d_read(tn_ + 1, true), dc_u = d_read(tn_ + 1, false);
            auto entry = [&](int Rt, int Pt, double vown, double old, double cval) -> double {
                const int gi = 4 * Rt + jq, gk = 4 * Pt + kq;
                double val = (gk <= gi) ? cval : 0.0;             // both new
                const bool dead = vo_new && gi > base;            // (only read for rows base .. base + 2)
                val = (gk < base) ? (dead ? 0.0 : vown) : val;    // new row, old column
                val = (gi >= base + 3) ? ((gi == gk) ? 1.0 : 0.0) : val;   // rows that do not exist yet: identity
                val = (gi < base) ? old : val;                    // old rows keep what they had
                return val;
            };
            // which tile row does this lane's new row belong to (columns i0.. stay in tile tn, the wrapped ones open tn+1)
            const bool isnew = jq != ycol;                        // column jq carries the new row (n_h + c), c = (jq - i0) & 3 <= 2
            const bool deadrow = vo_new && (((jq - i0) & 3) >= 1);     // a derivative row of a value-only point
            const int myrow = (jq >= i0) ? tn : tn + 1;
            const unsigned rowtile = (unsigned)tri(myrow);       // per-lane: the two target tile rows differ
            // off-diagonal tiles against complete old tile rows p < tn: -v
            {
                auto store_from = [&](auto self, auto pcn) -> void {     // early exit: tn branches instead of NT
                    constexpr int p = decltype(pcn)::value;
                    if constexpr (p < NT) {
                        if (p < tn) {
                            // (unconditional: a lane that holds no new row names an offset beyond the descriptor's end - the store is
                            // dropped without traffic and without an exec-mask region per tile)
                            const unsigned e = rowtile + p;
                            tile_store(isnew ? lane16 + (e >> 1) * 1024u + (e & 1u) * 8u : 0x7ffff000u, 0, deadrow ? 0.0 : -V[p]);
                            self(self, std::integral_constant<int, p + 1>{});
                        }
                    }
                };
                store_from(store_from, std::integral_constant<int, 0>{});
            }
            TPS(8);                                               // H2: the new rows' tiles against the complete old tile rows (stores)
            // the tile pairs that involve tile tn itself (old rows of the incomplete tile) and the new tile tn+1
            // (a search for V[tn] over the register array turns it into a scratch array: hipcc makes a table lookup of it)
            const double Vtn = ((n_h & 3) != 0) ? Vlast : 0.0;    // tile row tn exists only if it is incomplete
            const bool wraps = i0 >= 2;                           // rows n_h .. n_h+2 reach into tile tn+1
            // diagonal tile tn: U = L^T (natural), its row / column scalings
            auto diag_scal = [&](int Rt, double oldv, bool rowwise, double cv_) -> double {
                const int g = 4 * Rt + (rowwise ? kq : jq);
                double v = (g >= base + 3) ? 1.0 : cv_;
                v = (g < base) ? oldv : v;
                return v;
            };
            auto inverse_tile = [&](double U, double drow, double dcol) -> double {
                // U = Dg (I - M), M strictly upper:  U^-1 = (I + M)(I + M^2) Dg^-1   (M^4 = 0)
                const double M = Inat - drow * U;
                const double Mt = mfma_zero_v(M, Inat);           // M^T
                const double M2 = mfma_zero_v(Mt, M);             // M M
                const double Pq = mfma_zero_v(Inat + Mt, Inat + M2);   // (I + M)(I + M^2)
                return Pq * dcol;
            };
            {
                // old 1/diag of the incomplete tile, row-wise and column-wise copies travel in dcur through LDS-free
                // lookups: dcur holds 1/d on the diagonal lanes only; rebuild both scalings from it with two MFMAs
                const double ones = 1.0;
                const double dcur = dcP.get();
                const double drow_old = mfma_zero_v(dcur, ones);  // [k][i] = 1 / d_k
                const double dcol_old = mfma_zero_v(ones, dcur);  // [k][i] = 1 / d_i
                const double U = entry(tn, tn, Vtn, ucP.get(), cv_tt);
                const double drow = diag_scal(tn, drow_old, true, dr_t), dcol = diag_scal(tn, dcol_old, false, dc_t);
                const double Gt = inverse_tile(U, drow, dcol);
                TDBG(26, U);
                TDBG(27, drow);
                TDBG(28, dcol);
                TDBG(29, Gt);
                tile_store(lane16, tri(tn) + tn, Gt);
                double Unext = U, dnext = (kq == jq) ? drow : 0.0;
                if (wraps) {
                    const double X = entry(tn + 1, tn, Vtn, 0.0, cv_ut);
                    const bool newrow1 = jq <= i0 - 2;            // rows of tile tn+1 that exist now
                    if (newrow1) tile_store(lane16, tri(tn + 1) + tn, -X);
                    const double U1 = entry(tn + 1, tn + 1, 0.0, 0.0, cv_uu);
                    const double drow1 = diag_scal(tn + 1, 1.0, true, dr_u), dcol1 = diag_scal(tn + 1, 1.0, false, dc_u);
                    const double G1 = inverse_tile(U1, drow1, dcol1);
                    TDBG(30, U1);
                    TDBG(31, drow1);
                    TDBG(32, dcol1);
                    TDBG(33, G1);
                    TDBG(34, X);
                    tile_store(lane16, tri(tn + 1) + tn + 1, G1);
                    Unext = U1;
                    dnext = (kq == jq) ? drow1 : 0.0;
                }
                // the incomplete tile after this step: tn while i0 == 0 (rows 0..2 of a fresh tile), else tn + 1 (fresh
                // when nothing wrapped: identity)
                if (i0 == 1) {                                    // tile tn is complete now, the next tile is untouched
                    Unext = Inat;
                    dnext = (kq == jq) ? 1.0 : 0.0;
                }
                ucP.put(Unext);
                dcP.put(dnext);
            }
            TPS(9);                                               // H3: the incomplete diagonal tile(s): entries, inverses, stores
            // a tile row that became complete moves into its AGPRs
            if (i0 >= 1 && tn < NRA) {                            // (uniform) a resident row completed: its stores first -
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this lane re-reads only its own slots
                static_for<0, NRA>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if (r == tn) load_row_agpr<r + 1, (tri(r) & 1)>(wsr, lane16, (tri(r) >> 1) * 1024, At + tri(r));   // loads + their wait: one statement
                });
            }
            if (i0 >= 1 && tn >= NRA && tn < NRV) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                static_for<NRA, NRV>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if (r == tn) {
#pragma unroll
                        for (int p = 0; p <= r; ++p) Bt[tri(r) - tri(NRA) + p] = tile_load(lane16, tri(r) + p);
                    }
                });
            }
            n_pts += 1;
            TPS(10);                                              // H4: a completed tile row back into its AGPRs
        }

        TPH(5);
        // ---- state hand-over ---------------------------------------------------------------------------------------------
        if (!seeding) {
            double x[NX];
#pragma unroll
            for (int d = 0; d < NX; ++d) x[d] = xP[d].get();
            if (ENV == GPMPC_ENV_PENDULUM1D) {
                const double x0n = x[0] + x[1] * env_dt;
                x[1] = x[1] + y[0];
                x[0] = x0n;
            } else {
                // value samples of the three outputs: lane 0 of DPP rows 0, 1, 2
                const double g0 = readlane_f64(y[0], 0), g1 = readlane_f64(y[0], 16), g2 = readlane_f64(y[0], 32);
                constexpr int I2 = (NX > 2) ? 2 : 0, I3 = (NX > 3) ? 3 : 0;
                const double vv = x[I3];
                x[0] = x[0] + vv * g0;
                x[1] = x[1] + vv * g1;
                x[I2] = x[I2] + vv * g2;
                x[I3] = x[I3] + uP.get() * env_dt;
            }
#pragma unroll
            for (int d = 0; d < NX; ++d) xP[d].put(x[d]);
        }
    }

    TPH_STORE;
    if (l == 0 && live && o == 0) {
#pragma unroll
        for (int d = 0; d < NX; ++d) a.X_traj[(s * NX + d) * (H + 1) + H] = xP[d].get();
    }
    if (G_NY == 1) {
        if (l == 0 && live) a.info[s] = info_acc;
    } else {
        const int i1 = __builtin_amdgcn_readlane(info_acc, 16), i2 = __builtin_amdgcn_readlane(info_acc, 32);
        if (lane == 0) a.info[s] = info_acc | i1 | i2;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side.  Instantiated for the reference's grids (pendulum1D 4 x 9, car 5 x 9: n_data_x x n_data_u of the shipped YAMLs)
// and one size up each (5 x 9 / 6 x 9), with 32 tile rows (128 label rows per chain: H <= 43) or 48 (192: H <= 65, the
// shipped car H = 50 included); anything else stays with the one-chain-per-wave kernels.
// ---------------------------------------------------------------------------------------------------------------
static int tiles_mode() {                                        // 0 auto, 1 forced, -1 disabled
    if (g_rollout_pin != GPMPC_KERNEL_AUTO) return (g_rollout_pin == GPMPC_KERNEL_TILES) ? 1 : -1;
    const char* e = std::getenv("GPMPC_ROLLOUT_TILES");
    if (!e) return 0;
    return (e[0] == '1') ? 1 : ((e[0] == '0') ? -1 : 0);
}

// tile rows for H steps behind n_pre conditioning-only points (three row slots per point, value-only points included)
static int tiles_nt(int H, int n_pre = 0) {
    const int n = 3 * (n_pre + H - 1);
    return (n <= 128) ? 32 : ((n <= 160) ? 40 : ((n <= 192) ? 48 : 0));
}

template <int N0, int N1>
static size_t tiles_lds_bytes(int g_ny, int H, int n_pre) {
    const int gl = (g_ny == 1) ? 4 : 3;
    return (size_t)gl * TilesLds<N0, N1>::per_chain(n_pre + H - 1 > 1 ? n_pre + H - 1 : 1) * sizeof(double);
}

static size_t tiles_lds_for(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int H, int n_pre) {       // 0: shape not instantiated
    const int n0 = gp->grid_n0, n1 = gp->grid_n1;
    if (env->env_id == GPMPC_ENV_PENDULUM1D && gp->g_ny == 1 && n1 == 9) {
        if (n0 == 4) return tiles_lds_bytes<4, 9>(1, H, n_pre);
        if (n0 == 5) return tiles_lds_bytes<5, 9>(1, H, n_pre);
    }
    if (env->env_id == GPMPC_ENV_CAR_RESIDUAL && gp->g_ny == 3 && n1 == 9) {
        if (n0 == 5) return tiles_lds_bytes<5, 9>(3, H, n_pre);
        if (n0 == 6) return tiles_lds_bytes<6, 9>(3, H, n_pre);
    }
    return 0;
}

// n_h0 / n_v0 > 0: a seeded call (gpmpc_rollout_seeded without a kept factor state) - the seed points are conditioning-only
// passes of the same step body; hall_tasks == 1: value-only points keep three row slots (see the kernel)
bool rollout_tiles_eligible(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int mode, int hall_tasks, int H, int64_t Ns,
                            int n_h0, int n_v0) {
    const int md = tiles_mode();
    const int n_pre = n_h0 + n_v0;
    if (md < 0) return false;
    const char* e = std::getenv("GPMPC_DISABLE_FAST_ROLLOUT");
    if (e && e[0] == '1') return false;
    const char* eg = std::getenv("GPMPC_DISABLE_GRID_ROOT");
    if (eg && eg[0] == '1') return false;
    if (mode != GPMPC_MODE_RECONDITIONED || gp->T != 3 || gp->D != 2 || gp->real_has_grad) return false;
    if (!(hall_tasks == 3 || hall_tasks == 1)) return false;
    if (!plan_has_grid_root(gp->grid_n0, gp->grid_n1, gp->real_has_grad)) return false;
    if ((H < 2 && n_pre == 0) || H < 1 || tiles_nt(H, n_pre) == 0) return false;
    if (tiles_nt(H, n_pre) > 32 && !((env->env_id == GPMPC_ENV_PENDULUM1D && gp->grid_n0 == 4) || (env->env_id == GPMPC_ENV_CAR_RESIDUAL && gp->grid_n0 == 5)))
        return false;
    const size_t lds = tiles_lds_for(gp, env, H, n_pre);
    if (lds == 0 || lds > 160 * 1024 - 64) return false;
    if (md > 0) return true;
    // A wave carries four chains and takes ~1.6-1.9x as long as a wave of the one-chain-per-wave kernel: the tuned kernel
    // wins while it needs ONE round of the chip (pendulum: 1024 chains, one per SIMD; car: 256 samples, one three-wave
    // workgroup per CU) and loses from its second round on (tools/debug/tiles_threshold.py, sustained clocks: pendulum
    // Ns = 1024 0.109 vs 0.170 ms, 1536 0.214 vs 0.181, 3072 0.323 vs 0.208; car Ns = 256 0.216 vs 0.306, 384 0.425 vs
    // 0.324, 768 0.639 vs 0.380).  Shapes the tuned kernel does not take (other grids, 3 (H - 1) > 128) fall to the generic
    // kernel, 4-20x slower: there the tiled kernel is taken from 256 chains on.
    const int64_t chains = Ns * gp->g_ny;
    const bool tuned_alt = n_pre == 0 && rollout_fast_eligible(gp, env, mode, hall_tasks, H);
    return tuned_alt ? (chains > (gp->g_ny == 1 ? 1024 : 768)) : (chains >= 256);
}

size_t rollout_tiles_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int H, int n_pre) {
    const int nt = tiles_nt(H, n_pre) ? tiles_nt(H, n_pre) : 32;
    const int64_t waves = (gp->g_ny == 1) ? (Ns + 3) / 4 : Ns;
    return (size_t)waves * tri(nt) * 64 * sizeof(double);
}

template <int N0, int N1, int ENV, int NT>
static int launch_tiles(RolloutArgs& args, int g_ny, hipStream_t st) {
    const size_t lds = tiles_lds_bytes<N0, N1>(g_ny, args.H, args.n_h0 + args.n_v0);
    const long nblk = (g_ny == 1) ? (args.Ns + 3) / 4 : args.Ns;
    const bool seed = (args.n_h0 + args.n_v0 > 0) || args.hall_tasks == 1;
    auto k = seed ? rollout_tiles_kernel<N0, N1, ENV, NT, true> : rollout_tiles_kernel<N0, N1, ENV, NT, false>;
    GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(64), lds, st, args);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

int rollout_tiles_launch(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, RolloutArgs& args, void* ws, size_t ws_bytes,
                         hipStream_t st) {
    if (!ws || ws_bytes < rollout_tiles_workspace_bytes(gp, args.Ns, args.H, args.n_h0 + args.n_v0))
        return fail(GPMPC_E_WORKSPACE, "gpmpc_rollout: workspace too small");
    const int nt = tiles_nt(args.H, args.n_h0 + args.n_v0), n0 = gp->grid_n0;
    args.ws = (double*)ws;
    args.ws_chain_stride = (long)tri(nt) * 64;                   // doubles per wave
    // the shipped grids with every tile-row count, the one-size-up grids with 32 tile rows
    if (env->env_id == GPMPC_ENV_PENDULUM1D) {
        if (n0 == 4 && nt == 32) return launch_tiles<4, 9, GPMPC_ENV_PENDULUM1D, 32>(args, 1, st);
        if (n0 == 4 && nt == 40) return launch_tiles<4, 9, GPMPC_ENV_PENDULUM1D, 40>(args, 1, st);
        if (n0 == 4 && nt == 48) return launch_tiles<4, 9, GPMPC_ENV_PENDULUM1D, 48>(args, 1, st);
        if (n0 == 5 && nt == 32) return launch_tiles<5, 9, GPMPC_ENV_PENDULUM1D, 32>(args, 1, st);
    } else {
        if (n0 == 5 && nt == 32) return launch_tiles<5, 9, GPMPC_ENV_CAR_RESIDUAL, 32>(args, 3, st);
        if (n0 == 5 && nt == 40) return launch_tiles<5, 9, GPMPC_ENV_CAR_RESIDUAL, 40>(args, 3, st);
        if (n0 == 5 && nt == 48) return launch_tiles<5, 9, GPMPC_ENV_CAR_RESIDUAL, 48>(args, 3, st);
        if (n0 == 6 && nt == 32) return launch_tiles<6, 9, GPMPC_ENV_CAR_RESIDUAL, 32>(args, 3, st);
    }
    return fail(GPMPC_E_UNSUPPORTED, "rollout_tiles: shape not instantiated");
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_tiles(double* out /*[host] 4096*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_tiles_dbg), 64 * 64 * sizeof(double)));
    return GPMPC_OK;
}
