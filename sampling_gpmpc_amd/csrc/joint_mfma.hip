// joint_test_mfma_kernel: the test rows of the joint draw (mode "J") on the FP64 matrix pipe (gfx950).
//
// What it computes, per (sample, output) chain (reference src/agent.py:629-641, the model_i(x) call of
// every SQP iteration, src/solver.py:84-94):
//      X   = L^-1 [ K_o* | y ]          n_o x (m T + 1)    L = chol(K_oo + noise) = [[L_rr 0] [L_hr L_hh]], y = labels
//      mu  = X[:, :mT]^T X[:, mT]       posterior mean      (X[:, mT] = w = L^-1 y)
//      S   = K** - X[:, :mT]^T X[:, :mT] posterior covariance
// The factor L is NOT formed here: the plan holds L_rr, the factor cache the hallucinated rows [L_hr | L_hh] (joint_kernel,
// JOINT_PHASE_FACTOR, has extended it by this call's new rows before this kernel starts).
//
// Mapping.  One workgroup of EIGHT waves per chain; wave I owns the 16 test columns 16 I .. 16 I + 15 (column m T is w) and
// keeps its WHOLE 16-column block of X - up to 26 tiles of 16 conditioning slots, 4 FP64 registers per tile - in registers as
// v_mfma_f64_16x16x4_f64 accumulators: the right-looking block substitution
//      X_j = Linv_jj acc_j ;   acc_k -= L_kj X_j   (k > j)
// never writes X anywhere.  The D layout of the instruction (register v, lane l: row 4 v + (l >> 4), column l & 15) IS its B
// layout (K-slice v: k = l >> 4), so X_j feeds the next products as it stands.  The 16 rows of a tile are LABELLED so that
// register v, lane-row kk is conditioning slot 16 j + 4 kk + v: the A operand of K-slice v, lane (i, kk), is then
// L[row pi(i)][column 4 kk + v], v = 0..3 = FOUR CONSECUTIVE doubles of a factor row - two 16-byte pieces.  Tiles sit in LDS
// as [half h][kk][i] x 16 bytes (piece (h, kk, i) = columns 4 kk + 2 h, + 1 of row pi(i), pi = the 4 x 4 transpose of the row
// index): a wave reads a tile with two conflict-free ds_read_b128, and a wave WRITES a tile with two
// global_load_lds_dwordx4 (lane (i, kk) names the 16 bytes of piece (h, kk, i) in the factor cache; the LDS side is
// lane-linear): the L tiles go HBM/L2 -> LDS without passing through registers, every tile once per chain for all eight waves
// (0.66 MB per chain at n_o = 405 instead of the 5.7 MB the VALU update re-reads), a 32-tile ring, 16 tiles per barrier.
// The inverted diagonal tiles are made once per chain in the prologue (one inverse column per lane, four tiles per wave).
// Kernel entries: one lane per (conditioning POINT, test point) pair - one exponential for up to T x T entries - through a
// 32-slot LDS buffer into the accumulators.  Afterwards the Gram products X_I^T X_J run on the same registers in ONE pass: five
// accumulator tiles per wave start at -K** (formed in place), the A operand is the wave's own X read where it lies, the B
// operands are the other waves' tiles through LDS; the result is negated on the way out (the running value shrinks towards S as
// in the sequential form).
// The same kernel extends the factor (JOINT_MFMA_FACTOR: columns = the new hallucinated rows, out = their entries against the
// old columns + the Schur complement) and handles conditioning sets of more than 26 tiles in two launches (JOINT_MFMA_TEST_TOP /
// _BOTTOM, joint_args.hpp).  DESIGN.md 4.4c has the measurements, the register discipline and the MFMA source-read hazard.
#include <type_traits>

#include "joint_args.hpp"

namespace gpmpc {

typedef double jm_d4 __attribute__((ext_vector_type(4)));
typedef double jm_d2 __attribute__((ext_vector_type(2)));
typedef double jm_d8 __attribute__((ext_vector_type(8)));
typedef double jm_d16 __attribute__((ext_vector_type(16)));

#ifndef GPMPC_JM_NT
#define GPMPC_JM_NT 26
#endif
constexpr int JM_NT = GPMPC_JM_NT;                // slot tiles a wave can hold (8 registers each)
constexpr int JM_RT = 6;                          // off-diagonal tiles with real rows (tile rows k < kmin <= 4)
constexpr int JM_RING = 32;                       // ring of streamed tiles (2 KB each)
constexpr int JM_CH = 16;                         // tiles per chunk = per barrier
constexpr int JM_NTD = JM_NT * (JM_NT - 1) / 2;   // streamed tiles at most
constexpr int JM_NW = 8;                          // waves per workgroup = 16-column tiles of the test block (two waves per SIMD)
constexpr int JM_NCT = JM_NW;                     // column tiles
constexpr int JM_THREADS = JM_NW * 64;
constexpr int JM_KCH = 32;                        // conditioning slots per kernel-entry chunk
constexpr int JM_COLS = JM_NCT * 16;

constexpr int JM_STASH = 4;                       // tiles 22..25 of every wave go to LDS before the Gram phase (32 registers come free)
constexpr int JM_SMEM_DOUBLES = JM_NT * 256 + (JM_NW * JM_STASH - JM_NT) * 256 + JM_RING * 256 + JM_RT * 256 + 64 + JM_NT * 16 * 2 + JM_COLS * 2 + JM_NT * 16;
constexpr int JM_SMEM_SHORTS = (JM_NT * 16 + 8) + (JM_NTD + 7) + 32 + 32 + (JM_COLS + 8) + (JM_NT * 16 + 2 * JM_COLS) / 2 + 2 * (4 * JM_NW + 4);
constexpr size_t JM_SMEM_BYTES = (size_t)JM_SMEM_DOUBLES * 8 + (size_t)JM_SMEM_SHORTS * 2 + 64;

static_assert(JM_NTD + 7 >= (JOINT_MFMA_SPLIT / 16) * (JOINT_MFMA_BOTTOM_MAX / 16) + (JOINT_MFMA_BOTTOM_MAX / 16) * (JOINT_MFMA_BOTTOM_MAX / 16 - 1) / 2,
              "the stream table holds the BOTTOM launch's tiles");
static_assert(JOINT_MFMA_SPLIT == JM_NT * 16 && (8 + 2 * JM_NW) * 256 <= JM_NW * JM_STASH * 256,
              "TOP fills every tile; BOTTOM's <= 8 inverted diagonal tiles and the waves' X slots share the linv region");
__device__ __attribute__((aligned(16))) double g_jm_zero[4] = {0.0, 0.0, 0.0, 0.0};
// rows of a 16 x 16 identity, each followed by zeros: row i starts at g_jm_eye[i * 16] (the pad rows of the last diagonal tile)
__device__ __attribute__((aligned(16))) double g_jm_eye[16 * 16] = {
    1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
__device__ long long g_jm_phase[4][40];       // per launch mode (JOINT_MFMA_*): the last launch of that mode          // [20..27]: inside the prologue, [28..31]: inside the kernel entries (wave 0)

#ifdef GPMPC_PHASE_TIMERS
#define JMPH(idx) do { const long long _n = __builtin_readcyclecounter(); jph[idx] += _n - jt; jt = _n; } while (0)
#define JMPHS(idx) do { const long long _n = __builtin_readcyclecounter(); jphs[idx] += _n - jts; jts = _n; } while (0)
#define JMPHP(idx) do { const long long _n = __builtin_readcyclecounter(); jpp[idx] += _n - jtp; jtp = _n; } while (0)
#define JMPHE(idx) do { const long long _n = __builtin_readcyclecounter(); jpe[idx] += _n - jtp; jtp = _n; } while (0)
#else
#define JMPHP(idx)
#define JMPHE(idx)
#define JMPH(idx)
#define JMPHS(idx)
#endif

__device__ __forceinline__ int jm_pi(int i) { return 4 * (i & 3) + (i >> 2); }
// offset (doubles) of element (row r, column c) of a 16 x 16 tile in the LDS tile format
__device__ __forceinline__ int jm_off(int r, int c) { return ((c & 3) >> 1) * 128 + (jm_pi(r) + 16 * (c >> 2)) * 2 + (c & 1); }

__device__ __forceinline__ jm_d4 jm_mfma(double a, double b, jm_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// The accumulate chains are inline asm with TIED accumulators: left to the builtin, hipcc gives the MFMA results fresh registers
// and copies them back into the (switch-selected) accumulator tiles - 700 spilled registers at 26 tiles.  hipcc pads nothing
// around an asm MFMA, so every statement carries its own wait states (GCNHazardRecognizer's gfx940 table for the 16-pass
// v_mfma_f64_16x16x4_f64): VALU write -> MFMA source 2 (the leading s_nop 1); SrcC = the previous MFMA's own vDst, same opcode:
// 0 (back to back); MFMA result -> SrcA/B of a later MFMA or any VALU read 11, -> a memory / LDS store 18 (the trailing s_nops:
// whatever hipcc schedules behind the statement may read the tile).
#include "joint_mfma_gen.inc"

// the lane id, formed HERE (volatile: hipcc neither hoists it out of a loop nor keeps - i.e. spills - an earlier copy)
__device__ __forceinline__ int jm_lane_now() {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    return ln;
}

__device__ __forceinline__ void jm_glds16(const double* src, double* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// kern_entry (gpmpc_device.hpp) with per-LANE tasks: selects instead of q[a - 1] / inv_l2[a - 1] - a register array indexed by
// a lane-dependent value lives in scratch memory (the entries and the Gram tiles spent most of their time in scratch loads)
__device__ __forceinline__ double jm_kern_entry(double q0, double q1, double k, double il0, double il1, int a, int b) {
    const double qa = (a == 1) ? q0 : q1, qb = (b == 1) ? q0 : q1;
    if (a == 0) return (b == 0) ? k : k * qb;
    if (b == 0) return -k * qa;
    double v = -qa * qb;
    if (a == b) v += (a == 1) ? il0 : il1;
    return k * v;
}
// exp(x), x <= 0 (one shared, small implementation: the library exp is inlined at every call site)
__device__ __forceinline__ double jm_exp_neg(double x) {
    const double xa[1] = {x};
    double e[1];
    expn_neg<1>(xa, e);
    return e[0];
}
// (the Gram accumulators are read again as SrcC of the same opcode only - no wait states between the products; jm_settle() stands
// in front of their first other reader)
__device__ __forceinline__ void jm_settle(jm_d4& s0) {      // 18 wait states behind the MFMA that wrote s0 (store / VALU readers follow)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1" : "+v"(s0));
}

// mode 0 (JOINT_MFMA_TEST):   columns = the m T test slots + the label column; conditioning slots = real + all hallucinated;
//                             out: mean, S = K** - V^T V.
// mode 1 (JOINT_MFMA_FACTOR): columns = the n_ho - n_c NEW hallucinated rows; conditioning slots = real + the n_c cached ones;
//                             out: the new rows' entries against the old columns, X^T, into the factor cache, and the Schur
//                             complement K_nn + noise - X^T X (joint_kernel, JOINT_PHASE_CHOL, factorises it).
template <int T>
__global__ __launch_bounds__(JM_THREADS) void joint_test_mfma_kernel(const JointArgs a) {
    constexpr int D = 2;
    extern __shared__ __attribute__((aligned(16))) double jm_smem[];
    double* linv = jm_smem;                       // [JM_NT][256]  inverted diagonal tiles, tile format; + 6 tiles: [JM_NW][JM_STASH][256] stash (Gram phase)
    double* ring = linv + JM_NW * JM_STASH * 256;            // [JM_RING][256] streamed tiles; second half = kernel-entry buffer; all = Gram exchange
    double* realt = ring + JM_RING * 256;         // [JM_RT][256]  off-diagonal tiles with real rows
    double* yr = realt + JM_RT * 256;             // [64] L_rr w_r: the right-hand side that reproduces w_r on the real slots
    double* rptx = yr + 64;                       // [JM_NT * 16][2] input point of every slot run
    double* cptx = rptx + JM_NT * 16 * 2;         // [JM_COLS][2] input point of every column run
    double* ylab = cptx + JM_COLS * 2;            // [JM_NT * 16] label of every hallucinated slot (test mode: the label column's entries)
    short* run_start = reinterpret_cast<short*>(ylab + JM_NT * 16);     // [JM_NT * 16 + 8] first slot of every conditioning point's run
    unsigned short* stab = reinterpret_cast<unsigned short*>(run_start + JM_NT * 16 + 8);  // [JM_NTD + 7] streamed tile -> (k << 8) | j
    short* first_run = reinterpret_cast<short*>(stab + JM_NTD + 7);  // [32] run that contains slot 32 c
    short* starts_before = first_run + 32;                           // [32] runs that start before slot 32 c
    short* crun_start = starts_before + 32;                          // [JM_COLS + 8] first column of every column run
    signed char* stask = reinterpret_cast<signed char*>(crun_start + JM_COLS + 8);   // [JM_NT * 16] task of every slot
    signed char* ctask = stask + JM_NT * 16;                         // [JM_COLS] task of every column (-1: the label column)
    unsigned char* crun_of = reinterpret_cast<unsigned char*>(ctask + JM_COLS);     // [JM_COLS] column run of every column
    int* wtot = reinterpret_cast<int*>(crun_of + JM_COLS);           // [4][JM_NW]
    double* kbuf = ring + JM_CH * 256;            // [JM_KCH][JM_COLS]

    const GpParams& gp = a.gp;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // (wv: an SGPR)
    const bool fmode = a.mfma_mode == JOINT_MFMA_FACTOR;
    const bool top = a.mfma_mode == JOINT_MFMA_TEST_TOP, bottom = a.mfma_mode == JOINT_MFMA_TEST_BOTTOM;
    const int n_r = gp.n_r, Tr = gp.real_has_grad ? T : 1;
    const int m = a.m, mT = m * T;
    const int n_hc = fmode ? a.n_c : a.n_ho;      // hallucinated slots conditioned on
    // the slots of THIS launch are the LOCAL slots 0 .. n_o - 1 = the problem's slots sb .. sb + n_o - 1 (sb > 0: the BOTTOM launch
    // of a split conditioning set, whose first sb slots - nx tiles - the TOP launch has solved: their X tiles come from a.xbuf)
    const int sb = bottom ? JOINT_MFMA_SPLIT : 0;
    const int n_o = top ? JOINT_MFMA_SPLIT : n_r + n_hc - sb;
    const int nx = sb >> 4;
    const int ncols = fmode ? a.n_ho - a.n_c : mT + 1;
    const int nt = (n_o + 15) >> 4;               // slot tiles
    const int kmin = bottom ? 0 : (n_r + 15) >> 4;   // tile rows < kmin hold real rows: staged through registers
    const int ncta = (ncols + 15) >> 4;           // column tiles in use
    const int I0 = wv;                            // this wave's column tile
    const bool active = I0 < ncta;
    const long chain = a.chain0 + blockIdx.x;
    const long s = chain / gp.g_ny;
    const int o = (int)(chain - s * gp.g_ny);
    const double* Lrr = plan_L(a.plan, gp, o);
    const double* w_r = plan_w(a.plan, gp, o);
    double* fc = a.fcache + (chain - a.fc_chain_base) * a.fc_stride;
    const int CS = a.fc_cs;
    const double* Xh = a.X_h ? a.X_h + chain * (long)a.n_h * D : nullptr;
    const double* Yh = a.Y_h ? a.Y_h + chain * (long)a.n_h * T : nullptr;
    const double* Xs = a.X_s + chain * (long)m * D;
    double il2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) il2[d] = gp.inv_l2[o][d];
    const double os = gp.os[o];
    const double noise0 = gp.noise[0], noise1 = gp.noise[T > 1 ? 1 : 0], noise2 = gp.noise[T > 2 ? 2 : 0];
#ifdef GPMPC_PHASE_TIMERS
    long long jph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long jphs[6] = {0, 0, 0, 0, 0, 0};       // inside the substitution: diagonal steps, hand-overs + first tiles, runs; of the hand-overs: own DMA wait, barrier, requests
    long long jt = __builtin_readcyclecounter();
    long long jts = jt;
    long long jpp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long jpe[4] = {0, 0, 0, 0};
    long long jtp = jt;
#endif

    auto Lel = [&](int r, int c) -> double {      // factor entry (r, c), c <= r < n_o
        return (r < n_r) ? Lrr[(long)r * n_r + c] : fc[(long)(r - n_r) * CS + c];
    };

    auto Lptr = [&](int r, int c) -> const double* {      // &L(r, c) for any r, c >= 0: clamped into the factor
        const int rr = min(r, n_o - 1), cc = min(c, rr);
        return (rr < n_r) ? Lrr + (long)rr * n_r + cc : fc + (long)(rr - n_r) * CS + cc;
    };
    // the (at most four) diagonal tiles with real rows and the off-diagonal tiles with real rows go through registers (the plan's rows
    // are not 16-byte aligned): their loads are REQUESTED here, in front of the stream table, and stored behind it
    constexpr int NVD = 4 * 256 / JM_THREADS, NVR = JM_RT * 256 / JM_THREADS;
    double vd[NVD], vr[NVR];
#pragma unroll
    for (int it = 0; it < NVD; ++it) {
        const int e = tid + it * JM_THREADS;
        const int tj = e >> 8, i = (e >> 4) & 15, c = e & 15;
        vd[it] = *Lptr(16 * tj + i, 16 * tj + c);
    }
#pragma unroll
    for (int it = 0; it < NVR; ++it) {
        const int e = tid + it * JM_THREADS;
        const int ti = e >> 8, i = (e >> 4) & 15, c = e & 15;
        const int k = (ti < 1) ? 1 : ((ti < 3) ? 2 : 3);
        const int j = ti - k * (k - 1) / 2;
        vr[it] = *Lptr(16 * k + i, 16 * j + c);
    }
    // ---- table of the STREAMED tiles (k, j), k >= max(j + 1, kmin): rows that are all hallucinated slots, in consumption order
    // (column by column); the few tiles with real rows (k < kmin) sit in `realt` ------------------------------------------------
    // (closed form of the start of column jj's tiles: the columns j' < kmin - 1 have nt - kmin streamed tiles each, column j' >= kmin - 1
    // has nt - 1 - j'; one thread per (column, tile) - the first version let thread jj write column jj's entries one by one: 18 k cycles)
    auto stab_base = [&](int jj) -> int {
        const int c = min(jj, kmin - 1), n = jj - c;             // n columns j' = kmin - 1 .. jj - 1
        return c * max(0, nt - kmin) + n * (nt - 1) - (n * (2 * (kmin - 1) + n - 1)) / 2;
    };
    // BOTTOM: first the tiles against the nx solved columns (all nt tile rows of a column, column by column; bit 7 of the entry
    // marks them), then the triangle of the launch's own tiles
    const int ntd = bottom ? nx * nt + nt * (nt - 1) / 2 : stab_base(nt);
    if (bottom) {
        for (int e = tid; e < nx * nt; e += JM_THREADS) {
            const int jx = e / nt, k = e - jx * nt;
            stab[e] = (unsigned short)((k << 8) | 0x80 | jx);
        }
        for (int e = tid; e < nt * 32; e += JM_THREADS) {
            const int jj = e >> 5, k = jj + 1 + (e & 31);
            if (k < nt) stab[nx * nt + jj * (nt - 1) - (jj * (jj - 1)) / 2 + (e & 31)] = (unsigned short)((k << 8) | jj);
        }
    } else {
        for (int e = tid; e < nt * 32; e += JM_THREADS) {
            const int jj = e >> 5, k = max(jj + 1, kmin) + (e & 31);
            if (k < nt) stab[stab_base(jj) + (e & 31)] = (unsigned short)((k << 8) | jj);
        }
    }
    __syncthreads();

    // chunk c of the stream -> ring slots (c & 1) * 16 ..: the waves 0..3 - one per SIMD - move four tiles each.  Lane (i, kk) names the
    // 16-byte pieces (h, kk, i) of a tile: row 16 k + pi(i), columns 16 j + 4 kk + 2 h (+ 1).  Everything lane-dependent is formed ONCE (two
    // registers); a tile adds a uniform offset.  (The first version formed the address per tile from spilled values: the reload's
    // s_waitcnt vmcnt(0) stood right behind the first tile's loads and waited for them - 5 k cycles per hand-over.)
    // Why four waves and not all eight: the requests cost a wave 0.8-1.6 k cycles per hand-over (table look-up, addresses, zero-page
    // select), all eight waves paid them at the same moment - right behind the barrier - and the matrix pipe stood still meanwhile
    // (35 k cycles per chain on the slowest wave).  Now the other wave of every SIMD starts its products at once and has the pipe to
    // itself while its partner requests.
    auto issue_chunk = [&](int c) {
        if (wv >= 4) return;                      // (uniform)
        // (the lane-dependent part is RE-formed at every call from the lane id - six VALU instructions; kept in registers across the
        // substitution, where 16 VGPRs are free, hipcc spills it and the reload is a round trip to scratch memory per hand-over)
        const int ln = jm_lane_now();
        const int dma_row = jm_pi(ln & 15);                     // row of the tile this lane reads
        const double* dma_lane = fc + (long)(dma_row - n_r) * CS + 4 * (ln >> 4);
        const int q0 = c * JM_CH + wv * 4;
        // the wave's four table entries with one LDS read (8-byte aligned: q0 is a multiple of four; entries beyond ntd are not used)
        const unsigned long long kj4 = *reinterpret_cast<const unsigned long long*>(stab + q0);
        const unsigned kj_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kj4);
        const unsigned kj_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kj4 >> 32));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u;
            if (q < ntd) {                        // (uniform)
                const int kj = (int)(((u < 2 ? kj_lo : kj_hi) >> (16 * (u & 1))) & 0xffffu);
                const int k = kj >> 8, j = kj & 127, ext = (kj >> 7) & 1;      // (ext: a column of the TOP launch's slots)
                const long off = (long)(sb + 16 * k) * CS + (ext ? 0 : sb) + 16 * j;
                const double* src = (16 * k + dma_row < n_o) ? dma_lane + off : g_jm_zero;
                double* dst = ring + (q & (JM_RING - 1)) * 256;
                jm_glds16(src, dst);
                jm_glds16(src + 2, dst + 128);
            }
        }
    };
    issue_chunk(0);
    JMPHP(0);

    // ---- prologue: diagonal tiles (raw, row-major) into `linv`, real-row tiles, L_rr w_r, slot and column descriptors ---------
    // (every loop below has a compile-time trip count: its global loads are all in flight together - with run-time bounds hipcc
    // waits for each load before it issues the next, and the prologue was ~25 serial round trips to L2 / HBM)
    // (and every load is UNCONDITIONAL - a clamped address, the value selected afterwards: a load under a lane condition is a branch
    // around it, and hipcc waits for it before the next one)
    {   // (the real-row tiles requested in front of the stream table)
#pragma unroll
        for (int it = 0; it < NVD; ++it) {
            const int e = tid + it * JM_THREADS;
            const int tj = e >> 8, i = (e >> 4) & 15, c = e & 15;
            const int r = 16 * tj + i;
            const double val = (r < n_o) ? ((c <= i) ? vd[it] : 0.0) : ((c == i) ? 1.0 : 0.0);
            if (tj < kmin) linv[e] = val;
        }
        const int nrt = kmin * (kmin - 1) / 2;
#pragma unroll
        for (int it = 0; it < NVR; ++it) {
            const int e = tid + it * JM_THREADS;
            const int ti = e >> 8, i = (e >> 4) & 15, c = e & 15;
            const int k = (ti < 1) ? 1 : ((ti < 3) ? 2 : 3);
            if (e < nrt * 256) realt[ti * 256 + jm_off(i, c)] = (16 * k + i < n_o) ? vr[it] : 0.0;
        }
    }
    JMPHP(1);
    if (!fmode && !bottom) {                      // y' = L_rr w_r, eight lanes per row
        const int row = tid >> 3, part = tid & 7;
        double acc = 0.0;
        double lv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = part + 8 * u;
            const bool ok = row < n_r && c <= row;
            lv[u] = ok ? Lrr[(long)row * n_r + c] : 0.0;
            wv[u] = ok ? w_r[c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fma(lv[u], wv[u], acc);
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (row < n_r && part == 0) yr[row] = acc;
    }
    JMPHP(2);
    {   // runs of slots that share their input point (two slots per thread: tid, tid + 512), and the same for the columns
        static_assert(JM_NT * 16 <= 2 * JM_THREADS && JM_COLS <= JM_THREADS, "two slots, one column per thread");
        auto slot_point = [&](int sl, int& task) -> int {        // point id (real points first) and task of conditioning slot sl
            if (sb + sl < n_r) {
                task = sl % Tr;
                return sl / Tr;
            }
            const int hs = a.h_slots[sb + sl - n_r];
            task = hs % T;
            return gp.N_r + hs / T;
        };
        auto col_point = [&](int c, int& task) -> int {          // the same for column c
            if (fmode) {
                const int hs = a.h_slots[a.n_c + c];
                task = hs % T;
                return hs / T;
            }
            task = (c < mT) ? c % T : -1;
            return (c < mT) ? c / T : m;
        };
        bool flag[3];
        int excl_w[3], pt[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int idx = (r < 2) ? tid + r * JM_THREADS : tid;
            const int lim = (r < 2) ? n_o : ncols;
            int task = 0, tprev = 0;
            flag[r] = false;
            pt[r] = 0;
            if (idx < lim) {
                pt[r] = (r < 2) ? slot_point(idx, task) : col_point(idx, task);
                flag[r] = idx == 0 || pt[r] != ((r < 2) ? slot_point(idx - 1, tprev) : col_point(idx - 1, tprev));
                if (r < 2) {
                    stask[idx] = (signed char)task;
                    if (!fmode && sb + idx >= n_r) ylab[idx] = Yh[a.h_slots[sb + idx - n_r]];
                } else {
                    ctask[idx] = (signed char)task;
                }
            }
            const unsigned long long bal = __ballot(flag[r]);
            excl_w[r] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wtot[r * JM_NW + wv] = __popcll(bal);
        }
        __syncthreads();
        int tot = 0, ctot = 0, off[3] = {0, 0, 0};
#pragma unroll
        for (int w = 0; w < JM_NW; ++w) {
            const int c0 = wtot[w], c1 = wtot[JM_NW + w], c2 = wtot[2 * JM_NW + w];
            if (w < wv) off[0] += c0, off[1] += c1, off[2] += c2;
            off[1] += c0;
            tot += c0 + c1;
            ctot += c2;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int sl = tid + r * JM_THREADS;
            const int excl = off[r] + excl_w[r];
            if (flag[r]) {
                run_start[excl] = (short)sl;
                const double* xp = (sb + sl < n_r) ? a.X_r + (long)pt[r] * D : Xh + (long)(pt[r] - gp.N_r) * D;
                rptx[2 * excl] = xp[0], rptx[2 * excl + 1] = xp[1];
            }
            if (sl < n_o && (sl & (JM_KCH - 1)) == 0) {
                starts_before[sl / JM_KCH] = (short)excl;
                first_run[sl / JM_KCH] = (short)(flag[r] ? excl : excl - 1);
            }
        }
        if (tid < ncols) crun_of[tid] = (unsigned char)(off[2] + excl_w[2] - (flag[2] ? 0 : 1));
        if (flag[2]) {
            const int excl = off[2] + excl_w[2];
            crun_start[excl] = (short)tid;
            double x0 = 0.0, x1 = 0.0;
            if (fmode || tid < mT) {
                const double* xp = fmode ? Xh + (long)pt[2] * D : Xs + (long)pt[2] * D;
                x0 = xp[0], x1 = xp[1];
            }
            cptx[2 * excl] = x0, cptx[2 * excl + 1] = x1;
        }
        if (tid == 0) {
            run_start[tot] = (short)n_o;
            starts_before[(n_o + JM_KCH - 1) / JM_KCH] = (short)tot;
            crun_start[ctot] = (short)ncols;
            wtot[3 * JM_NW] = ctot;
        }
    }
    __syncthreads();
    JMPHP(3);
    const int ncr = wtot[3 * JM_NW];              // column runs
    // The diagonal tiles whose rows are all hallucinated slots (tj >= kmin) go HBM -> LDS directly, row-major as the inversion reads
    // them: lane l of a request moves the 16-byte piece (row 8 h + l / 8, columns 2 (l % 8), + 1) - the LDS side is lane-linear, a
    // tile is two requests.  They are requested HERE - behind the last load of the prologue that anything waits for - and needed by the
    // inversion, which stands behind K_cc (13 k cycles of exponentials and stores, no load anything waits for) (as register loads
    // in front of the point-run scan they were a 16-19 k cycle stall: the first touch of 52 KB per chain by 256 workgroups that
    // start together).  Pieces
    // above the diagonal and pad rows come from an identity tile (the inversion reads the lower triangle and the diagonal only).
    {
        const int ln = jm_lane_now();
        const int di = ln >> 3, dc = 2 * (ln & 7);
#pragma unroll
        for (int nq = 0; nq < (2 * JM_NT + JM_NW - 1) / JM_NW; ++nq) {
            const int q = wv + JM_NW * nq, tj = kmin + (q >> 1), hh = q & 1;
            if (tj < nt) {                        // (uniform)
                const int i = 8 * hh + di, r = 16 * tj + i;
                const double* src = (r < n_o && dc <= i) ? fc + (long)(sb + r - n_r) * CS + sb + 16 * tj + dc : g_jm_eye + i * 16 + dc;
                jm_glds16(src, linv + tj * 256 + hh * 128);
            }
        }
    }

    JMPHP(4);
    // pad columns of the kernel-entry buffer: zero once (no entry is ever written there)
    for (int e = tid; e < JM_KCH * (JM_COLS - ncols); e += JM_THREADS) {
        const int w = JM_COLS - ncols;
        const int row = e / w, col = ncols + (e - row * w);
        kbuf[row * JM_COLS + col] = 0.0;
    }
    __syncthreads();
    JMPHP(5);
    JMPHP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the diagonal tiles have landed
    __syncthreads();
    // invert the diagonal tiles in place: lane (tq, c) of wave w forms column c of the inverse of tile 32 pass + 4 w + tq
    for (int pass = 0; pass * 4 * JM_NW < nt; ++pass) {
        const int tq = lane >> 4, c = lane & 15;
        const int tj = pass * 4 * JM_NW + 4 * wv + tq;
        const bool on = tj < nt;
        const double* Lt = linv + (on ? tj : 0) * 256;
        const double rc = 1.0 / Lt[c * 16 + c];
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc = 0.0;
#pragma unroll
            for (int p = 0; p < i; ++p) sacc = fma(Lt[i * 16 + p], x[p], sacc);
            const double ri = __shfl(rc, (lane & 48) | i, 64);
            x[i] = (i < c) ? 0.0 : ((i == c) ? ri : -sacc * ri);
            asm volatile("" : "+v"(x[i]) : : "memory");      // row i's reads stay in row i (hoisted, the 120 LDS reads of a tile spill)
        }
        // every lane of the wave has read its tiles (they are this wave's alone): overwrite them in the tile format
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (on) {
            double* Lw = linv + tj * 256;
#pragma unroll
            for (int i = 0; i < 16; ++i) Lw[jm_off(i, c)] = x[i];
        }
    }
    JMPHP(7);
    JMPH(0);

    // ---- kernel entries K_o* (and the label column) into the accumulators, JM_KCH slots at a time: one lane per (slot run,
    // column run) pair = one exponential for up to T x T entries -------------------------------------------------------------
    const int nck = fmode ? ncr : ncr - 1;       // column runs of kernel entries (test mode: the last run is the label column)
    const unsigned nck_magic = 0xFFFFFFFFu / (unsigned)nck + 1u;      // e / nck = umulhi(e, magic) for e < 65536
    // kernel entries of the slots 32 ch .. 32 ch + 31 against every column into kbuf: one lane per (slot run, column run) pair
    auto fill_chunk = [&](int ch) {
        const int s0 = ch * JM_KCH;
        const int r0 = first_run[ch];
        const int r1 = starts_before[min(ch + 1, (n_o + JM_KCH - 1) / JM_KCH)];
        const int npairs = (r1 - r0) * nck;
        for (int e = tid; e < npairs; e += JM_THREADS) {
            const int rr = (int)__umulhi((unsigned)e, nck_magic), cr = e - rr * nck;
            const int sa = run_start[r0 + rr], sb = run_start[r0 + rr + 1];
            const int ca = crun_start[cr], cb = crun_start[cr + 1];
            const double xc[D] = {rptx[2 * (r0 + rr)], rptx[2 * (r0 + rr) + 1]};
            const double xt[D] = {cptx[2 * cr], cptx[2 * cr + 1]};
            double qq[D];
            const double k = os * jm_exp_neg(-0.5 * kern_sqdist<D>(xc, xt, il2, qq)); // r = x_slot - x_column
            const int ns = sb - sa;
            if (T == 3 && cb - ca == 3 && (ns == 3 || (ns == 1 && stask[sa] == 0))) {
                // a whole point (value + two derivatives) or a value-only slot against a whole point: the rows of the 3 x 3
                // block, no task look-ups, no loops - also for a point the chunk boundary cuts (its rows outside the chunk are
                // not stored).  (The general form below costs ten dependent LDS round trips per pair, and ONE lane that takes
                // it makes its whole wave walk it: the real slots and every boundary point did - 3-5 k cycles per chunk.)
                const double kq0 = k * qq[0], kq1 = k * qq[1];
                double* row = kbuf + (sa - s0) * JM_COLS + ca;
                if (sa >= s0) row[0] = k, row[1] = kq0, row[2] = kq1;
                if (ns == 3) {
                    if (sa + 1 >= s0 && sa + 1 < s0 + JM_KCH)
                        row[JM_COLS] = -kq0, row[JM_COLS + 1] = k * (il2[0] - qq[0] * qq[0]), row[JM_COLS + 2] = -kq0 * qq[1];
                    if (sa + 2 < s0 + JM_KCH)
                        row[2 * JM_COLS] = -kq1, row[2 * JM_COLS + 1] = -kq1 * qq[0], row[2 * JM_COLS + 2] = k * (il2[1] - qq[1] * qq[1]);
                }
            } else {
                const int sl0 = max(sa, s0), sl1 = min(sb, s0 + JM_KCH);
                for (int sl = sl0; sl < sl1; ++sl) {
                    const int ta = stask[sl];
                    for (int c = ca; c < cb; ++c) kbuf[(sl - s0) * JM_COLS + c] = jm_kern_entry(qq[0], qq[1], k, il2[0], il2[1], ta, ctask[c]);
                }
            }
        }
        // the label column (test mode): one lane of the LAST wave per slot (as a column run of the pair loop it cost every wave
        // that held one of its lanes a divergent walk over the run's slots)
        if (!fmode && tid >= JM_THREADS - JM_KCH) {
            const int sl = s0 + tid - (JM_THREADS - JM_KCH);
            if (sl < n_o) kbuf[(sl - s0) * JM_COLS + mT] = (sb + sl < n_r) ? yr[sl] : ylab[sl];
        }
        if (n_o < s0 + JM_KCH) {                                                          // pad slots: zero rows
            const int z0 = (n_o - s0) * JM_COLS;
            for (int e = z0 + tid; e < JM_KCH * JM_COLS; e += JM_THREADS) kbuf[e] = 0.0;
        }
    };
    JmAcc A;
    jm_acc_begin_a(A);                            // (the tiles' registers are taken from here on, not before: every tile that is read is set below)
    constexpr int JM_CHUNKS_A = 10;
    for (int ch = 0; 2 * ch < nt && ch < JM_CHUNKS_A; ++ch) {     // tiles 0..19: the statements name those only - the other six are not live yet         // (a run-time loop: no C++ branch ever surrounds a statement that writes A)
        fill_chunk(ch);
        JMPHE(0);
        __syncthreads();
        JMPHE(1);
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {             // (tile 2 ch + 1 may lie beyond nt: it is written - with the buffer's stale rows - and never used)
            double t[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) t[v] = kbuf[(16 * u + 4 * (lane >> 4) + v) * JM_COLS + 16 * I0 + (lane & 15)];
            jm_acc_set_a(A, 2 * ch + u, t[0], t[1], t[2], t[3]);
        }
        JMPHE(2);
        __syncthreads();
        JMPHE(3);
    }
    jm_acc_begin_b(A);
    for (int ch = JM_CHUNKS_A; 2 * ch < nt; ++ch) {         // (a run-time loop: no C++ branch ever surrounds a statement that writes A)
        fill_chunk(ch);
        __syncthreads();
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {             // (tile 2 ch + 1 may lie beyond nt: it is written - with the buffer's stale rows - and never used)
            double t[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) t[v] = kbuf[(16 * u + 4 * (lane >> 4) + v) * JM_COLS + 16 * I0 + (lane & 15)];
            jm_acc_set(A, min(2 * ch + u, JM_NT - 1), t[0], t[1], t[2], t[3]);
        }
        __syncthreads();
    }
    JMPH(1);

    // ---- right-looking block substitution ------------------------------------------------------------------------------------
    // Run-time loops over the columns j and, per column, RUNS of tiles k > j whose factor tiles sit in consecutive slots of one chunk
    // of the stream: a run is one statement (jm_acc_fma_run, joint_mfma_gen.inc) that picks its accumulator tiles by computed jumps
    // and fetches every next tile's A operand in the shadow of the current tile's MFMAs (three rotating register sets).  Where the
    // next tile opens a new chunk the statement fetches nothing: barrier (the chunk has landed: its loads were issued a chunk ago;
    // the chunk before it is consumed), the loads of the chunk after it go out, and the tile's A operand is read (jm_sets_load).
    {
        int seq = 0;                              // stream index of the next streamed tile
#ifdef GPMPC_PHASE_TIMERS
        jts = __builtin_readcyclecounter();
#endif
        const unsigned ring_b = (unsigned)(size_t)(__attribute__((address_space(3))) double*)ring;
        const unsigned realt_b = (unsigned)(size_t)(__attribute__((address_space(3))) double*)realt;
        auto streamed_now = [&](int k) { return k < kmin || (seq & (JM_CH - 1)) != 0; };        // tile k can be fetched without a hand-over
        auto tile_addr = [&](int k, int j) -> unsigned {         // LDS byte address of tile (k, j)'s slot (takes the stream's next slot); uniform
            if (k < kmin) return realt_b + (unsigned)(k * (k - 1) / 2 + j) * 2048u;
            const unsigned ad = ring_b + (unsigned)(seq & (JM_RING - 1)) * 2048u;
            ++seq;
            return ad;
        };
        JmSets S;
        jm_sets_begin(S);
        int set = 0;                              // the set that holds (or will take) the next tile's A operand
        unsigned addr_p = realt_b;                // (any valid address: a run without a successor fetches it and drops it)
        bool have = false;                        // set `set` holds the A operand of the next tile
        // the runs of one column: tiles kfirst .. nt - 1 take -= L_kj X_j (xn = -X_j); (next_k, next_j) is the first tile of the column
        // after this one (has_next: there is one) - the last run of the column fetches its A operand
        auto do_column = [&](int jcur, int kfirst, const jm_d4& xn, int next_k, int next_j, bool has_next) {
            int k = kfirst;
            while (k < nt) {
                if (!have) {
                    if (k >= kmin && (seq & (JM_CH - 1)) == 0) {
                        // every wave waits for ITS OWN pieces of the chunk, then the barrier: hipcc's __syncthreads carries no
                        // vmcnt wait for the LDS-DMA loads (it only puts one in front of an LDS read it can see - behind the
                        // barrier, where it covers this wave's pieces and nobody else's)
                        JMPHS(1);
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        JMPHS(3);
                        __syncthreads();
                        JMPHS(4);
                        issue_chunk((seq >> 4) + 1);
                        JMPHS(5);
                    }
                    addr_p = tile_addr(k, jcur);
                    jm_sets_load(S, set, addr_p);
                    JMPHS(1);
                }
                // the run: tile k and the tiles behind it in this column, up to the end of the chunk
                int n = 1;
                if (k >= kmin) n = min(nt - k, JM_CH - ((seq - 1) & (JM_CH - 1)));
                seq += n - 1;
                // the tile behind the run: (k + n, j), or the first tile of the next column, or none
                const bool same = k + n < nt;
                const int kn = same ? k + n : next_k, jn = same ? jcur : next_j;
                have = (same || has_next) && streamed_now(kn);
                const unsigned first = addr_p + 2048u;
                unsigned tail = addr_p;
                if (have) tail = addr_p = tile_addr(kn, jn);
                jm_acc_fma_run(A, S, set, k, n, xn[0], xn[1], xn[2], xn[3], n > 1 ? first : tail, tail);      // acc_k -= L_kj X_j
                set = (set + n) % 3;              // (the last iteration's fetch went there: the next tile, or a dummy that is overwritten)
                k += n;
                JMPHS(2);
            }
        };
        // BOTTOM launch: the nx columns the TOP launch has solved.  X_j of this wave's 16 columns is a 2 KB register dump in a.xbuf;
        // it comes HBM -> LDS into one of two wave-private slots (behind the launch's <= 8 inverted diagonal tiles), requested a
        // column ahead - nothing of it lives in registers across a column (a VGPR the substitution carries is spilled).
        {
            const int nxr = bottom ? nx : 0;      // (a run-time trip count, not a branch around the statements)
            double* xst = linv + 8 * 256 + wv * 512;
            const double* xsrc = a.xbuf + (chain - a.chain0) * JOINT_MFMA_XBUF_DOUBLES + I0 * 256;
            auto x_request = [&](int jx) {
                if (jx < nxr) {                   // (uniform)
                    const int ln = jm_lane_now();
                    const double* src = xsrc + (long)jx * 2048 + ln * 2;
                    double* dst = xst + (jx & 1) * 256;
                    jm_glds16(src, dst);
                    jm_glds16(src + 128, dst + 128);
                }
            };
            x_request(0);
            for (int jx = 0; jx < nxr; ++jx) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's X tile has landed
                jm_d4 x;
                {
                    const int ln = jm_lane_now();
                    const double* xs = xst + (jx & 1) * 256 + ln;
                    x[0] = xs[0], x[1] = xs[64], x[2] = xs[128], x[3] = xs[192];
                }
                x_request(jx + 1);
                const jm_d4 xn = -x;
                JMPHS(0);
                do_column(jx, 0, xn, (jx + 1 < nxr) ? 0 : 1, (jx + 1 < nxr) ? jx + 1 : 0, (jx + 1 < nxr) || nt > 1);
            }
        }
        for (int j = 0; j < nt; ++j) {
            jm_d4 x;
            {
                const int ln = jm_lane_now();      // (re-formed: see issue_chunk)
                const jm_d2 d01 = *reinterpret_cast<const jm_d2*>(linv + j * 256 + ln * 2);
                const jm_d2 d23 = *reinterpret_cast<const jm_d2*>(linv + j * 256 + 128 + ln * 2);
                jm_acc_diag(A, S, j, d01.x, d01.y, d23.x, d23.y, x);  // X_j = Linv_jj acc_j
            }
            const jm_d4 xn = -x;
            // X_j is final: it leaves NOW where somebody else wants it - factor mode: X^T = the new rows' entries against the old
            // columns, into the cache (row n_c + column, 4 consecutive slots per lane); TOP launch: the register dump for the BOTTOM
            // launch.  (Written out in one loop behind the substitution these were 0.84 GB / 1.3 GB per launch in a burst of 256
            // workgroups in lockstep: 35-40 k cycles per chain; here they trickle out under the products.)
            if (fmode && active) {
                const int ln = jm_lane_now();
                const int col = 16 * I0 + (ln & 15), sl = 16 * j + 4 * (ln >> 4);
                if (col < ncols) {
                    double* dst = fc + (long)(a.n_c + col) * CS + sl;
                    if (sl + 3 < n_o) {
                        *reinterpret_cast<jm_d2*>(dst) = jm_d2{x[0], x[1]};
                        *reinterpret_cast<jm_d2*>(dst + 2) = jm_d2{x[2], x[3]};
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (sl + v < n_o) dst[v] = x[v];
                    }
                }
            }
            if (a.pend_write && active) {                       // (uniform; test mode, one launch: the next call's new rows)
                const int ln = jm_lane_now();
                const int col = 16 * I0 + (ln & 15), sl = 16 * j + 4 * (ln >> 4);
                if (col < mT) {
                    double* dst = fc + (long)(a.n_ho + col) * CS + sl;
                    if (sl + 3 < n_o) {
                        *reinterpret_cast<jm_d2*>(dst) = jm_d2{x[0], x[1]};
                        *reinterpret_cast<jm_d2*>(dst + 2) = jm_d2{x[2], x[3]};
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (sl + v < n_o) dst[v] = x[v];
                    }
                }
            }
            if (top) {
                double* xd = a.xbuf + (chain - a.chain0) * JOINT_MFMA_XBUF_DOUBLES + I0 * 256 + (long)j * 2048 + jm_lane_now();
#pragma unroll
                for (int v = 0; v < 4; ++v) xd[v * 64] = x[v];
            }
            JMPHS(0);
            do_column(j, j + 1, xn, j + 2, j + 1, j + 2 < nt);
        }
    }
    __syncthreads();                              // every wave is done with the ring
    JMPH(2);
    // tiles 22..25 of every wave's X go to LDS (the inverted diagonal tiles are no longer needed): the statements of the Gram phase
    // name the tiles 0..21 only, 32 registers come free for the Gram accumulators; the four tiles come back into the registers of
    // tiles 0..3 once those are done with (jm_gram_reload)
    double* stash = linv + wv * (JM_STASH * 256);
#pragma unroll 1
    for (int u = 0; u < JM_STASH; ++u) {
        double t[4];
        jm_acc_get(A, JM_NT - JM_STASH + u, t[0], t[1], t[2], t[3]);
#pragma unroll
        for (int v = 0; v < 4; ++v) stash[(u * 4 + v) * 64 + lane] = t[v];
    }

    // ---- Gram: G_IJ = K_IJ - X_I^T X_J for J = I + d (mod 8), d = 0..4 (d = 4 counts for the tiles I < 4), ONE pass: five accumulators
    // per wave start at -K_IJ, take + X_I^T X_J tile by tile and are negated on the way out.  Groups of four slot tiles: every wave
    // publishes its four X tiles in the exchange buffer (the ring: 8 waves x 4 tiles x 2 KB), barrier, every wave forms its 4 x 5
    // products (A operand: its own tile, in place in its registers; B operands: the published tiles).  (Rounds 1-4 of this kernel
    // ran three passes of two accumulators - every tile published three times, two MFMAs per LDS round trip: 208 k cycles per chain.)
    // test mode: K = K** and the label column / row of G is -mean; factor mode: K = K_nn + noise (the Schur complement)
    const int ldS = fmode ? ncols : mT;           // the matrix's size ...
    // ... and where it goes: factor mode - the chain's S buffer, leading dimension ncols (joint_chol_mfma_kernel's input); the test modes - the
    // covariance view (JointArgs::Sv*: the S buffer, or - pending rows - the diagonal block of the cache rows the next call's new slots take)
    const int ldO = fmode ? ncols : a.Sv_ld;
    double* Sm = fmode ? a.Sall + chain * (long)mT * mT : a.Sv + (chain - a.Sv_chain_base) * a.Sv_cs;
    double* mean = a.mean + chain * (long)mT;
    // -K_cc tile (I, J) straight into the accumulators' registers: register v, lane (kk, jj) = -K_cc(row 16 I + 4 v + kk, column 16 J + jj)
    // (test mode: K**; factor mode: K_nn + noise).  One exponential per ENTRY here - 20 per lane against 1.6 per thread when the
    // prologue formed K_cc by 3 x 3 blocks into the chain's S buffer - but those 115 KB per chain went out to HBM and came back
    // (0.7 GB per launch, written in the burst in which 256 workgroups also fetch their first tiles).
    auto tile_init = [&](int I, int J) -> jm_d4 {
        jm_d4 r;
        const int t2 = 16 * J + (lane & 15);
        const int t2c = min(t2, ncols - 1);
        const int rb = crun_of[t2c], tb = ctask[t2c];
        const double xb[D] = {cptx[2 * rb], cptx[2 * rb + 1]};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int t1 = 16 * I + 4 * v + (lane >> 4);
            const int t1c = min(t1, ncols - 1);
            const int ra = crun_of[t1c], ta = ctask[t1c];
            const double xa[D] = {cptx[2 * ra], cptx[2 * ra + 1]};
            double qq[D];
            const double kv = os * jm_exp_neg(-0.5 * kern_sqdist<D>(xa, xb, il2, qq));
            double val = jm_kern_entry(qq[0], qq[1], kv, il2[0], il2[1], ta, tb);
            if (fmode && t1 == t2) val += (ta == 0) ? noise0 : ((ta == 1) ? noise1 : noise2);
            r[v] = (t1 < ldS && t2 < ldS) ? -val : 0.0;
        }
        return r;
    };
    // BOTTOM launch: the accumulators continue from the TOP launch's results: -S_top, and +mean_top on the label row / column
    auto tile_cont = [&](int I, int J) -> jm_d4 {
        jm_d4 r;
        const int t2 = 16 * J + (lane & 15);
        double ls[4], lm[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {                         // (unconditional loads, clamped)
            const int t1 = 16 * I + 4 * v + (lane >> 4);
            ls[v] = Sm[(long)min(t1, ldS - 1) * ldO + min(t2, ldS - 1)];
            lm[v] = mean[min((t1 == mT) ? t2 : t1, mT - 1)];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int t1 = 16 * I + 4 * v + (lane >> 4);
            const bool lab = (t1 == mT && t2 < mT) || (t2 == mT && t1 < mT);
            r[v] = (t1 < ldS && t2 < ldS) ? -ls[v] : (lab ? lm[v] : 0.0);
        }
        return r;
    };
    int Jt[5];
    jm_d4 kinit[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        Jt[d] = (I0 + d) & (JM_NCT - 1);
        kinit[d] = bottom ? tile_cont(I0, Jt[d]) : tile_init(I0, Jt[d]);
    }
    const unsigned ring_gb = (unsigned)(size_t)(__attribute__((address_space(3))) double*)ring;
    const unsigned stash_gb = (unsigned)(size_t)(__attribute__((address_space(3))) double*)linv + (unsigned)wv * (JM_STASH * 2048u);
    double* tbuf = ring + wv * 272;               // [16][17] per wave, inside the exchange buffer: used between two barriers of its own
    // pend_write: S also goes into the diagonal block of the cache rows the next call's new slots will occupy (both triangles) - unless
    // the covariance view already IS that block (the dispatcher points Sv there when joint_tail_mfma_kernel follows)
    double* pblk = fc + (long)a.n_ho * CS + n_r + a.n_ho;
    const bool pend2 = a.pend_write && pblk != Sm;
    auto tile_out = [&](int I, int J, int d, const jm_d4& sn) {
        // the tile and its mirror image, both as 128-byte row segments: the mirror through a per-wave LDS transpose
        const int jj = lane & 15, kk = lane >> 4;
        const jm_d4 sv = -sn;
#pragma unroll
        for (int v = 0; v < 4; ++v) tbuf[(4 * v + kk) * 17 + jj] = sv[v];
        const int t2 = 16 * J + jj;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int t1 = 16 * I + 4 * v + kk;
            const double val = sv[v];
            if (t1 < ldS && t2 < ldS) {
                if (d > 0 || t1 >= t2) Sm[(long)t1 * ldO + t2] = val;      // diagonal tiles: the lower part here, its mirror below
                if (pend2 && (d > 0 || t1 >= t2)) pblk[(long)t1 * CS + t2] = val;
            } else if (!fmode && t1 == mT && t2 < mT) {
                mean[t2] = -val;                                            // the label row
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r2 = 16 * J + 4 * v + kk, c2 = 16 * I + jj;           // entry (r2, c2) of S = entry (c2, r2) of the tile
            const double val = tbuf[jj * 17 + 4 * v + kk];
            if (r2 < ldS && c2 < ldS) {
                if (d > 0 || c2 > r2) Sm[(long)r2 * ldO + c2] = val;
                if (pend2 && (d > 0 || c2 > r2)) pblk[(long)r2 * CS + c2] = val;
            } else if (!fmode && r2 == mT && c2 < mT && d > 0) {
                mean[c2] = -val;                                            // the label column of an off-diagonal tile
            }
        }
    };
    {
        bool on[5];
        unsigned ab[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            on[d] = I0 < ncta && Jt[d] < ncta && (d < 4 || I0 < 4);         // (what is written out; every product is formed)
            ab[d] = ring_gb + (unsigned)Jt[d] * 8192u;
        }
        JmGram Gm;
        Gm.s0 = kinit[0], Gm.s1 = kinit[1], Gm.s2 = kinit[2], Gm.s3 = kinit[3], Gm.s4 = kinit[4];
        jm_gram_begin(Gm);
        JMPH(3);
        const unsigned pub = ring_gb + (unsigned)I0 * 8192u;
        for (int g = 0; 4 * g < nt; ++g) {
            __syncthreads();                      // the previous group's tiles have been read
            jm_gram_reload(A, Gm, stash_gb, __builtin_amdgcn_readfirstlane(g == 2 ? 1 : 0));
            const int nu = min(4, nt - 4 * g);
#pragma unroll 1
            for (int u = 0; u < nu; ++u) jm_gram_publish(A, Gm, 4 * g + u, pub);
            __syncthreads();
            JMPH(4);
#pragma unroll 1
            for (int u = 0; u < nu; ++u) jm_gram_tile(A, Gm, 4 * g + u, ab[0], ab[1], ab[2], ab[3], ab[4]);
            JMPH(5);
        }
        __syncthreads();                          // every wave has read the last group's tiles: the ring takes the transposes
        jm_settle(Gm.s4);                         // (the last MFMA issued wrote s4; the others are older)
        if (on[0]) tile_out(I0, Jt[0], 0, Gm.s0);
        if (on[1]) tile_out(I0, Jt[1], 1, Gm.s1);
        if (on[2]) tile_out(I0, Jt[2], 2, Gm.s2);
        if (on[3]) tile_out(I0, Jt[3], 3, Gm.s3);
        if (on[4]) tile_out(I0, Jt[4], 4, Gm.s4);
        JMPH(6);
    }
#ifdef GPMPC_PHASE_TIMERS
    if (blockIdx.x == 0 && (tid == 0 || tid == JM_THREADS - 64))
    {
        jph[7] = jphs[0];                         // (slot 7 of the phase record: diagonal steps; the other two are recovered below)
        for (int i = 0; i < 8; ++i) g_jm_phase[a.mfma_mode][i + (tid ? 8 : 0)] = jph[i];
        g_jm_phase[a.mfma_mode][16 + (tid ? 2 : 0)] = jphs[1], g_jm_phase[a.mfma_mode][17 + (tid ? 2 : 0)] = jphs[2];
        if (tid == 0) {
            for (int i = 0; i < 8; ++i) g_jm_phase[a.mfma_mode][20 + i] = jpp[i];
            for (int i = 0; i < 4; ++i) g_jm_phase[a.mfma_mode][28 + i] = jpe[i];
        }
        for (int i = 0; i < 3; ++i) g_jm_phase[a.mfma_mode][32 + i + (tid ? 3 : 0)] = jphs[3 + i];
    }
#endif
}

bool joint_mfma_eligible(int n_r, int n_hc, int ncols, int T) {
    const int n_o = n_r + n_hc;
    // (T = 3 only: a value-only model - T = 1 - never has hallucinated data behind the reference's call surface, src/agent.py:221-226;
    // the kernel is written for both, but no test reaches the T = 1 form: it is not instantiated)
    return T == 3 && n_r >= 1 && n_r <= 64 && n_hc >= 0 && (n_o + 15) / 16 <= JM_NT && ncols >= 1 && ncols <= JM_COLS;
}

bool joint_mfma_split_eligible(int n_r, int n_hc, int ncols, int T) {
    const int n_o = n_r + n_hc;
    return T == 3 && n_r >= 1 && n_r <= 64 && n_o > JOINT_MFMA_SPLIT && n_o <= JOINT_MFMA_SPLIT + JOINT_MFMA_BOTTOM_MAX &&
           ncols >= 1 && ncols <= JM_COLS;
}

int joint_mfma_launch(const JointArgs& a, hipStream_t st) {
    if (a.gp.T != 3) return fail(GPMPC_E_UNSUPPORTED, "joint_test_mfma_kernel is instantiated for T = 3");
    // per launch, like every other launcher here: the attribute is per DEVICE (a process may drive several) and a static
    // flag would not be thread safe
    GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)joint_test_mfma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)JM_SMEM_BYTES));
    const dim3 g((unsigned)(a.chain1 - a.chain0)), b(JM_THREADS);
    hipLaunchKernelGGL((joint_test_mfma_kernel<3>), g, b, JM_SMEM_BYTES, st, a);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_joint_mfma_phases(long long* out /*[host] 40*/) {      // the test-mode launch
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_jm_phase), 40 * sizeof(long long), 0));
    return GPMPC_OK;
}
// the same record of the last launch of `mode` (JOINT_MFMA_TEST / _FACTOR / _TEST_TOP / _TEST_BOTTOM)
extern "C" int gpmpc_debug_read_joint_mfma_phases_of(int mode, long long* out /*[host] 40*/) {
    if (mode < 0 || mode > 3) return GPMPC_E_ARG;
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_jm_phase), 40 * sizeof(long long), (size_t)mode * 40 * sizeof(long long)));
    return GPMPC_OK;
}
