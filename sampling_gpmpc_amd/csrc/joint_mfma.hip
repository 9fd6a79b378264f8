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
// 32-slot LDS buffer into the accumulators.  Afterwards the Gram products X_I^T X_J run on the same registers (A operand = the
// wave's own X, negated; B = wave J's X through LDS), accumulators starting at K** so that the running value shrinks towards S
// as in the sequential form.
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

constexpr int JM_SMEM_DOUBLES = JM_NT * 256 + JM_RING * 256 + JM_RT * 256 + 64 + JM_NT * 16 * 2 + JM_COLS * 2;
constexpr int JM_SMEM_SHORTS = (JM_NT * 16 + 8) + (JM_NTD + 7) + 32 + 32 + (JM_COLS + 8) + (JM_NT * 16 + JM_COLS) / 2 + 2 * (4 * JM_NW + 4);
constexpr size_t JM_SMEM_BYTES = (size_t)JM_SMEM_DOUBLES * 8 + (size_t)JM_SMEM_SHORTS * 2 + 64;

__device__ __attribute__((aligned(16))) double g_jm_zero[2] = {0.0, 0.0};
__device__ long long g_jm_phase[8];

#ifdef GPMPC_PHASE_TIMERS
#define JMPH(idx) do { const long long _n = __builtin_readcyclecounter(); jph[idx] += _n - jt; jt = _n; } while (0)
#else
#define JMPH(idx)
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

__device__ __forceinline__ void jm_glds16(const double* src, double* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// nd accumulators += a * b[.] (Gram products: independent chains back to back; the accumulators are read again as SrcC of the same
// opcode only - no trailing wait states; jm_settle() stands in front of their first other reader)
__device__ __forceinline__ void jm_gram_fma3(jm_d4& s0, jm_d4& s1, jm_d4& s2, double an, double b0, double b1, double b2) {
    asm("s_nop 1\n\t"
        "v_mfma_f64_16x16x4_f64 %0, %3, %4, %0\n\t"
        "v_mfma_f64_16x16x4_f64 %1, %3, %5, %1\n\t"
        "v_mfma_f64_16x16x4_f64 %2, %3, %6, %2"
        : "+v"(s0), "+v"(s1), "+v"(s2)
        : "v"(an), "v"(b0), "v"(b1), "v"(b2));
}
__device__ __forceinline__ void jm_gram_fma2(jm_d4& s0, jm_d4& s1, double an, double b0, double b1) {
    asm("s_nop 1\n\t"
        "v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\t"
        "v_mfma_f64_16x16x4_f64 %1, %2, %4, %1"
        : "+v"(s0), "+v"(s1)
        : "v"(an), "v"(b0), "v"(b1));
}
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
    double* linv = jm_smem;                       // [JM_NT][256]  inverted diagonal tiles, tile format
    double* ring = linv + JM_NT * 256;            // [JM_RING][256] streamed tiles; second half = kernel-entry buffer; all = Gram exchange
    double* realt = ring + JM_RING * 256;         // [JM_RT][256]  off-diagonal tiles with real rows
    double* yr = realt + JM_RT * 256;             // [64] L_rr w_r: the right-hand side that reproduces w_r on the real slots
    double* rptx = yr + 64;                       // [JM_NT * 16][2] input point of every slot run
    double* cptx = rptx + JM_NT * 16 * 2;         // [JM_COLS][2] input point of every column run
    short* run_start = reinterpret_cast<short*>(cptx + JM_COLS * 2);     // [JM_NT * 16 + 8] first slot of every conditioning point's run
    unsigned short* stab = reinterpret_cast<unsigned short*>(run_start + JM_NT * 16 + 8);  // [JM_NTD + 7] streamed tile -> (k << 8) | j
    short* first_run = reinterpret_cast<short*>(stab + JM_NTD + 7);  // [32] run that contains slot 32 c
    short* starts_before = first_run + 32;                           // [32] runs that start before slot 32 c
    short* crun_start = starts_before + 32;                          // [JM_COLS + 8] first column of every column run
    signed char* stask = reinterpret_cast<signed char*>(crun_start + JM_COLS + 8);   // [JM_NT * 16] task of every slot
    signed char* ctask = stask + JM_NT * 16;                         // [JM_COLS] task of every column (-1: the label column)
    int* wtot = reinterpret_cast<int*>(ctask + JM_COLS);             // [4][JM_NW]
    double* kbuf = ring + JM_CH * 256;            // [JM_KCH][JM_COLS]

    const GpParams& gp = a.gp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool fmode = a.mfma_mode == JOINT_MFMA_FACTOR;
    const int n_r = gp.n_r, Tr = gp.real_has_grad ? T : 1;
    const int m = a.m, mT = m * T;
    const int n_hc = fmode ? a.n_c : a.n_ho;      // hallucinated slots conditioned on
    const int n_o = n_r + n_hc;
    const int ncols = fmode ? a.n_ho - a.n_c : mT + 1;
    const int nt = (n_o + 15) >> 4;               // slot tiles
    const int kmin = (n_r + 15) >> 4;             // tile rows < kmin hold real rows: staged through registers
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
#ifdef GPMPC_PHASE_TIMERS
    long long jph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long jt = __builtin_readcyclecounter();
#endif

    auto Lel = [&](int r, int c) -> double {      // factor entry (r, c), c <= r < n_o
        return (r < n_r) ? Lrr[(long)r * n_r + c] : fc[(long)(r - n_r) * CS + c];
    };

    // ---- table of the STREAMED tiles (k, j), k >= max(j + 1, kmin): rows that are all hallucinated slots, in consumption order
    // (column by column); the few tiles with real rows (k < kmin) sit in `realt` ------------------------------------------------
    int ntd = 0;
    {
        int sbase = 0;
        for (int jj = 0; jj < nt; ++jj) {
            const int k0 = max(jj + 1, kmin);
            if (jj == tid)
                for (int k = k0; k < nt; ++k) stab[sbase + k - k0] = (unsigned short)((k << 8) | jj);
            sbase += max(0, nt - k0);
        }
        ntd = sbase;
    }
    __syncthreads();

    // chunk c of the stream -> ring slots (c & 1) * 16 ..: wave w moves tiles 2 w, 2 w + 1 of the chunk
    auto issue_chunk = [&](int c) {
#pragma unroll
        for (int u = 0; u < JM_CH / JM_NW; ++u) {
            const int q = c * JM_CH + wv * (JM_CH / JM_NW) + u;
            if (q < ntd) {
                const int kj = __builtin_amdgcn_readfirstlane((int)stab[q]);
                const int k = kj >> 8, j = kj & 255;
                const int r = 16 * k + jm_pi(lane & 15);
                const bool ok = r < n_o;
                const double* src = ok ? fc + (long)(r - n_r) * CS + 16 * j + 4 * (lane >> 4) : g_jm_zero;
                double* dst = ring + (q & (JM_RING - 1)) * 256;
                jm_glds16(src, dst);
                jm_glds16(ok ? src + 2 : g_jm_zero, dst + 128);
            }
        }
    };
    issue_chunk(0);

    // ---- prologue: diagonal tiles (raw, row-major) into `linv`, real-row tiles, L_rr w_r, slot and column descriptors ---------
    for (int e = tid; e < nt * 256; e += JM_THREADS) {
        const int tj = e >> 8, i = (e >> 4) & 15, c = e & 15;
        const int r = 16 * tj + i;
        double v;
        if (r < n_o) v = (c <= i) ? Lel(r, 16 * tj + c) : 0.0;
        else v = (c == i) ? 1.0 : 0.0;
        linv[e] = v;
    }
    {
        const int nrt = kmin * (kmin - 1) / 2;
        for (int e = tid; e < nrt * 256; e += JM_THREADS) {
            const int ti = e >> 8, i = (e >> 4) & 15, c = e & 15;
            const int k = (ti < 1) ? 1 : ((ti < 3) ? 2 : 3);
            const int j = ti - k * (k - 1) / 2;
            const int r = 16 * k + i;
            realt[ti * 256 + jm_off(i, c)] = (r < n_o) ? Lel(r, 16 * j + c) : 0.0;
        }
    }
    if (!fmode) {                                 // y' = L_rr w_r, eight lanes per row
        const int row = tid >> 3, part = tid & 7;
        double acc = 0.0;
        if (row < n_r)
            for (int c = part; c <= row; c += 8) acc = fma(Lrr[(long)row * n_r + c], w_r[c], acc);
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (row < n_r && part == 0) yr[row] = acc;
    }
    {   // runs of slots that share their input point (two slots per thread: tid, tid + 512), and the same for the columns
        static_assert(JM_NT * 16 <= 2 * JM_THREADS && JM_COLS <= JM_THREADS, "two slots, one column per thread");
        auto slot_point = [&](int sl, int& task) -> int {        // point id (real points first) and task of conditioning slot sl
            if (sl < n_r) {
                task = sl % Tr;
                return sl / Tr;
            }
            const int hs = a.h_slots[sl - n_r];
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
                if (r < 2) stask[idx] = (signed char)task;
                else ctask[idx] = (signed char)task;
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
                const double* xp = (sl < n_r) ? a.X_r + (long)pt[r] * D : Xh + (long)(pt[r] - gp.N_r) * D;
                rptx[2 * excl] = xp[0], rptx[2 * excl + 1] = xp[1];
            }
            if (sl < n_o && (sl & (JM_KCH - 1)) == 0) {
                starts_before[sl / JM_KCH] = (short)excl;
                first_run[sl / JM_KCH] = (short)(flag[r] ? excl : excl - 1);
            }
        }
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
    const int ncr = wtot[3 * JM_NW];              // column runs
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
    // pad columns of the kernel-entry buffer: zero once (no entry is ever written there)
    for (int e = tid; e < JM_KCH * (JM_COLS - ncols); e += JM_THREADS) {
        const int w = JM_COLS - ncols;
        const int row = e / w, col = ncols + (e - row * w);
        kbuf[row * JM_COLS + col] = 0.0;
    }
    __syncthreads();
    JMPH(0);

    // ---- kernel entries K_o* (and the label column) into the accumulators, JM_KCH slots at a time: one lane per (slot run,
    // column run) pair = one exponential for up to T x T entries -------------------------------------------------------------
    JmAcc A;
    A.g0 = A.g1 = A.g2 = A.g3 = A.g4 = A.g5 = jm_d16(0.0);
    A.g6 = jm_d8(0.0);
    const unsigned ncr_magic = 0xFFFFFFFFu / (unsigned)ncr + 1u;      // e / ncr = umulhi(e, magic) for e < 65536
    for (int ch = 0; 2 * ch < nt; ++ch) {         // (a run-time loop: no C++ branch ever surrounds a statement that writes A)
        const int s0 = ch * JM_KCH;
        const int r0 = first_run[ch];
        const int r1 = starts_before[min(ch + 1, (n_o + JM_KCH - 1) / JM_KCH)];
        const int npairs = (r1 - r0) * ncr;
        for (int e = tid; e < npairs; e += JM_THREADS) {
            const int rr = (int)__umulhi((unsigned)e, ncr_magic), cr = e - rr * ncr;
            const int sa = run_start[r0 + rr], sb = run_start[r0 + rr + 1];
            const int ca = crun_start[cr], cb = crun_start[cr + 1];
            const int sl0 = max(sa, s0), sl1 = min(sb, s0 + JM_KCH);
            if (ctask[ca] >= 0) {
                const double xc[D] = {rptx[2 * (r0 + rr)], rptx[2 * (r0 + rr) + 1]};
                const double xt[D] = {cptx[2 * cr], cptx[2 * cr + 1]};
                double qq[D];
                const double k = kern_scalar<D>(xc, xt, il2, os, qq);                     // r = x_slot - x_column
                for (int sl = sl0; sl < sl1; ++sl) {
                    const int ta = stask[sl];
                    for (int c = ca; c < cb; ++c) kbuf[(sl - s0) * JM_COLS + c] = kern_entry<D>(qq, k, il2, ta, ctask[c]);
                }
            } else {                                                                      // the label column
                for (int sl = sl0; sl < sl1; ++sl)
                    kbuf[(sl - s0) * JM_COLS + ca] = (sl < n_r) ? yr[sl] : Yh[a.h_slots[sl - n_r]];
            }
        }
        if (n_o < s0 + JM_KCH) {                                                          // pad slots: zero rows
            const int z0 = (n_o - s0) * JM_COLS;
            for (int e = z0 + tid; e < JM_KCH * JM_COLS; e += JM_THREADS) kbuf[e] = 0.0;
        }
        __syncthreads();
#pragma unroll
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
    // Run-time loops over the columns j and their tiles k > j; the statements pick their accumulator tile by a computed jump
    // (joint_mfma_gen.inc): the schedule (which tiles exist, where a chunk of the stream ends) depends on n_o, the code does not.
    // The next tile's A operand is requested from LDS before this tile's MFMAs are issued (unless it lies behind a chunk boundary).
    {
        int seq = 0;                              // streamed tiles fetched so far
        auto fetch = [&](int k, int j, jm_d2& a01, jm_d2& a23) {
            const double* base;
            if (k >= kmin) {
                if ((seq & (JM_CH - 1)) == 0) {
                    __syncthreads();              // chunk seq / 16 has landed (its loads were issued a chunk ago); the chunk before is consumed
                    issue_chunk((seq >> 4) + 1);
                }
                base = ring + (seq & (JM_RING - 1)) * 256;
                ++seq;
            } else {
                base = realt + (k * (k - 1) / 2 + j) * 256;
            }
            a01 = *reinterpret_cast<const jm_d2*>(base + lane * 2);
            a23 = *reinterpret_cast<const jm_d2*>(base + 128 + lane * 2);
        };
        for (int j = 0; j < nt; ++j) {
            jm_d4 x;
            {
                const jm_d2 d01 = *reinterpret_cast<const jm_d2*>(linv + j * 256 + lane * 2);
                const jm_d2 d23 = *reinterpret_cast<const jm_d2*>(linv + j * 256 + 128 + lane * 2);
                jm_acc_diag(A, j, d01.x, d01.y, d23.x, d23.y, x);     // X_j = Linv_jj acc_j
            }
            const jm_d4 xn = -x;
            jm_d2 c01 = jm_d2{0.0, 0.0}, c23 = jm_d2{0.0, 0.0};
            if (j + 1 < nt) fetch(j + 1, j, c01, c23);
            for (int k = j + 1; k < nt; ++k) {
                jm_d2 n01 = c01, n23 = c23;
                const bool more = k + 1 < nt;
                const bool ahead = more && !(k + 1 >= kmin && (seq & (JM_CH - 1)) == 0);
                if (ahead) fetch(k + 1, j, n01, n23);
                jm_acc_fma(A, k, c01.x, c01.y, c23.x, c23.y, xn[0], xn[1], xn[2], xn[3]);      // acc_k -= L_kj X_j
                if (more && !ahead) fetch(k + 1, j, n01, n23);
                c01 = n01, c23 = n23;
            }
        }
    }
    __syncthreads();                              // every wave is done with the ring
    JMPH(2);

    // ---- Gram: G_IJ = K_IJ - X_I^T X_J for J = I + d (mod 8), d = 0..4 (d = 4 counts for the tiles I < 4) ---------------------
    // test mode: K = K** and the label column / row of G is -mean; factor mode: K = K_nn + noise (the Schur complement)
    const int ldS = fmode ? ncols : mT;
    double* Sm = a.Sall + chain * (long)mT * mT;
    double* mean = a.mean + chain * (long)mT;
    auto tile_init = [&](int I, int J, bool on) -> jm_d4 {
        jm_d4 r = jm_d4{0.0, 0.0, 0.0, 0.0};
        if (on) {
            // register v, lane (kk, jj) = (row 16 I + 4 v + kk, column 16 J + jj)
            const int t2 = 16 * J + (lane & 15);
            const int c2 = min(t2, ldS - 1);
            int j2, b2;
            if (fmode) {
                const int hs = a.h_slots[a.n_c + c2];
                j2 = hs / T, b2 = hs - j2 * T;
            } else {
                j2 = c2 / T, b2 = c2 - j2 * T;
            }
            const double* x2 = (fmode ? Xh : Xs) + (long)j2 * D;
            for (int v = 0; v < 4; ++v) {
                const int t1 = 16 * I + 4 * v + (lane >> 4);
                const int c1 = min(t1, ldS - 1);
                int j1, b1;
                if (fmode) {
                    const int hs = a.h_slots[a.n_c + c1];
                    j1 = hs / T, b1 = hs - j1 * T;
                } else {
                    j1 = c1 / T, b1 = c1 - j1 * T;
                }
                const double* x1 = (fmode ? Xh : Xs) + (long)j1 * D;
                double qq[D];
                const double kv = kern_scalar<D>(x1, x2, il2, os, qq);
                double val = kern_entry<D>(qq, kv, il2, b1, b2);
                if (fmode && t1 == t2) val += gp.noise[b1];
                r[v] = (t1 < ldS && t2 < ldS) ? val : 0.0;
            }
        }
        return r;
    };
    auto tile_out = [&](int I, int J, int d, const jm_d4& sv) {
        const int t2 = 16 * J + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int t1 = 16 * I + 4 * v + (lane >> 4);
            const double val = sv[v];
            const bool lower_ok = d > 0 || t1 >= t2;        // diagonal tiles: the lower part and its mirror (one writer per entry)
            if (t1 < ldS && t2 < ldS) {
                if (lower_ok) {
                    Sm[(long)t1 * ldS + t2] = val;
                    if (t1 != t2) Sm[(long)t2 * ldS + t1] = val;
                }
            } else if (!fmode && t1 == mT && t2 < mT) {
                if (lower_ok) mean[t2] = -val;
            } else if (!fmode && t2 == mT && t1 < mT) {
                if (d > 0) mean[t1] = -val;
            }
        }
    };
    auto gram_pass = [&](auto d0c, auto ndc) {
        constexpr int d0 = decltype(d0c)::value, nd = decltype(ndc)::value;
        static_assert(nd == 2 || nd == 3, "two or three accumulators per pass");
        jm_d4 sacc[3];
        bool on[3];
        int Jt[3];
#pragma unroll
        for (int dd = 0; dd < nd; ++dd) {
            const int d = d0 + dd;
            Jt[dd] = (I0 + d) & (JM_NCT - 1);
            on[dd] = I0 < ncta && Jt[dd] < ncta && (d < 4 || I0 < 4);
            sacc[dd] = tile_init(I0, Jt[dd], on[dd]);
        }
        for (int j0 = 0; j0 < nt; j0 += 4) {
            __syncthreads();                      // the previous group's tiles have been read
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + u, nt - 1);
                double t[4];
                jm_acc_get(A, j, t[0], t[1], t[2], t[3]);
                if (active) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) ring[((I0 * 4 + u) * 4 + v) * 64 + lane] = t[v];
                    if (d0 == 0 && fmode && j0 + u < nt) {
                        // X^T = the new rows' entries against the old columns: row n_c + column, 4 consecutive slots per lane
                        const int col = 16 * I0 + (lane & 15), sl = 16 * j + 4 * (lane >> 4);
                        if (col < ncols) {
                            double* dst = fc + (long)(a.n_c + col) * CS + sl;
                            if (sl + 3 < n_o) {
                                *reinterpret_cast<jm_d2*>(dst) = jm_d2{t[0], t[1]};
                                *reinterpret_cast<jm_d2*>(dst + 2) = jm_d2{t[2], t[3]};
                            } else {
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    if (sl + v < n_o) dst[v] = t[v];
                            }
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool valid = j0 + u < nt;   // (a tile beyond nt: the last tile again, with a zero A operand)
                double t[4];
                jm_acc_get(A, min(j0 + u, nt - 1), t[0], t[1], t[2], t[3]);      // (again: 4 moves are cheaper than 8 live registers per tile)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double an = valid ? -t[v] : 0.0;
                    double b[3];
#pragma unroll
                    for (int dd = 0; dd < nd; ++dd) {
                        const double bv = (d0 + dd == 0) ? t[v] : ring[((Jt[dd] * 4 + u) * 4 + v) * 64 + lane];
                        b[dd] = on[dd] ? bv : 0.0;
                    }
                    if constexpr (nd == 3) jm_gram_fma3(sacc[0], sacc[1], sacc[2], an, b[0], b[1], b[2]);
                    else jm_gram_fma2(sacc[0], sacc[1], an, b[0], b[1]);
                }
            }
        }
#pragma unroll
        for (int dd = 0; dd < nd; ++dd) {
            jm_settle(sacc[dd]);
            if (on[dd]) tile_out(I0, Jt[dd], d0 + dd, sacc[dd]);
        }
    };
    gram_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
    gram_pass(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
    JMPH(3);
#ifdef GPMPC_PHASE_TIMERS
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) g_jm_phase[i] = jph[i];
#endif
}

bool joint_mfma_eligible(int n_r, int n_hc, int ncols, int T) {
    const int n_o = n_r + n_hc;
    return (T == 1 || T == 3) && n_r >= 1 && n_r <= 64 && n_hc >= 0 && (n_o + 15) / 16 <= JM_NT && ncols >= 1 && ncols <= JM_COLS;
}

int joint_mfma_launch(const JointArgs& a, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)joint_test_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)JM_SMEM_BYTES));
        GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)joint_test_mfma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)JM_SMEM_BYTES));
        attr_done = true;
    }
    const dim3 g((unsigned)(a.chain1 - a.chain0)), b(JM_THREADS);
    if (a.gp.T == 1) hipLaunchKernelGGL((joint_test_mfma_kernel<1>), g, b, JM_SMEM_BYTES, st, a);
    else hipLaunchKernelGGL((joint_test_mfma_kernel<3>), g, b, JM_SMEM_BYTES, st, a);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // namespace gpmpc

extern "C" int gpmpc_debug_read_joint_mfma_phases(long long* out /*[host] 8*/) {
    GPMPC_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(gpmpc::g_jm_phase), 8 * sizeof(long long)));
    return GPMPC_OK;
}
