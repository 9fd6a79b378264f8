// Eigendecomposition root of the joint posterior covariance (mode "J"), in-kernel, gfx950.
//
// Reference behaviour (gpytorch 1.13 / linear_operator at reference src/agent.py:640-641): when the Cholesky of the
// m*T x m*T posterior covariance S fails even after the three jitter retries - which is what happens on EVERY joint
// draw of the shipped params_car_residual.yaml:51 (Dyn_gp_jitter 1e-20; S is numerically singular) -
// root_decomposition() falls back, for the WHOLE batch, to
//        evals, evecs = eigh(S);   R = evecs * sqrt(clamp(evals, 0));   y = mean + R z        (ascending evals)
//
// What runs here: ONE WAVE per (sample, output) chain (64-thread workgroups, no barrier anywhere on the LDS path; the
// CU overlaps ~9 chains), n = m*T <= 256 test slots, lane owns rows lane, lane+64, ...:
//   A. diagonally pivoted Cholesky  S ~= L L^T  (left-looking, L is n x r, stops when the largest residual diagonal
//      entry is <= tol): S has numerical rank r ~ 30..60 of n = 120; everything below tol is the round-off noise of
//      S = K** - V^T V itself (its eigenvalues there are +-1e-16 and their eigenvectors arbitrary, also for LAPACK);
//   B. G = L^T L  (r x r, FP64 MFMA 16x16x4) - G has exactly the non-zero eigenvalues of L L^T;
//   C. cyclic two-sided Jacobi on G, round-robin ordering IN POSITION SPACE: the matrix is kept permuted so that the
//      pairs of every round are the fixed positions (a, rp-1-a); a round rotates the rp/2 disjoint pairs, updates the
//      symmetric matrix 2x2 block by 2x2 block (block (I,K) only needs the rotations of pairs I and K) and writes every
//      entry to the position its indices take in the next round (position 0 fixed, the others shift by one).  All
//      addresses are therefore round independent: each lane precomputes the read / write slots of its <= 7 blocks once,
//      a round is "read my blocks - rotate - write them shifted" on the packed upper triangle in LDS (<= 12.8 KB).
//      After every sweep (rp-1 rounds) the positions are back where they started.  Rotation tangents are logged;
//   D. eigenvalues = diag(G) (>= 0 by construction: no clamp needed), ascending rank -> which base sample belongs to
//      which eigenvector (R's columns are ordered like eigh's: the n-r zero columns first); t = W z~ by replaying the
//      logged rounds in reverse on the permuted base-sample vector (W itself is never formed);
//   E. y = mean + L t  (R = L W: column j is sqrt(lambda_j) u_j), then the post-processing of sample_gp;
//   F. only when the caller asks for the root (tests): column j of R = L (W e_j), one replay per column.
// L W has the columns sqrt(lambda_j) u_j of the eigh root up to their signs (solver specific in LAPACK too) and up to
// the noise directions; tests align the signs per column and compare the samples.
// Ranks beyond EIGH_LDS_RANK take the same algorithm with G double-buffered in the HBM/L2 workspace.
#pragma once
#include <cfloat>
#include <cstdlib>

namespace gpmpc {

struct EighArgs {
    GpParams gp;
    long Ns;
    int m;
    const double* z;
    double var_zero_thr, beta;
    int apply_clip;
    const double* mean;      // written by joint_kernel
    const double* var;
    double* y;
    int* info;
    const double* Sall;      // chain c at Sall + c * S_cs, element (r, c2) at r * S_ld + c2, both triangles valid (default: S_ld = n, S_cs = n * n; with
    int S_ld;                //   pending rows written it is the diagonal block of the factor-cache rows: S_ld = the cache's row stride)
    long S_cs;
    const int* any_fail;     // device flag set by joint_kernel when a chain's jitter chain failed
    int force;
    double* ws;              // per slot: L [n*n] | 2 x packed G [np*(np+1)/2] | rotation log [EIGH_MAX_SWEEPS/2 * np*np]
    long ws_slot_stride;
    double* root;            // optional [chains][n][n] row-major
    int lds_cap;             // largest (even) rank whose packed Gram matrix fits the dynamic LDS
    double tol_mult;         // pivoted-Cholesky stop: residual diagonal <= tol_mult * eps * max prior variance
    // slot layout (doubles from the slot's start): L [n][l_cols] column-major | 2 x packed G (ranks beyond the LDS cap) | rotation log
    long gg_off, rlog_off;
    // two-launch form for batches of low rank (the closed loop's points: ranks 6..16 of 120): EIGH_PASS_NARROW runs first with a
    // small LDS cap - twice the chains resident per CU - and DEFERS a chain whose rank exceeds defer_rank (its id goes to
    // defer_list, nothing of it is written); EIGH_PASS_DEFERRED then handles exactly the listed chains with the full cap.  A chain's
    // result does not depend on which launch produced it: both take the LDS path of the same code for its rank.
    int pass;                // EIGH_PASS_ALL / _NARROW / _DEFERRED
    int defer_rank;
    int* defer_list;         // [chains]
    int* defer_count;
    int* rank_hint;          // max rank seen (atomicMax): the host reads it - whenever it arrives - to pick the next call's form
};
enum : int { EIGH_PASS_ALL = 0, EIGH_PASS_NARROW = 1, EIGH_PASS_DEFERRED = 2 };

__device__ long long g_eigh_phase[8];
__device__ int g_eigh_stat[4];         // phase-timer builds: max rank, chains on the HBM/L2 path
// work the eigh kernel actually did since the last reset (bench.py's roofline sub-object): FLOP, chains, sum of ranks, sum of
// sweeps.  Two atomics per chain.  FLOP = pivoted Cholesky passes (2 n per candidate and previous column, 2 n per accepted
// pivot and remaining candidate) + Gram tiles (2048 per MFMA issued) + Jacobi (per round: rp/2 rotations of ~30 and
// (rp/2)^2 two-sided 2x2 block updates of 24) + replay (6 per pair and round) + y = mean + L t (2 n r)
__device__ unsigned long long g_eigh_work[4];
__device__ unsigned long long g_eigh_deferred;      // chains the narrow launch handed to the second one (tests)
#ifdef GPMPC_PHASE_TIMERS
#define EPH(idx) do { const long long _n = __builtin_readcyclecounter(); eph[idx] += _n - et; et = _n; } while (0)
#else
#define EPH(idx)
#endif

#ifndef GPMPC_EIGH_WPE
#define GPMPC_EIGH_WPE 2               // waves per SIMD the eigh kernel is compiled for (one wave per chain)
#endif
constexpr int EIGH_MAX_SWEEPS = 16;
constexpr double EIGH_TINY_ROT = 1e-8; // a sweep whose largest rotation tangent is below this is the last one
constexpr int EIGH_LDS_RANK = 64;      // packed 64 x 64 Gram matrix = 16.6 KB of LDS per chain: ~7 chains per CU
constexpr int EIGH_PB = 8;             // candidate pivots per pass of the pivoted Cholesky
#ifndef GPMPC_EIGH_NARROW_WPE
#define GPMPC_EIGH_NARROW_WPE 4        // waves per SIMD of the narrow launch
#endif
constexpr int EIGH_NARROW_RANK = 32;   // LDS rank cap of the narrow launch (EIGH_PASS_NARROW): 8.6 KB of LDS per chain at m T = 120
#ifndef GPMPC_EIGH_PIVOT_THRESHOLD
#define GPMPC_EIGH_PIVOT_THRESHOLD (1.0 / 16.0)
#endif
constexpr double EIGH_PIVOT_THRESHOLD = GPMPC_EIGH_PIVOT_THRESHOLD;   // accepted pivot >= this x the largest residual diagonal left
constexpr int EIGH_MAXIT = 8;          // off-diagonal 2x2 blocks per lane at the LDS rank cap: 16 * 32 / 64

typedef double double4_e __attribute__((ext_vector_type(4)));
typedef double double2_e __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) double lds_double;

__host__ __device__ inline long eigh_packed(int np) { return (long)np * (np + 1) / 2; }
// LDS form (round 4): rows 2 I and 2 I + 1 both start at column 2 I - every row starts on an even slot and the two columns of
// a pair are one aligned 16-byte access (np even: np (np + 2) / 2 slots, np / 2 more than the packed triangle)
__host__ __device__ inline long eigh_padded(int np) { return (long)np * (np + 2) / 2; }
// dynamic LDS of one chain (doubles) for an n x n covariance with the Gram matrix of ranks <= cap in LDS
__host__ __device__ inline long eigh_lds_doubles(int n, int cap) {
    const long np = (n + 1) & ~1;
    return eigh_padded(cap) + 2 + 2 * (np / 2 + 2) + (n > 256 ? n : 256) + np + (np + 3) / 4 + 2;
}
__host__ __device__ inline long eigh_slot_doubles(int n) {
    const long np = (n + 1) & ~1;
    return (long)n * n + 2 * eigh_packed((int)np) + (long)(EIGH_MAX_SWEEPS / 2) * np * np + 8;
}

// slot of the narrow launch: L for ranks <= EIGH_NARROW_RANK + EIGH_PB (a chain is deferred at the top of the pass that finds it
// beyond the cap), the rotation log for ranks <= EIGH_NARROW_RANK
__host__ __device__ inline long eigh_narrow_slot_doubles(int n) {
    return (long)n * (EIGH_NARROW_RANK + EIGH_PB) + (long)(EIGH_MAX_SWEEPS / 2) * EIGH_NARROW_RANK * EIGH_NARROW_RANK + 8;
}

// packed upper triangle, row a holds columns a..rp-1
__device__ __forceinline__ int tri_idx(int a, int b, int rp) { return a * rp - (a * (a - 1)) / 2 + (b - a); }
__device__ __forceinline__ int sym_idx(int a, int b, int rp) { return a <= b ? tri_idx(a, b, rp) : tri_idx(b, a, rp); }
// where the index at position a sits in the next round (round-robin: position 0 fixed, the others shift by one)
__device__ __forceinline__ int rr_shift(int a, int rp) { return a == 0 ? 0 : (a == rp - 1 ? 1 : a + 1); }
// The LDS path keeps the pairs of a round on ADJACENT positions (2 i, 2 i + 1): the tournament table with its top row on the
// even and its bottom row on the odd positions - top[0] fixed, bottom[0] -> top[1], the top row moves right, top[h-1] ->
// bottom[h-1], the bottom row moves left: one cycle over the other rp - 1 positions, back after rp - 1 rounds.
__device__ __forceinline__ int adj_shift(int a, int rp) {
    if (a == 0 || rp == 2) return a;
    if (a & 1) return (a == 1) ? 2 : a - 2;
    return (a == rp - 2) ? rp - 1 : a + 2;
}
__device__ __forceinline__ int adj_idx(int a, int b, int rp) {     // a <= b
    const int I = a >> 1;
    return 2 * I * (rp - I + 1) + (a & 1) * (rp - 2 * I) + (b - 2 * I);
}
__device__ __forceinline__ int adj_sym(int a, int b, int rp) { return a <= b ? adj_idx(a, b, rp) : adj_idx(b, a, rp); }

// block b of the folded enumeration of the h (h + 1) / 2 blocks I <= K; false for the padding slots of odd h
__device__ __forceinline__ bool rr_block(int b, int h, int& I, int& K) {
    const int u = b / (h + 1), off = b - u * (h + 1);
    if (off < h - u) {
        I = u;
        K = u + off;
        return true;
    }
    I = h - 1 - u;
    K = I + (off - (h - u));
    return I != u;
}

__device__ __forceinline__ double wave_max_nonneg(double v) {     // max over the wave of values >= 0 (0 is neutral)
    double t = fmax(v, dpp_f64<0x111, 0xf, 0xf>(v));
    t = fmax(t, dpp_f64<0x112, 0xf, 0xf>(v));
    t = fmax(t, dpp_f64<0x113, 0xf, 0xf>(v));
    t = fmax(t, dpp_f64<0x114, 0xf, 0xe>(t));
    t = fmax(t, dpp_f64<0x118, 0xf, 0xc>(t));
    t = fmax(t, dpp_f64<0x142, 0xa, 0xf>(t));
    t = fmax(t, dpp_f64<0x143, 0xc, 0xf>(t));
    return readlane_f64(t, 63);
}

// Jacobi rotation annihilating g_pq: tangent t (0 = skip), c, s with J = [[c, s], [-s, c]] on the (p, q) plane.
// Division free (two rsqrt_fast): with x = dq^2 + 4 gpq^2, cos(2 theta) = |dq| / sqrt(x) =: C2, u = (1 + C2) / 2:
//     c = sqrt(u),   s = sign(dq) gpq / sqrt(x) / c,   t = s / c = sign(dq) gpq / sqrt(x) / u
// (the textbook t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = dq / (2 gpq), without its division and two IEEE
// square roots: the rotation sits on the serial spine of every round)
__device__ __forceinline__ bool jacobi_rot(double gpp, double gqq, double gpq, double thr, double& t, double& c, double& s) {
    const bool rot = fabs(gpq) > thr;
    const double dq = gqq - gpp;
    const double x = rot ? fma(dq, dq, 4.0 * gpq * gpq) : 1.0;
    const double rinv = rsqrt_fast(x);
    const double u = fma(0.5, fabs(dq) * rinv, 0.5);             // in [0.5, 1]
    const double ic = rsqrt_fast(u);
    double sg = gpq * rinv;
    sg = (dq < 0.0) ? -sg : sg;
    c = rot ? u * ic : 1.0;
    s = rot ? sg * ic : 0.0;
    t = rot ? s * ic : 0.0;
    return rot;
}

// G <- J^T G J on one 2x2 block (rows pair I, columns pair K)
__device__ __forceinline__ void rot_block(double cI, double sI, double cK, double sK, double b00, double b01, double b10,
                                          double b11, double& n00, double& n01, double& n10, double& n11) {
    const double r00 = cI * b00 - sI * b10, r01 = cI * b01 - sI * b11;
    const double r10 = sI * b00 + cI * b10, r11 = sI * b01 + cI * b11;
    n00 = cK * r00 - sK * r01;
    n01 = sK * r00 + cK * r01;
    n10 = cK * r10 - sK * r11;
    n11 = sK * r10 + cK * r11;
}

// ---- B. G = L^T L (packed upper triangle), FP64 MFMA: A[i][k] = L[k0+k][I*16+i], B[k][j] = L[k0+k][J*16+j] -----------
template <bool ADJ, int NB, class GP>
__device__ __forceinline__ void eigh_gram(const double* __restrict__ Lm, int n, int r, int rp, GP G) {
    const int lane = threadIdx.x & 63;
    const int ntile = (rp + 15) / 16;
    for (int I = 0; I < ntile; ++I) {
        for (int J = I; J < ntile; ++J) {
            const int ca = I * 16 + (lane & 15), cb = J * 16 + (lane & 15), kr = lane >> 4;
            const bool va = ca < r, vb = cb < r;
            const double* pa = Lm + (long)(va ? ca : 0) * n;
            const double* pb = Lm + (long)(vb ? cb : 0) * n;
            double4_e acc = {0.0, 0.0, 0.0, 0.0};
            // NB 16-row steps per batch: their 8 NB loads are in flight together (one step at a time, every step waited for its own
            // eight loads: at the closed loop's ranks - one tile pair - this phase was eight serial round trips to L2)
            for (int k0 = 0; k0 < n; k0 += 16 * NB) {
                double av[NB][4], bv[NB][4];
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int row = k0 + 16 * b + 4 * u + kr;
                        const int rc = min(row, n - 1);
                        const double la = pa[rc], lb = pb[rc];           // (unconditional, clamped: no branch around a load)
                        av[b][u] = (va && row < n) ? la : 0.0;
                        bv[b][u] = (vb && row < n) ? lb : 0.0;
                    }
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (k0 + 16 * b < n) {                               // uniform
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[b][u], bv[b][u], acc, 0, 0, 0);
                    }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {                        // D: col = lane & 15, row = (lane >> 4) + 4 v
                const int ra = I * 16 + kr + 4 * v, cc = J * 16 + (lane & 15);
                if (ra <= cc && cc < rp) G[ADJ ? adj_idx(ra, cc, rp) : tri_idx(ra, cc, rp)] = acc[v];
            }
        }
    }
}

// ---- C (LDS). all sweeps; returns the number of rounds logged -------------------------------------------------------
// block b of the folded enumeration of the h (h - 1) / 2 OFF-DIAGONAL blocks I < K
__device__ __forceinline__ bool rr_block_off(int b, int h, int& I, int& K) {
    const int hh = h - 1;                                        // rows I = 0..h-2, row I has hh - I blocks
    const int u = b / h, off = b - u * h;
    if (off < hh - u) {
        I = u;
        K = I + 1 + off;
        return true;
    }
    I = hh - 1 - u;
    K = I + 1 + (off - (hh - u));
    return I != u;
}

// G has eigh_packed(rp) entries plus ONE pad slot: lanes with fewer than MAXIT blocks run dummy blocks on the pad slot
// with the identity rotation e_cs[h] - a round has no branch besides "lane < h owns a pair".  The lane that owns pair i
// computes its rotation AND updates the pair's diagonal block (g_pp, g_pq, g_qq: textbook update); the off-diagonal 2x2
// blocks are spread over the wave.
template <int MAXIT>
__device__ __forceinline__ long eigh_jacobi_lds(lds_double* G, int rp, double* __restrict__ rlog, double (*e_cs)[2],
                                                bool& conv, int& sweeps) {
    const int lane = threadIdx.x & 63;
    const int h = rp / 2, nslots = (h / 2) * h;                  // ceil((h - 1) / 2) * h
    const int pad = (int)eigh_padded(rp);                        // (even: a dummy block's 16-byte reads are aligned)
    // this lane's blocks: the slots of its two row halves (ri: columns 2 K, 2 K + 1 of rows 2 I and 2 I + 1 - one 16-byte read
    // each) and the four slots written (wi: the slots of the shifted positions)
    int ri[MAXIT][2], wi[MAXIT][4], bI[MAXIT], bK[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int b = lane + 64 * it;
        int I = 0, K = 1;
        const bool ok = (b < nslots) && rr_block_off(b, h, I, K);
        bI[it] = ok ? I : h;
        bK[it] = ok ? K : h;
        const int pI = 2 * I, qI = 2 * I + 1, pK = 2 * K, qK = 2 * K + 1;
        const int sp = adj_shift(pI, rp), sq = adj_shift(qI, rp), tp = adj_shift(pK, rp), tq = adj_shift(qK, rp);
        ri[it][0] = ok ? adj_idx(pI, pK, rp) : pad;
        ri[it][1] = ok ? adj_idx(qI, pK, rp) : pad;
        wi[it][0] = ok ? adj_sym(sp, tp, rp) : pad;
        wi[it][1] = ok ? adj_sym(sp, tq, rp) : pad;
        wi[it][2] = ok ? adj_sym(sq, tp, rp) : pad;
        wi[it][3] = ok ? adj_sym(sq, tq, rp) : pad;
    }
    const bool own = lane < h;
    const int p1 = own ? 2 * lane : 0, q1 = p1 + 1;              // this lane's pair
    const int dpp_ = adj_idx(p1, p1, rp), dqq_ = adj_idx(q1, q1, rp);       // (g_pq sits right behind g_pp)
    const int sp1 = adj_shift(p1, rp), sq1 = adj_shift(q1, rp);
    const int wpp_ = own ? adj_sym(sp1, sp1, rp) : pad, wpq_ = own ? adj_sym(sp1, sq1, rp) : pad;
    const int wqq_ = own ? adj_sym(sq1, sq1, rp) : pad;
    if (lane == 0) {
        G[pad] = 0.0;
        G[pad + 1] = 0.0;
        e_cs[h][0] = 1.0;
        e_cs[h][1] = 0.0;
    }
    const double thr = DBL_EPSILON * G[0];                       // G[0][0] = squared norm of the first (largest) column
    const int nround = rp - 1;
    long g = 0;
    conv = false;
    for (sweeps = 0; sweeps < EIGH_MAX_SWEEPS && !conv; ++sweeps) {
        bool rot_any = false;
        double tmax = 0.0;
        for (int rho = 0; rho < nround; ++rho, ++g) {
            // every read of the round precedes every write (a block's write slot is another block's read slot); one
            // wave, LDS operations complete in order: no barrier.  The block reads are in flight under the rotation.
            const double2_e gd = *reinterpret_cast<const __attribute__((address_space(3))) double2_e*>(G + dpp_);
            const double gpp = gd.x, gpq = gd.y, gqq = G[dqq_];
            double b[MAXIT][4];
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const double2_e r0 = *reinterpret_cast<const __attribute__((address_space(3))) double2_e*>(G + ri[it][0]);
                const double2_e r1 = *reinterpret_cast<const __attribute__((address_space(3))) double2_e*>(G + ri[it][1]);
                b[it][0] = r0.x, b[it][1] = r0.y, b[it][2] = r1.x, b[it][3] = r1.y;
            }
            double t, c, s;
            const bool rot = jacobi_rot(gpp, gqq, gpq, thr, t, c, s) && own;
            rot_any |= rot;
            tmax = fmax(tmax, rot ? fabs(t) : 0.0);
            if (own) {
                e_cs[lane][0] = c;
                e_cs[lane][1] = s;
                rlog[g * h + lane] = t;
            }
            G[wpp_] = fma(-t, gpq, gpp);
            G[wqq_] = fma(t, gpq, gqq);
            G[wpq_] = rot ? 0.0 : gpq;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const double2_e ci = *reinterpret_cast<const double2_e*>(e_cs[bI[it]]);
                const double2_e ck = *reinterpret_cast<const double2_e*>(e_cs[bK[it]]);
                double n00, n01, n10, n11;
                rot_block(ci.x, ci.y, ck.x, ck.y, b[it][0], b[it][1], b[it][2], b[it][3], n00, n01, n10, n11);
                G[wi[it][0]] = n00;
                G[wi[it][1]] = n01;
                G[wi[it][2]] = n10;
                G[wi[it][3]] = n11;
            }
        }
        // done when nothing was rotated - or when every rotation of the sweep was tiny: the entries it leaves behind are
        // O(t^2) relative (quadratic convergence), far below thr; saves the sweep that would only confirm it
        conv = !__any(rot_any) || !__any(tmax > EIGH_TINY_ROT);
    }
    return g;
}

// ---- C (HBM/L2). same rounds with the packed matrix double-buffered in the workspace (ranks beyond the LDS cap) -----
__device__ __forceinline__ long eigh_jacobi_global(double*& Gc, double*& Gn, int rp, double* __restrict__ rlog,
                                                   double (*e_cs)[2], bool& conv, int& sweeps) {
    const int lane = threadIdx.x & 63;
    const int h = rp / 2, nslots = ((h + 1) / 2) * (h + 1);
    const double thr = DBL_EPSILON * Gc[0];
    const int nround = rp - 1;
    long g = 0;
    conv = false;
    for (sweeps = 0; sweeps < EIGH_MAX_SWEEPS && !conv; ++sweeps) {
        bool rot_any = false;
        for (int rho = 0; rho < nround; ++rho, ++g) {
            for (int i = lane; i < h; i += 64) {
                const int p = i, q = rp - 1 - i;
                double t, c, s;
                rot_any |= jacobi_rot(Gc[tri_idx(p, p, rp)], Gc[tri_idx(q, q, rp)], Gc[tri_idx(p, q, rp)], thr, t, c, s);
                e_cs[i][0] = c;
                e_cs[i][1] = s;
                rlog[g * h + i] = t;
            }
            __syncthreads();
            for (int b = lane; b < nslots; b += 64) {
                int I, K;
                if (!rr_block(b, h, I, K)) continue;
                const int pI = I, qI = rp - 1 - I, pK = K, qK = rp - 1 - K;
                const int sp = rr_shift(pI, rp), sq = rr_shift(qI, rp), tp = rr_shift(pK, rp), tq = rr_shift(qK, rp);
                const double cI = e_cs[I][0], sI = e_cs[I][1], cK = e_cs[K][0], sK = e_cs[K][1];
                const double b00 = Gc[sym_idx(pI, pK, rp)], b01 = Gc[sym_idx(pI, qK, rp)];
                const double b10 = Gc[sym_idx(qI, pK, rp)], b11 = Gc[sym_idx(qI, qK, rp)];
                double n00, n01, n10, n11;
                if (I == K) {
                    const double tt = sI / cI;
                    n00 = b00 - tt * b01;
                    n11 = b11 + tt * b01;
                    n01 = n10 = (sI != 0.0) ? 0.0 : b01;
                } else {
                    rot_block(cI, sI, cK, sK, b00, b01, b10, b11, n00, n01, n10, n11);
                }
                Gn[sym_idx(sp, tp, rp)] = n00;
                Gn[sym_idx(sp, tq, rp)] = n01;
                Gn[sym_idx(sq, tp, rp)] = n10;
                Gn[sym_idx(sq, tq, rp)] = n11;
            }
            __syncthreads();
            double* tmp = Gc;
            Gc = Gn;
            Gn = tmp;
        }
        conv = !__any(rot_any);
    }
    return g;
}

// v <- Q_1 ... Q_K v: the logged rounds in reverse (Q_g = J_g Pi: un-shift the positions, then rotate the pairs)
template <bool ADJ>
__device__ __forceinline__ void eigh_replay(double* e_vec, const double* __restrict__ rlog, long gtot, int rp) {
    const int lane = threadIdx.x & 63;
    const int h = rp / 2;
    const int i0 = lane, i1 = lane + 64;                         // h <= 128: at most two pairs per lane
    const bool v0 = i0 < h, v1 = i1 < h;
    // positions of the pairs (p, q) and where their indices sit one round later
    const int p0 = ADJ ? 2 * i0 : i0, q0 = ADJ ? 2 * i0 + 1 : rp - 1 - i0;
    const int p1 = ADJ ? 2 * i1 : i1, q1 = ADJ ? 2 * i1 + 1 : rp - 1 - i1;
    const int a0 = v0 ? (ADJ ? adj_shift(p0, rp) : rr_shift(p0, rp)) : 0, b0 = v0 ? (ADJ ? adj_shift(q0, rp) : rr_shift(q0, rp)) : 0;
    const int a1 = v1 ? (ADJ ? adj_shift(p1, rp) : rr_shift(p1, rp)) : 0, b1 = v1 ? (ADJ ? adj_shift(q1, rp) : rr_shift(q1, rp)) : 0;
    for (long g1 = gtot; g1 > 0; g1 -= 8) {
        double tq0[8], tq1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long gg = g1 - 1 - u;
            tq0[u] = (gg >= 0 && v0) ? rlog[gg * h + i0] : 0.0;
            tq1[u] = (gg >= 0 && v1) ? rlog[gg * h + i1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (g1 - 1 - u >= 0) {                                // uniform
                double up0 = 0.0, uq0 = 0.0, up1 = 0.0, uq1 = 0.0;
                if (v0) up0 = e_vec[a0], uq0 = e_vec[b0];
                if (v1) up1 = e_vec[a1], uq1 = e_vec[b1];
                if (v0) {
                    const double c = rsqrt_fast(fma(tq0[u], tq0[u], 1.0)), s = tq0[u] * c;
                    e_vec[p0] = c * up0 + s * uq0;
                    e_vec[q0] = c * uq0 - s * up0;
                }
                if (v1) {
                    const double c = rsqrt_fast(fma(tq1[u], tq1[u], 1.0)), s = tq1[u] * c;
                    e_vec[p1] = c * up1 + s * uq1;
                    e_vec[q1] = c * uq1 - s * up1;
                }
            }
        }
    }
}

template <int T, int RPL, int WPE>
__global__ __launch_bounds__(64, WPE) void joint_eigh_kernel(const EighArgs a) {
    // dynamic LDS, sized by the launch (eigh_lds_doubles): packed Gram matrix + pad slot | rotations (c, s) | staging /
    // eigenvalues / raw sample | rotated base samples | ranks, candidates
    extern __shared__ __attribute__((aligned(16))) double e_dyn[];
    if (!a.force && *a.any_fail == 0) return;                    // no chain of the batch failed: Cholesky roots stand
    const GpParams& gp = a.gp;
    const int lane = threadIdx.x;
    const int n = a.m * T;
    double (*e_cs)[2] = reinterpret_cast<double (*)[2]>(e_dyn + eigh_padded(a.lds_cap) + 2);
    double* e_y = e_dyn + eigh_padded(a.lds_cap) + 2 + 2 * (((n + 1) & ~1) / 2 + 2);
    double* e_vec = e_y + (n > 256 ? n : 256);
    short* e_rank = reinterpret_cast<short*>(e_vec + ((n + 1) & ~1));
    const long nchains = a.Ns * gp.g_ny;
    double* Lm = a.ws + (long)blockIdx.x * a.ws_slot_stride;     // [n][n] column-major, r columns used
    const int np = (n + 1) & ~1;
    double* Gg0 = Lm + a.gg_off;
    double* Gg1 = Gg0 + eigh_packed(np);
    double* rlog = Lm + a.rlog_off;
    const long nwork = (a.pass == EIGH_PASS_DEFERRED) ? (long)*a.defer_count : nchains;

    for (long item = blockIdx.x; item < nwork; item += gridDim.x) {
        const long chain = (a.pass == EIGH_PASS_DEFERRED) ? (long)a.defer_list[item] : item;
        const int o = (int)(chain % gp.g_ny);
        const double* Sm = a.Sall + chain * a.S_cs;
        const int ldS = a.S_ld;
        double kmax = gp.os[o];
        if (T > 1) {
#pragma unroll
            for (int d = 0; d < T - 1; ++d) kmax = fmax(kmax, gp.os[o] * gp.inv_l2[o][d]);
        }
        const double tol = a.tol_mult * DBL_EPSILON * kmax;
        long long eph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long et = __builtin_readcyclecounter();
#ifdef GPMPC_PHASE_TIMERS
        const long long rt0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz: cycles / ticks = shader clock
#endif
        (void)eph;
        (void)et;

        // ---- A. diagonally pivoted Cholesky, left-looking, EIGH_PB pivots per pass; lane owns rows lane + 64 i --------
        // A pass picks the EIGH_PB largest residual diagonal entries as candidates, forms their EIGH_PB columns
        // against all previous columns at once (every own-row entry of L is loaded once per pass instead of once per
        // pivot: the loads' latency, not their volume, bounds this phase), then eliminates the candidates against each
        // other in registers, greedily by their exact current residual; a candidate whose residual has dropped to <=
        // tol is dropped, and the pass ends when the best candidate left is below 1/16 of the largest residual of ANY
        // row (threshold pivoting: ~10 passes for rank 50, the residual S - L L^T as small as with one pivot per pass).
        double d[RPL];
        bool pivoted[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int t = lane + 64 * i;
            d[i] = (t < n) ? Sm[(long)t * ldS + t] : 0.0;
            pivoted[i] = !(t < n);
        }
        int r = 0;
        unsigned long long work = 0;                             // FLOP of this chain (see g_eigh_work)
        bool deferred = false;
        while (r < n) {
            if (a.pass == EIGH_PASS_NARROW && r > a.defer_rank) {      // (uniform) beyond this launch's LDS cap: the second launch's
                deferred = true;
                break;
            }
            // candidates: repeated arg max; the row index rides in the low mantissa byte (ties -> lowest row)
            bool taken[RPL];
#pragma unroll
            for (int i = 0; i < RPL; ++i) taken[i] = pivoted[i];
            int cand[EIGH_PB], nc = 0;
#pragma unroll
            for (int c = 0; c < EIGH_PB; ++c) {
                cand[c] = 0;
                double key = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    if (!taken[i] && d[i] > tol) {
                        const unsigned long long kb = (__builtin_bit_cast(unsigned long long, d[i]) & ~0xFFull) |
                                                      (unsigned long long)(255 - (lane + 64 * i));
                        key = fmax(key, __builtin_bit_cast(double, kb));
                    }
                }
                key = wave_max_nonneg(key);
                if (key > 0.0 && nc == c) {                      // uniform
                    const int p = 255 - (int)(__builtin_bit_cast(unsigned long long, key) & 0xFFull);
                    cand[c] = p;
                    nc = c + 1;
#pragma unroll
                    for (int i = 0; i < RPL; ++i)
                        if (lane + 64 * i == p) taken[i] = true;
                }
            }
            if (nc == 0) break;                                  // largest residual <= tol: done
            if (lane < EIGH_PB) e_rank[lane] = (short)cand[0];
#pragma unroll
            for (int c = 1; c < EIGH_PB; ++c)
                if (lane == c) e_rank[lane] = (short)cand[c];
            // candidate columns of S
            double acc[RPL][EIGH_PB];
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int t = min(lane + 64 * i, n - 1);
#pragma unroll
                for (int c = 0; c < EIGH_PB; ++c) {
                    const int p = cand[c];
                    acc[i][c] = (c < nc) ? Sm[(long)p * ldS + t] : 0.0;   // joint_kernel mirrors S: a column is contiguous
                }
            }
            // minus the previous columns: chunks of 32 columns, the candidates' rows of L staged in LDS ([column][cand])
            for (int c0 = 0; c0 < r; c0 += 32) {
                const int cw = min(32, r - c0);
                for (int e = lane; e < 32 * EIGH_PB; e += 64) {
                    const int jj = e / EIGH_PB, c = e - jj * EIGH_PB;
                    e_y[e] = (jj < cw && c < nc) ? Lm[(long)(c0 + jj) * n + e_rank[c]] : 0.0;
                }
                for (int j0 = 0; j0 < cw; j0 += 8) {
                    double l8[8][RPL];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const long col = (long)(c0 + min(j0 + u, cw - 1)) * n;
#pragma unroll
                        for (int i = 0; i < RPL; ++i) l8[u][i] = Lm[col + min(lane + 64 * i, n - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (j0 + u < cw) {                       // uniform
                            double lp[EIGH_PB];
#pragma unroll
                            for (int c = 0; c < EIGH_PB; c += 2) {
                                const double2_e pr = *reinterpret_cast<const double2_e*>(&e_y[(j0 + u) * EIGH_PB + c]);
                                lp[c] = pr.x, lp[c + 1] = pr.y;
                            }
#pragma unroll
                            for (int i = 0; i < RPL; ++i)
#pragma unroll
                                for (int c = 0; c < EIGH_PB; ++c) acc[i][c] = fma(-l8[u][i], lp[c], acc[i][c]);
                        }
                    }
                }
            }
            work += 2ull * n * nc * r;
            // eliminate the candidates against each other, greedily by their current residual
            unsigned used = 0;
            for (int q = 0; q < nc; ++q) {
                int cs_ = -1;
                double best = tol;
#pragma unroll
                for (int c = 0; c < EIGH_PB; ++c) {
                    if (c < nc && !(used & (1u << c))) {
                        const int p = cand[c];
                        double dsel = d[0];
#pragma unroll
                        for (int i = 1; i < RPL; ++i)
                            if ((p >> 6) == i) dsel = d[i];
                        const double dc = readlane_f64(dsel, p & 63);     // exact residual of the candidate row
                        if (dc > best) {
                            best = dc;
                            cs_ = c;
                        }
                    }
                }
                if (cs_ < 0) break;                              // every remaining candidate dropped to <= tol
                // threshold pivoting: a pivot far below the largest residual left (it may sit at the noise floor of S
                // while 1e-9 residuals remain) would amplify that noise into the other rows - end the pass instead
                double gmax = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i)
                    if (!pivoted[i]) gmax = fmax(gmax, d[i]);
                gmax = wave_max_nonneg(fmax(gmax, 0.0));
                if (best < EIGH_PIVOT_THRESHOLD * gmax) break;
                used |= 1u << cs_;
                int p = 0;
                double col[RPL];
#pragma unroll
                for (int i = 0; i < RPL; ++i) col[i] = 0.0;
#pragma unroll
                for (int c = 0; c < EIGH_PB; ++c) {
                    if (c == cs_) {                              // uniform select
                        p = cand[c];
#pragma unroll
                        for (int i = 0; i < RPL; ++i) col[i] = acc[i][c];
                    }
                }
                double inv;
                const double sd = sqrt_rsqrt_fast(best, inv);
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const int t = lane + 64 * i;
                    const double l = (t == p) ? sd : (pivoted[i] ? 0.0 : col[i] * inv);   // pivoted rows: residual exactly 0
                    col[i] = l;
                    if (t < n) Lm[(long)r * n + t] = l;
                    if (t == p) pivoted[i] = true;
                    d[i] -= l * l;
                }
                // the other candidates' columns lose their component along the new column
#pragma unroll
                for (int c = 0; c < EIGH_PB; ++c) {
                    if (c < nc && !(used & (1u << c))) {         // uniform
                        const int pc = cand[c];
                        double lsel = col[0];
#pragma unroll
                        for (int i = 1; i < RPL; ++i)
                            if ((pc >> 6) == i) lsel = col[i];
                        const double lpc = readlane_f64(lsel, pc & 63);
#pragma unroll
                        for (int i = 0; i < RPL; ++i) acc[i][c] = fma(-col[i], lpc, acc[i][c]);
                    }
                }
                work += 2ull * n * nc;
                ++r;
            }
        }
        // a pass adds up to EIGH_PB pivots: a chain can leave the loop at full rank (r == n, or nothing left above tol) with
        // r beyond this launch's LDS cap WITHOUT having passed the check at the top of a pass (EIGH_NARROW_RANK < m T <=
        // EIGH_NARROW_RANK + EIGH_PB: T = 3 with H = 11..13).  It belongs to the second launch all the same
        if (a.pass == EIGH_PASS_NARROW && r > a.defer_rank) deferred = true;
        EPH(0);
        if (deferred) {
            if (lane == 0) {
                a.defer_list[atomicAdd(a.defer_count, 1)] = (int)chain;
                atomicAdd(&g_eigh_deferred, 1ull);
            }
            continue;
        }
        if (lane == 0 && a.rank_hint) atomicMax(a.rank_hint, r);

        const int rp = (r + 1) & ~1;
        int info = GPMPC_INFO_ROOT_EIGH;
#ifdef GPMPC_PHASE_TIMERS
        if (lane == 0) {
            atomicMax((int*)&g_eigh_stat[0], r);
            if (rp > a.lds_cap) atomicAdd((int*)&g_eigh_stat[1], 1);
        }
#endif
        long gtot = 0;
        if (r > 0) {
            bool conv;
            int sweeps;
            if (rp <= a.lds_cap) {
                lds_double* G = (lds_double*)e_dyn;
                eigh_gram<true, (WPE == GPMPC_EIGH_NARROW_WPE ? 2 : 4)>(Lm, n, r, rp, G);
                EPH(1);
                const int h = rp / 2, nsl = (h / 2) * h;
                // (ranks <= 16 - the closed loop's points - have at most 28 off-diagonal blocks: ONE per lane; with two the second is a dummy
                // on the pad slot whose reads, rotation and writes are a fifth of a round's instructions, and the kernel is VALU bound)
                if (nsl <= 64) gtot = eigh_jacobi_lds<1>(G, rp, rlog, e_cs, conv, sweeps);
                else if (nsl <= 128 || WPE == GPMPC_EIGH_NARROW_WPE) gtot = eigh_jacobi_lds<2>(G, rp, rlog, e_cs, conv, sweeps);     // (the narrow launch: ranks <= 32, nsl <= 128)
                else if (nsl <= 256) gtot = eigh_jacobi_lds<4>(G, rp, rlog, e_cs, conv, sweeps);
                else if (nsl <= 384) gtot = eigh_jacobi_lds<6>(G, rp, rlog, e_cs, conv, sweeps);
                else gtot = eigh_jacobi_lds<EIGH_MAXIT>(G, rp, rlog, e_cs, conv, sweeps);
                for (int i = lane; i < r; i += 64) e_y[i] = G[adj_idx(i, i, rp)];
            } else {
                double *Gc = Gg0, *Gn = Gg1;
                eigh_gram<false, 4>(Lm, n, r, rp, Gc);
                __syncthreads();
                EPH(1);
                gtot = eigh_jacobi_global(Gc, Gn, rp, rlog, e_cs, conv, sweeps);
                for (int i = lane; i < r; i += 64) e_y[i] = Gc[tri_idx(i, i, rp)];
            }
            if (!conv) info |= GPMPC_INFO_EIGH_NOCONV;
            {
                const unsigned long long nt16 = (rp + 15) / 16, kst = (n + 15) / 16 * 4, hh = rp / 2;
                work += nt16 * (nt16 + 1) / 2 * kst * 2048ull;                              // Gram
                work += (unsigned long long)sweeps * (rp - 1) * (hh * 30ull + hh * hh * 24ull + hh * 6ull);   // Jacobi + replay
                work += 2ull * n * r;
                if (lane == 0) atomicAdd(&g_eigh_work[3], (unsigned long long)sweeps);
            }
            EPH(2);
#ifdef GPMPC_PHASE_TIMERS
            eph[6] = sweeps;
#endif
            __syncthreads();
            // ---- D. ascending rank of the eigenvalues; t = W z~ ----------------------------------------------------
            for (int i = lane; i < rp; i += 64) {
                double zt = 0.0;
                if (i < r) {
                    const double lam = e_y[i];
                    int cnt = 0;
                    for (int j = 0; j < r; ++j) {
                        const double lj = e_y[j];
                        cnt += (lj < lam || (lj == lam && j < i)) ? 1 : 0;
                    }
                    e_rank[i] = (short)cnt;
                    zt = a.z[chain * (long)n + (n - r + cnt)];
                }
                e_vec[i] = zt;
            }
            __syncthreads();
            if (rp <= a.lds_cap) eigh_replay<true>(e_vec, rlog, gtot, rp);
            else eigh_replay<false>(e_vec, rlog, gtot, rp);
            __syncthreads();
        }
        EPH(3);

        // ---- E. y = mean + L t, post-processing of sample_gp (reference src/agent.py:646-708) ------------------------
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int t = lane + 64 * i;
            if (t < n) {
                double acc = 0.0;
                for (int j0 = 0; j0 < r; j0 += 8) {
                    double l8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) l8[u] = Lm[(long)min(j0 + u, r - 1) * n + t];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < r) acc = fma(l8[u], e_vec[j0 + u], acc);
                }
                e_y[t] = acc + a.mean[chain * (long)n + t];
            }
        }
        __syncthreads();
        for (int j = lane; j < a.m; j += 64) {
            double vv[T], mm[T];
            bool all_zero = (a.var_zero_thr >= 0.0);
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const long off = chain * (long)n + j * T + b;
                vv[b] = a.var[off];
                mm[b] = a.mean[off];
                all_zero = all_zero && (vv[b] <= a.var_zero_thr);
            }
#pragma unroll
            for (int b = 0; b < T; ++b) {
                double yb = all_zero ? mm[b] : e_y[j * T + b];
                if (a.apply_clip) {
                    const double sd = a.beta * sqrt(vv[b]);
                    yb = fmin(fmax(yb, mm[b] - sd), mm[b] + sd);
                }
                a.y[chain * (long)n + j * T + b] = yb;
            }
        }
        EPH(4);

        // ---- F. optional: the root itself, R = L W with eigh's column order (tests) ------------------------------------
        if (a.root) {
            double* Rout = a.root + chain * (long)n * n;
            for (int t = lane; t < n; t += 64)
                for (int c = 0; c < n - r; ++c) Rout[(long)t * n + c] = 0.0;
            for (int j = 0; j < r; ++j) {
                __syncthreads();
                for (int i = lane; i < rp; i += 64) e_vec[i] = (i == j) ? 1.0 : 0.0;
                __syncthreads();
                if (rp <= a.lds_cap) eigh_replay<true>(e_vec, rlog, gtot, rp);
                else eigh_replay<false>(e_vec, rlog, gtot, rp);
                __syncthreads();
                const int col = n - r + e_rank[j];
                for (int t = lane; t < n; t += 64) {
                    double acc = 0.0;
                    for (int k = 0; k < r; ++k) acc = fma(Lm[(long)k * n + t], e_vec[k], acc);
                    Rout[(long)t * n + col] = acc;
                }
            }
        }
#ifdef GPMPC_PHASE_TIMERS
        if ((chain == 0 || chain == 1000) && lane == 0) {    // chain 1000 runs with the CU fully loaded, chain 0 too
            eph[7] = r;
            eph[5] = __builtin_amdgcn_s_memrealtime() - rt0;
            for (int i = 0; i < 8; ++i) g_eigh_phase[i] = eph[i];
            g_eigh_phase[4] = g_eigh_stat[0] * 100000LL + g_eigh_stat[1];
        }
#endif
        if (lane == 0) {
            // After a redraw that a failed chain caused (not a forced one) every chain reports the BATCH's outcome - all retries
            // exhausted, root failed, eigendecomposition root - whatever its own Cholesky attempts had reached: chains that saw
            // the flag stopped theirs (joint.hip: abandon_root), and which ones did is a matter of timing.
            const int prev = a.info[chain];
            a.info[chain] = a.force ? (prev | info) : ((prev & ~0x000E) | (3 << 1) | GPMPC_INFO_ROOT_FAIL | info);
            atomicAdd(&g_eigh_work[0], work);
            atomicAdd(&g_eigh_work[1], 1ull);
            atomicAdd(&g_eigh_work[2], (unsigned long long)r);
        }
        __syncthreads();
    }
}

}  // namespace gpmpc
