// Eigendecomposition root of the joint posterior covariance (mode "J"), in-kernel, gfx950.
//
// Reference behaviour (gpytorch 1.13 / linear_operator at reference src/agent.py:640-641): when the Cholesky of the
// m*T x m*T posterior covariance S fails even after the three jitter retries - which is what happens on EVERY joint
// draw of the shipped params_car_residual.yaml:51 (Dyn_gp_jitter 1e-20; S is numerically singular) -
// root_decomposition() falls back, for the WHOLE batch, to
//        evals, evecs = eigh(S);   R = evecs * sqrt(clamp(evals, 0));   y = mean + R z        (ascending evals)
//
// What runs here, one 256-thread workgroup per (sample, output) chain, thread == test slot (n = m*T <= 256):
//   A. diagonally pivoted Cholesky  S ~= L L^T  (left-looking, L is n x r, stops when the largest residual diagonal
//      entry is <= tol): S has numerical rank r ~ 30..60 of n = 120, everything below tol is the round-off noise of
//      S = K** - V^T V itself (its eigenvalues there are +-1e-16 and their eigenvectors arbitrary, also for LAPACK);
//   B. G = L^T L  (r x r, FP64 MFMA 16x16x4) - G has exactly the non-zero eigenvalues of L L^T;
//   C. cyclic two-sided Jacobi on G with the round-robin ordering: every round rotates r/2 disjoint index pairs, the
//      symmetric update is done 2x2 block by 2x2 block (block (I,K) only needs the rotations of pairs I and K, so a
//      round is two barriers), G lives in LDS (upper triangle); the rotation tangents are logged to the workspace;
//   D. eigenvalues = diag(G) (>= 0 by construction: no clamp needed), ascending rank -> which base sample belongs to
//      which eigenvector (R's columns are ordered like eigh's: the n-r zero columns first); t = W z~ by applying the
//      logged rotations in reverse order to the permuted base-sample vector (W itself is never formed);
//   E. y = mean + L t  (R = L W: column j is sqrt(lambda_j) u_j), then the post-processing of sample_gp;
//   F. only when the caller asks for the root (tests): W from the log, R = L W written out.
// L W has the columns sqrt(lambda_j) u_j of the eigh root up to their signs (solver specific in LAPACK too) and up to
// the noise directions; tests align the signs per column and compare the samples.
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
    const double* Sall;      // [chains][n*n]  column-major, lower part valid
    const int* any_fail;     // device flag set by joint_kernel when a chain's jitter chain failed
    int force;
    double* ws;              // per slot: L [n*n] | G [np*np] | rotation log [EIGH_MAX_SWEEPS/2 * np*np]
    long ws_slot_stride;
    double* root;            // optional [chains][n][n] row-major
    int lds_cap;             // largest (even) rank whose Gram matrix fits the dynamic LDS
    double tol_mult;         // pivoted-Cholesky stop: residual diagonal <= tol_mult * eps * max prior variance
};

constexpr int EIGH_NT = 256;
constexpr int EIGH_MAX_SWEEPS = 16;
constexpr int EIGH_LDS_RANK = 60;      // Gram matrices up to 60 x 60 (28.8 KB) stay in LDS: four workgroups per CU

typedef double double4_e __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) double lds_double;

__host__ __device__ inline long eigh_slot_doubles(int n) {
    const long np = (n + 1) & ~1;
    return (long)n * n + np * np + (long)(EIGH_MAX_SWEEPS / 2) * np * np + 8;
}

// pair i of round rho in the round-robin schedule over rp (even) indices: every index appears once per round, every
// pair once per sweep of rp-1 rounds
__device__ __forceinline__ void rr_pair(int i, int rho, int rp, int& p, int& q) {
    int a_, b_;
    if (i == 0) {
        a_ = rp - 1;
        b_ = rho;
    } else {
        a_ = (rho + i) % (rp - 1);
        b_ = (rho + rp - 1 - i) % (rp - 1);
    }
    p = min(a_, b_);
    q = max(a_, b_);
}

// Phases B..F for one chain; GP is `double*` (G in the HBM/L2 workspace) or `lds_double*` (G in LDS).
template <int T, class GP>
__device__ __forceinline__ int eigh_tail(const EighArgs& a, long chain, int n, int r, const double* __restrict__ Lm,
                                         GP G, double* __restrict__ rlog, double* e_vec, double (*e_cs)[2],
                                         short (*e_pq)[2], short* e_rank, double* e_y, int* e_cnt) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rp = (r + 1) & ~1, h = rp / 2, ldg = rp;
    int info = GPMPC_INFO_ROOT_EIGH;
    long gtot = 0;
    if (r > 0) {
        // ---- B. G = L^T L (upper triangle), FP64 MFMA: A[i][k] = L[k0+k][I*16+i], B[k][j] = L[k0+k][J*16+j] -------
        const int ntile = (rp + 15) / 16, npairs = ntile * (ntile + 1) / 2;
        for (int tp = wv; tp < npairs; tp += EIGH_NT / 64) {
            int I = 0, rem = tp;
            while (rem >= ntile - I) {
                rem -= ntile - I;
                ++I;
            }
            const int J = I + rem;
            const int ca = I * 16 + (lane & 15), cb = J * 16 + (lane & 15), kr = lane >> 4;
            const bool va = ca < r, vb = cb < r;
            const double* pa = Lm + (long)(va ? ca : 0) * n;
            const double* pb = Lm + (long)(vb ? cb : 0) * n;
            double4_e acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int k0 = 0; k0 < n; k0 += 4) {
                const int row = k0 + kr;
                const bool vr = row < n;
                const double av = (va && vr) ? pa[row] : 0.0;
                const double bv = (vb && vr) ? pb[row] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {                        // D: col = lane & 15, row = (lane >> 4) + 4 v
                const int ra = I * 16 + kr + 4 * v, cc = J * 16 + (lane & 15);
                if (ra <= cc && cc < rp) G[ra * ldg + cc] = acc[v];
            }
        }
        __syncthreads();

        // ---- C. cyclic Jacobi, round-robin ordering, 2x2-block symmetric update ---------------------------------
        const double thr = DBL_EPSILON * G[0];                   // G[0][0] = squared norm of the first (largest) column
        const int nround = rp - 1;
        const int nitem = ((h + 1) / 2) * (h + 1);
        bool conv = false;
        int sweeps = 0;
        for (; sweeps < EIGH_MAX_SWEEPS && !conv; ++sweeps) {
            if (tid == 0) *e_cnt = 0;
            __syncthreads();
            for (int rho = 0; rho < nround; ++rho, ++gtot) {
                if (tid < h) {
                    int p, q;
                    rr_pair(tid, rho, rp, p, q);
                    const double gpp = G[p * ldg + p], gqq = G[q * ldg + q], gpq = G[p * ldg + q];
                    double t = 0.0, c = 1.0, s = 0.0;
                    if (fabs(gpq) > thr) {
                        const double zeta = (gqq - gpp) / (2.0 * gpq);
                        t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                        atomicAdd(e_cnt, 1);
                    }
                    e_cs[tid][0] = c;
                    e_cs[tid][1] = s;
                    e_pq[tid][0] = (short)p;
                    e_pq[tid][1] = (short)q;
                    rlog[gtot * h + tid] = t;
                }
                __syncthreads();
                for (int b = tid; b < nitem; b += EIGH_NT) {
                    const int u = b / (h + 1), off = b - u * (h + 1);
                    int I, K;
                    bool ok = true;
                    if (off < h - u) {
                        I = u;
                        K = u + off;
                    } else {
                        I = h - 1 - u;
                        K = I + (off - (h - u));
                        ok = (I != u);
                    }
                    if (!ok) continue;
                    const double cI = e_cs[I][0], sI = e_cs[I][1];
                    const int pI = e_pq[I][0], qI = e_pq[I][1];
                    if (I == K) {
                        if (sI != 0.0) {
                            const double t = sI / cI, gpq = G[pI * ldg + qI];
                            G[pI * ldg + pI] -= t * gpq;
                            G[qI * ldg + qI] += t * gpq;
                            G[pI * ldg + qI] = 0.0;
                        }
                    } else {
                        const double cK = e_cs[K][0], sK = e_cs[K][1];
                        if (sI == 0.0 && sK == 0.0) continue;
                        const int pK = e_pq[K][0], qK = e_pq[K][1];
                        const int i00 = min(pI, pK) * ldg + max(pI, pK), i01 = min(pI, qK) * ldg + max(pI, qK);
                        const int i10 = min(qI, pK) * ldg + max(qI, pK), i11 = min(qI, qK) * ldg + max(qI, qK);
                        const double b00 = G[i00], b01 = G[i01], b10 = G[i10], b11 = G[i11];
                        const double r00 = cI * b00 - sI * b10, r01 = cI * b01 - sI * b11;
                        const double r10 = sI * b00 + cI * b10, r11 = sI * b01 + cI * b11;
                        G[i00] = cK * r00 - sK * r01;
                        G[i01] = sK * r00 + cK * r01;
                        G[i10] = cK * r10 - sK * r11;
                        G[i11] = sK * r10 + cK * r11;
                    }
                }
                __syncthreads();
            }
            conv = (*e_cnt == 0);
            __syncthreads();
        }
        if (!conv) info |= GPMPC_INFO_EIGH_NOCONV;

        // ---- D. ascending rank of the eigenvalues; t = W z~ (logged rotations, reverse order) --------------------
        double lam = 0.0;
        if (tid < r) {
            lam = G[tid * ldg + tid];
            e_y[tid] = lam;
        }
        __syncthreads();
        if (tid < rp) {
            double zt = 0.0;
            if (tid < r) {
                int cnt = 0;
                for (int i = 0; i < r; ++i) {
                    const double li = e_y[i];
                    cnt += (li < lam || (li == lam && i < tid)) ? 1 : 0;
                }
                e_rank[tid] = (short)cnt;
                zt = a.z[chain * (long)n + (n - r + cnt)];
            }
            e_vec[tid] = zt;
        }
        __syncthreads();
        for (long g1 = gtot; g1 > 0; g1 -= 8) {
            double tq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long gg = g1 - 1 - u;
                tq[u] = (gg >= 0 && tid < h) ? rlog[gg * h + tid] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long gg = g1 - 1 - u;
                if (gg >= 0) {                                   // uniform
                    if (tid < h && tq[u] != 0.0) {
                        int p, q;
                        rr_pair(tid, (int)(gg % nround), rp, p, q);
                        const double c = 1.0 / sqrt(1.0 + tq[u] * tq[u]), s = tq[u] * c;
                        const double vp = e_vec[p], vq = e_vec[q];
                        e_vec[p] = c * vp + s * vq;
                        e_vec[q] = c * vq - s * vp;
                    }
                    __syncthreads();
                }
            }
        }
    }

    // ---- E. y = mean + L t, post-processing of sample_gp (reference src/agent.py:646-708) ----------------------------
    if (tid < n) {
        double acc = 0.0;
        for (int j0 = 0; j0 < r; j0 += 8) {
            double l8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) l8[u] = Lm[(long)min(j0 + u, r - 1) * n + tid];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < r) acc = fma(l8[u], e_vec[j0 + u], acc);
        }
        e_y[tid] = acc + a.mean[chain * (long)n + tid];
    }
    __syncthreads();
    for (int j = tid; j < a.m; j += EIGH_NT) {
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

    // ---- F. optional: the root itself, R = L W with eigh's column order (tests) --------------------------------------
    if (a.root) {
        double* Rout = a.root + chain * (long)n * n;
        __syncthreads();
        if (r > 0) {
            for (int e = tid; e < rp * rp; e += EIGH_NT) G[e] = (e / rp == e % rp) ? 1.0 : 0.0;
            __syncthreads();
            const int nround = rp - 1;
            for (long gg = gtot - 1; gg >= 0; --gg) {               // V <- J_g V, ends as W
                if (tid < h) {
                    const double t = rlog[gg * h + tid];
                    const double c = 1.0 / sqrt(1.0 + t * t);
                    int p, q;
                    rr_pair(tid, (int)(gg % nround), rp, p, q);
                    e_cs[tid][0] = c;
                    e_cs[tid][1] = t * c;
                    e_pq[tid][0] = (short)p;
                    e_pq[tid][1] = (short)q;
                }
                __syncthreads();
                for (int e = tid; e < h * rp; e += EIGH_NT) {
                    const int i = e / rp, col = e - i * rp;
                    const double c = e_cs[i][0], s = e_cs[i][1];
                    if (s != 0.0) {
                        const int p = e_pq[i][0], q = e_pq[i][1];
                        const double vp = G[p * ldg + col], vq = G[q * ldg + col];
                        G[p * ldg + col] = c * vp + s * vq;
                        G[q * ldg + col] = c * vq - s * vp;
                    }
                }
                __syncthreads();
            }
        }
        if (tid < n) {
            for (int c = 0; c < n - r; ++c) Rout[(long)tid * n + c] = 0.0;
            for (int j = 0; j < r; ++j) {
                double acc = 0.0;
                for (int k = 0; k < r; ++k) acc = fma(Lm[(long)k * n + tid], (double)G[k * ldg + j], acc);
                Rout[(long)tid * n + (n - r + e_rank[j])] = acc;
            }
        }
        __syncthreads();
    }
    return info;
}

template <int T>
__global__ __launch_bounds__(EIGH_NT, 4) void joint_eigh_kernel(const EighArgs a) {
    extern __shared__ __attribute__((aligned(16))) double e_dyn[];
    __shared__ double e_vec[256];                                // pivot row / rotated base samples
    __shared__ double e_y[256];                                  // eigenvalues / raw sample
    __shared__ double e_cs[128][2];
    __shared__ short e_pq[128][2];
    __shared__ short e_rank[256];
    __shared__ double e_rv[EIGH_NT / 64];
    __shared__ int e_ri[EIGH_NT / 64];
    __shared__ int e_cnt;
    if (!a.force && *a.any_fail == 0) return;                    // no chain of the batch failed: Cholesky roots stand
    const GpParams& gp = a.gp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.m * T;
    const long nchains = a.Ns * gp.g_ny;
    double* Lm = a.ws + (long)blockIdx.x * a.ws_slot_stride;     // [n][n] column-major, r columns used
    const int np = (n + 1) & ~1;
    double* Gg = Lm + (long)n * n;
    double* rlog = Gg + (long)np * np;

    for (long chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
        const int o = (int)(chain % gp.g_ny);
        const double* Sm = a.Sall + chain * (long)n * n;
        double kmax = gp.os[o];
        if (T > 1) {
#pragma unroll
            for (int d = 0; d < T - 1; ++d) kmax = fmax(kmax, gp.os[o] * gp.inv_l2[o][d]);
        }
        const double tol = a.tol_mult * DBL_EPSILON * kmax;

        // ---- A. diagonally pivoted Cholesky, left-looking; thread == row ------------------------------------------
        const double NEG_INF = -__builtin_huge_val();
        double d = NEG_INF;
        if (tid < n) d = Sm[(long)tid * n + tid];
        int r = 0;
        for (int k = 0; k < n; ++k) {
            double bv = d;
            int bi = tid;
#pragma unroll
            for (int off = 32; off; off >>= 1) {
                const double ov = __shfl_xor(bv, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ov > bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (lane == 0) {
                e_rv[wv] = bv;
                e_ri[wv] = bi;
            }
            __syncthreads();
            bv = e_rv[0];
            bi = e_ri[0];
#pragma unroll
            for (int w = 1; w < EIGH_NT / 64; ++w) {
                const double ov = e_rv[w];
                const int oi = e_ri[w];
                if (ov > bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            const int p = bi;
            const double dp = bv;
            if (!(dp > tol)) break;                              // uniform
            for (int j = tid; j < k; j += EIGH_NT) e_vec[j] = Lm[(long)j * n + p];
            double v = 0.0;
            if (tid < n) v = (tid >= p) ? Sm[(long)p * n + tid] : Sm[(long)tid * n + p];
            __syncthreads();
            if (tid < n) {
                for (int j0 = 0; j0 < k; j0 += 8) {
                    double l8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) l8[u] = Lm[(long)min(j0 + u, k - 1) * n + tid];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < k) v = fma(-l8[u], e_vec[j0 + u], v);
                }
                const double sd = sqrt(dp);
                const bool done = (d == NEG_INF);                // row pivoted earlier: its residual is exactly zero
                const double l = (tid == p) ? sd : (done ? 0.0 : v / sd);
                Lm[(long)k * n + tid] = l;
                d = (tid == p || done) ? NEG_INF : d - l * l;
            }
            r = k + 1;
        }
        __syncthreads();

        const int rp = (r + 1) & ~1;
        int info;
        if (rp <= a.lds_cap)
            info = eigh_tail<T>(a, chain, n, r, Lm, (lds_double*)e_dyn, rlog, e_vec, e_cs, e_pq, e_rank, e_y, &e_cnt);
        else
            info = eigh_tail<T>(a, chain, n, r, Lm, Gg, rlog, e_vec, e_cs, e_pq, e_rank, e_y, &e_cnt);
        if (tid == 0) a.info[chain] |= info;
        __syncthreads();
    }
}

}  // namespace gpmpc
