// joint_panel_kernel: the joint draw of joint.hip for launches whose computed label rows fit RW <= 256 row slots (with the
// factor cache that is every call of the closed loop: new hallucinated rows + w + test rows).  Included by joint.hip.
//
// joint_kernel streams the thread's own row once per 16-column block: 16 FMAs per 8 bytes = 4 FLOP per byte, and at
// ~10 bytes per cycle and CU of HBM/L2 stream (MI355X_MICROARCH.md) that bounds the update at about a third of the
// FP64 pipe.  Here a wave holds 16 ROWS x G COLUMN GROUPS (G = 4: lane = (DPP row = column group, lane & 15 = row)):
// the sixteen lanes of a DPP row own sixteen different rows and the same 16 pivot columns, so
//   * a wave-wide load of "my row's entry of column k" touches 16 x 8 contiguous bytes and feeds 16 x 64 FMAs
//     (16 FLOP per byte), with no cross-wave sharing and no LDS staging of the row stream;
//   * the pivot-row entries of column k for all 16 G columns of the outer block sit in LDS one per lane
//     (Pt[k][lane & (16 G - 1)]): one ds_read_b64 per column, broadcast inside the DPP row by v_fmac_f64_dpp
//     row_newbcast exactly as in joint_kernel.
// The column blocks become OUTER blocks of 16 G columns: one pass over k < (columns before the outer block) for all G
// groups at once, then the G inner 16-column blocks in sequence (diagonal block, row-wise solve, and the 16 columns of
// the finished inner block applied to the groups behind it - the same update routine over k in that block, the
// rows' new entries read back from the workspace).  Every accumulator sees the FMAs of joint_kernel in the same order:
// the two kernels give bit-identical results (tests/test_hip_closed_loop.py compares them).
#pragma once

namespace gpmpc {

// acc[q] -= sum_{kbeg <= k < kend} W[row][k] * P[lane's column][k] for the lane's 16 columns.  wr = &W[row][0] (column
// stride ld); pivot-row entry (slot q of the outer block, column k) at P[q * p_rs + k * p_cs], q < ncols.  Called by the
// whole workgroup (barriers inside); lanes whose result is not needed run along (their rows' pointers are clamped).
template <int G, int NT, int KC>
__device__ __forceinline__ void panel_update(const double* __restrict__ wr, int ld, int kbeg, int kend,
                                             const double* P, long p_rs, long p_cs, int ncols,
                                             double (&acc)[16], double (*Pt)[KC][16 * G]) {
    constexpr int OB = 16 * G, KU = 4, RD = 2;
    static_assert(OB * KC <= NT, "one staged pivot entry per thread");
    static_assert(RD * KU == KC, "the row ring turns once per pivot chunk");
    if (kend <= kbeg) return;                                        // uniform
    const int tid = threadIdx.x;
    const int lcol = (((tid >> 4) % G) << 4) | (tid & 15);           // this lane's column inside the outer block
    // The pivot entry this thread stages: slot tid % OB of column tid / OB of the chunk, so that its place in the LDS
    // buffer is the thread's own index.  Everything in the loop is branch-free (clamped addresses, selected values):
    // the compiler's wait-count insertion falls back to vmcnt(0) at control-flow joins, and a spilled address reloaded
    // inside the loop waits for ALL loads in flight - either one serialises the row stream.
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const bool stager = wbase < OB * KC;                             // wave-uniform
    const int pq = tid % OB, pkk = tid / OB;
    const double* pp = P + pq * p_rs + (long)(kbeg + pkk) * p_cs;    // the entry this thread stages for the first chunk
    int prem = (stager && pq < ncols) ? kend - kbeg - pkk : 0;       // > 0: that entry exists
    const long pstep = (long)KC * p_cs;
    const double* prd = &Pt[0][0][lcol];
    // The prefetch of the next chunk's pivot entry is issued and awaited BY HAND: the compiler sinks a plain load to its
    // use (the LDS write at the end of the chunk, a full memory latency exposed per chunk, and its vmcnt(0) drains the
    // row ring).  An untracked load in flight only makes the compiler's own vmcnt waits stricter.  At least 2 KU row
    // loads are issued between issue and landing (fetch_rows ends in a fence), so vmcnt(KU) covers it.
    double pv;
    auto issue_piv = [&]() {
        const double* src = (prem > 0) ? pp : P;                     // entries that do not exist: any valid address, never consumed
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pv) : "v"(src) : "memory");
        pp += pstep;
        prem -= KC;
    };
    auto land_piv = [&](int boff) {                                  // staging address = reading address + a wave-uniform offset (G = 4)
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pv) : "n"(KU) : "memory");
        if constexpr (OB == 64) {
            if (stager) const_cast<double*>(prd)[boff + wbase] = pv;
        } else {
            if (stager) (&Pt[0][0][0])[boff + tid] = pv;
        }
    };
    double ring[RD][KU];
    auto fetch_rows = [&](int kbase, double (&mreg)[KU]) {
#pragma unroll
        for (int j = 0; j < KU; ++j) mreg[j] = wr[(long)min(kbase + j, kend - 1) * ld];   // clamped: never consumed beyond kend
        asm volatile("" ::: "memory");                               // issued here, not sunk to the first use
    };
    issue_piv();
    fetch_rows(kbeg, ring[0]);
    land_piv(0);
    __syncthreads();
    int boff = 0;                                                    // doubles: 0 / KC * OB (the buffer being consumed)
    int k0 = kbeg;
    for (; k0 + KC <= kend; k0 += KC) {                              // full chunks
        issue_piv();                                                 // next chunk's pivot entries, in flight during this one
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            fetch_rows(k0 + (r + RD - 1) * KU, ring[(r + RD - 1) % RD]);
            double pk[KU];
#pragma unroll
            for (int j = 0; j < KU; ++j) pk[j] = prd[boff + (r * KU + j) * OB];
#pragma unroll
            for (int j = 0; j < KU; ++j) fmac16_dpp(acc, pk[j], ring[r][j]);
        }
        boff ^= KC * OB;
        land_piv(boff);
        __syncthreads();
    }
    if (k0 < kend) {                                                 // the last, partial chunk
        const int kc = kend - k0;
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            if (r + 1 < RD) fetch_rows(k0 + (r + 1) * KU, ring[r + 1]);
#pragma unroll
            for (int j = 0; j < KU; ++j) {
                if (r * KU + j < kc) {
                    const double pk = prd[boff + (r * KU + j) * OB];
                    fmac16_dpp(acc, pk, ring[r][j]);
                }
            }
        }
        __syncthreads();
    }
}

#ifdef GPMPC_PHASE_TIMERS
#define PPH(idx) do { const long long _n = __builtin_readcyclecounter(); pph[idx] += _n - pt; pt = _n; } while (0)
#else
#define PPH(idx)
#endif

template <int T, int G, int NT>
__global__ __launch_bounds__(NT, 4) void joint_panel_kernel(const JointArgs a) {
    constexpr int D = 2, NB = 16, RPT = 1;
    constexpr int OB = NB * G;                                    // columns per outer block
    constexpr int KC = 8;
    __shared__ __attribute__((aligned(16))) double Pt[2][KC][OB];
    __shared__ __attribute__((aligned(16))) double piv[KC][NB];   // the root phase's staging (joint_tail.inc)
    __shared__ double blk[NB][NB + 1];
    __shared__ double dinv_s[NB];
    __shared__ int s_flag;
    __shared__ int s_info;
    __shared__ __attribute__((aligned(16))) double colx[OB][D];   // input point / task / label of the outer block's pivot slots
    __shared__ int colt[OB];
    __shared__ double coly[OB];
    const GpParams& gp = a.gp;
    const int tid = threadIdx.x;
    constexpr int nt = NT;
    const int lane = tid & 63;
    const int g = (lane >> 4) % G;                                                    // column group
    const int rslot = ((tid >> 6) * (4 / G) + (lane >> 4) / G) * 16 + (lane & 15);    // row slot, 0 .. NT/G-1
    const int n_r = gp.n_r, Tr = gp.real_has_grad ? T : 1;
    const int n_ho = a.n_ho, m = a.m, mT = m * T;
    const int n_o = n_r + n_ho;
    const int ld = a.ld;
    const int wrow = n_ho, trow0 = n_ho + 1, nrow = n_ho + 1 + mT;
    const long nchains = a.Ns * gp.g_ny;

    double* M = a.ws + (long)blockIdx.x * a.ws_chain_stride;     // [n_o][ld]   column-major, rows = label slots
    double* Rm = M + (long)n_o * ld;                              // [mT][mT]    factor attempts
    double* muv = Rm + (long)mT * mT;                             // [mT]
    double* yv = muv + mT;                                        // [mT]   mean + R z

    for (long chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
        double* Sm = a.Sall + chain * (long)mT * mT;
        double* fc = a.fcache ? a.fcache + chain * a.fc_stride : nullptr;
        double* fdinv = fc ? fc + (long)a.fc_cap * a.fc_cs : nullptr;
        const int n_c = fc ? a.n_c : 0, CS = a.fc_cs;
        const bool fill = fc && n_ho <= a.fc_cap;
        const long s = chain / gp.g_ny;
        const int o = (int)(chain - s * gp.g_ny);
        const double* w_r = plan_w(a.plan, gp, o);
        const double* Lrr = plan_L(a.plan, gp, o);
        const double* Xh = a.X_h ? a.X_h + chain * (long)a.n_h * D : nullptr;
        const double* Yh = a.Y_h ? a.Y_h + chain * (long)a.n_h * T : nullptr;
        const double* Xs = a.X_s + chain * (long)m * D;
        double il2[D];
#pragma unroll
        for (int d = 0; d < D; ++d) il2[d] = gp.inv_l2[o][d];
        const double os = gp.os[o];
        int info_acc = 0;
        if (tid == 0) s_info = 0;
        __syncthreads();
#ifdef GPMPC_PHASE_TIMERS
        long long pph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long pt = __builtin_readcyclecounter();
#endif

        auto row_point = [&](int r, const double*& xp, int& task) {
            if (r < n_ho) {
                const int sl = a.h_slots[r];
                const int j = sl / T;
                task = sl - j * T;
                xp = Xh + (long)j * D;
            } else {
                const int tau = r - trow0;
                const int j = tau / T;
                task = tau - j * T;
                xp = Xs + (long)j * D;
            }
        };

        const int row = n_c + rslot;                               // the cached rows have no thread
        const bool valid = row < nrow;
        const double* wr = M + min(row, nrow - 1);
        double rowx[D] = {0.0, 0.0};
        int rowt = 0;
        if (valid && row != wrow) {
            const double* xp;
            row_point(row, xp, rowt);
#pragma unroll
            for (int d = 0; d < D; ++d) rowx[d] = xp[d];
        }

        // ---- real columns: M[row, :n_r] = L_rr^-1 k_r(row)  (w row = w_r) ------------------------------------------
        for (int C0 = 0; C0 < n_r; C0 += OB) {
            const int ncols = min(OB, n_r - C0);
            if (tid < OB) {
                const int i = min(C0 + tid, n_r - 1);
                const int pi = i / Tr;
                colx[tid][0] = a.X_r[pi * D];
                colx[tid][1] = a.X_r[pi * D + 1];
                colt[tid] = i - pi * Tr;
                coly[tid] = w_r[i];
            }
            __syncthreads();
            const int nbg = max(0, min(NB, ncols - NB * g));
            double acc[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                acc[q] = 0.0;
                if (q < nbg && valid && row != wrow) {
                    double qq[D];
                    const double k = kern_scalar<D>(colx[NB * g + q], rowx, il2, os, qq);    // r = x_real - x_row
                    acc[q] = kern_entry<D>(qq, k, il2, colt[NB * g + q], rowt);
                }
                if ((q & 3) == 3) asm volatile("" ::: "memory");
            }
            const double* Pb = Lrr + (long)C0 * n_r;
            panel_update<G, NT, KC>(wr, ld, 0, C0, Pb, n_r, 1, ncols, acc, Pt);
            for (int j = 0; j < G; ++j) {
                const int cj = C0 + NB * j, nbj = min(NB, ncols - NB * j);
                if (nbj <= 0) break;                                 // uniform
                for (int e = tid; e < NB * NB; e += nt) {
                    const int q = e / NB, c = e - q * NB;
                    if (c < q) blk[q][c] = (q < nbj) ? Lrr[(long)(cj + q) * n_r + cj + c] : 0.0;
                }
                if (tid < NB) {
                    const int i = min(cj + tid, n_r - 1);
                    dinv_s[tid] = 1.0 / Lrr[(long)i * n_r + i];
                }
                __syncthreads();
                if (g == j && valid) {
                    double x[NB];
                    block_solve<NB>(acc, x, blk, dinv_s, nbj, NB);
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (q < nbj) M[(long)(cj + q) * ld + row] = (row == wrow) ? coly[NB * j + q] : x[q];
                    if (fill && row < n_ho) {
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q < nbj) fc[(long)row * CS + cj + q] = x[q];
                    }
                }
                __syncthreads();
                if (NB * (j + 1) < ncols) panel_update<G, NT, KC>(wr, ld, cj, cj + nbj, Pb, n_r, 1, ncols, acc, Pt);
            }
        }
        __syncthreads();
        PPH(0);

        // ---- hallucinated columns: outer blocks of OB, cached region first (pivot rows from the factor cache) -------
        for (int C0 = 0; C0 < n_ho;) {
            const bool cached = C0 < n_c;                            // uniform
            const int ncols = min(OB, (cached ? n_c : n_ho) - C0);
            if (tid < OB) {
                int tc = 0;
                double x0 = 0.0, x1 = 0.0, yl = 0.0;
                if (tid < ncols) {
                    const double* xc;
                    row_point(C0 + tid, xc, tc);
                    x0 = xc[0];
                    x1 = xc[1];
                    yl = Yh[a.h_slots[C0 + tid]];
                }
                colx[tid][0] = x0;
                colx[tid][1] = x1;
                colt[tid] = tc;
                coly[tid] = yl;
            }
            __syncthreads();
            const int cg = C0 + NB * g, nbg = max(0, min(NB, ncols - NB * g));
            double acc[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) acc[q] = 0.0;
            if (valid && row >= cg) {
                if (row == wrow) {
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (q < nbg) acc[q] = coly[NB * g + q];
                } else {
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        if (q < nbg) {
                            const int tc = colt[NB * g + q];
                            double qq[D];
                            const double k = kern_scalar<D>(rowx, colx[NB * g + q], il2, os, qq);   // r = x_row - x_c
                            double kv = kern_entry<D>(qq, k, il2, rowt, tc);
                            if (row == cg + q) kv += gp.noise[tc];
                            acc[q] = kv;
                        }
                        if ((q & 3) == 3) asm volatile("" ::: "memory");
                    }
                }
            }
            const double* Pb = cached ? fc + (long)C0 * CS : M + C0;
            const long p_rs = cached ? CS : 1, p_cs = cached ? 1 : ld;
            PPH(1);
            panel_update<G, NT, KC>(wr, ld, 0, n_r + C0, Pb, p_rs, p_cs, ncols, acc, Pt);
            PPH(2);
            for (int j = 0; j < G; ++j) {
                const int cj = C0 + NB * j, nbj = min(NB, ncols - NB * j);
                if (nbj <= 0) break;                                 // uniform
                if (cached) {
                    for (int e = tid; e < NB * NB; e += nt) {
                        const int q = e / NB, c = e - q * NB;
                        if (c <= q) blk[q][c] = fc[(long)(cj + q) * CS + n_r + cj + c];
                    }
                    if (tid < NB) dinv_s[tid] = fdinv[cj + tid];
                    if (tid == 0) s_flag = 0;
                } else {
                    if (g == j && row >= cj && row < cj + nbj) {
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q <= row - cj) blk[row - cj][q] = acc[q];
                    }
                    __syncthreads();
                    if (tid < 64) block_factor<NB>(blk, dinv_s, nbj, &s_flag);
                }
                __syncthreads();
                PPH(3);
                if (s_flag) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
                if (fill && !cached) {                               // the diagonal block as block_factor left it, and 1/diag
                    for (int e = tid; e < NB * NB; e += nt) {
                        const int q = e / NB, c = e - q * NB;
                        if (c <= q && q < nbj) fc[(long)(cj + q) * CS + n_r + cj + c] = blk[q][c];
                    }
                    if (tid < nbj) fdinv[cj + tid] = dinv_s[tid];
                }
                if (g == j && valid && row >= cj) {
                    double x[NB];
                    block_solve<NB>(acc, x, blk, dinv_s, nbj, (row < n_ho) ? row - cj : NB);
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (q < nbj) M[(long)(n_r + cj + q) * ld + row] = x[q];
                    if (fill && row >= cj + nbj && row < n_ho) {
#pragma unroll
                        for (int q = 0; q < NB; ++q)
                            if (q < nbj) fc[(long)row * CS + n_r + cj + q] = x[q];
                    }
                }
                __syncthreads();
                if (NB * (j + 1) < ncols)
                    panel_update<G, NT, KC>(wr, ld, n_r + cj, n_r + cj + nbj, Pb, p_rs, p_cs, ncols, acc, Pt);
                PPH(4);
            }
            C0 += ncols;
        }

        // ---- posterior mean and covariance: pivot slots [w | test slots] = rows wrow .. wrow+mT of M ----------------
        // slot 0 gives mu = V^T w (accumulator starts at 0, mu = -acc), slot 1+t2 gives S[:, t2] = K** - V^T V
        {
            const int t1 = rslot;
            const double* wt = M + trow0 + min(t1, mT - 1);
            const int j1 = min(t1, mT - 1) / T, b1 = min(t1, mT - 1) - j1 * T;
            for (int C0 = 0; C0 < mT + 1; C0 += OB) {
                const int ncols = min(OB, mT + 1 - C0);
                double acc[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    acc[q] = 0.0;
                    const int sl = C0 + NB * g + q;                  // pivot slot; test slot t2 = sl - 1
                    if (sl >= 1 && sl <= mT && t1 < mT && t1 >= sl - 1) {
                        const int t2 = sl - 1, j2 = t2 / T, b2 = t2 - j2 * T;
                        double qq[D];
                        const double k = kern_scalar<D>(Xs + (long)j1 * D, Xs + (long)j2 * D, il2, os, qq);
                        acc[q] = kern_entry<D>(qq, k, il2, b1, b2);
                    }
                    if ((q & 3) == 3) asm volatile("" ::: "memory");
                }
                panel_update<G, NT, KC>(wt, ld, 0, n_o, M + wrow + C0, 1, ld, ncols, acc, Pt);
                if (t1 < mT) {
#pragma unroll
                    for (int q = 0; q < NB; ++q) {
                        const int sl = C0 + NB * g + q;
                        if (sl == 0) muv[t1] = -acc[q];
                        else if (sl <= mT && t1 >= sl - 1) Sm[(long)(sl - 1) * mT + t1] = acc[q];
                    }
                }
            }
        }
        __syncthreads();
        PPH(5);
#pragma push_macro("JPH")
#undef JPH
#define JPH(idx)
#include "joint_tail.inc"
#pragma pop_macro("JPH")
        PPH(6);
#ifdef GPMPC_PHASE_TIMERS
        if (blockIdx.x == 0 && tid == 0)
            for (int i = 0; i < 8; ++i) g_joint_phase[i] = pph[i];
#endif
        if (info_acc) atomicOr(&s_info, info_acc);
        __syncthreads();
        if (tid == 0) a.info[chain] = s_info;
        __syncthreads();
    }
}

}  // namespace gpmpc
