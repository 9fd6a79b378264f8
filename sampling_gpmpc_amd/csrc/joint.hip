// gpmpc_joint_sample: joint posterior draw at m test points per (sample, output) chain (mode "J", gfx950).
//
// One 256-thread workgroup per chain; chains are taken grid-stride so the HBM workspace is bounded by the grid.
// Everything is ONE left-looking factorisation over a tall matrix M whose ROWS are label slots and whose COLUMNS
// are the conditioning slots (column-major, leading dimension = rows, so "thread = row" is coalesced):
//
//      rows   : [ hallucinated slots (n_ho) | w (1) | test slots (m*T) ]
//      columns: [ real slots (n_r) | hallucinated slots (n_ho) ]
//
//   * real columns:   M[row, :n_r] = L_rr^-1 k_r(row)          (dense product with the plan's inverse; w row = w_r)
//   * column n_r+c:   M[row, n_r+c] = (k(row, c) - sum_k M[row,k] M[c,k]) / L_cc   for every row below pivot c
//                     -> hallucinated rows become L_hh, the w row becomes w_h = L_hh^-1 (y_h - L_hr w_r),
//                        test rows become V^T = (L^-1 K_o*)^T             
//   * mean = V^T w, covariance S = K** - V^T V, variance = diag(S) floored (A.8)
//   * root = Cholesky of S with the jitter-on-failure chain (A.7), y = mean + R z, post-processing of sample_gp.
#include "gpmpc_host.hpp"

namespace gpmpc {

struct JointArgs {
    GpParams gp;
    const double* plan;
    const double* X_r;
    long Ns;
    int n_h;
    const double* X_h;
    const double* Y_h;
    const int* h_slots;
    int n_ho;
    int m;
    const double* X_s;
    const double* z;
    double var_zero_thr, beta;
    int apply_clip;
    double* mean;
    double* var;
    double* y;
    double* covar;
    int* info;
    double* ws;
    long ws_chain_stride;   // doubles
    int ld;                 // rows of M (padded)
};

template <int T>
__global__ __launch_bounds__(256) void joint_kernel(const JointArgs a) {
    constexpr int D = 2;
    __shared__ double s_piv;
    __shared__ int s_flag;
    const GpParams& gp = a.gp;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n_r = gp.n_r, Tr = gp.real_has_grad ? T : 1;
    const int n_ho = a.n_ho, m = a.m, mT = m * T;
    const int n_o = n_r + n_ho;
    const int ld = a.ld;
    const int wrow = n_ho, trow0 = n_ho + 1, nrow = n_ho + 1 + mT;
    const long nchains = a.Ns * gp.g_ny;

    double* M = a.ws + (long)blockIdx.x * a.ws_chain_stride;     // [n_o][ld]
    double* Sm = M + (long)n_o * ld;                              // [mT][mT] column-major, lower part valid
    double* Rm = Sm + (long)mT * mT;                              // [mT][mT] factor attempts
    double* muv = Rm + (long)mT * mT;                             // [mT]

    for (long chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
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
        __syncthreads();

        // row descriptor: input point + task of label slot `row`
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

        // ---- real columns ---------------------------------------------------------------------------------
        for (int row = tid; row < nrow; row += nt) {
            if (row == wrow) {
                for (int i = 0; i < n_r; ++i) M[(long)i * ld + row] = w_r[i];
                continue;
            }
            const double* xp;
            int task;
            row_point(row, xp, task);
            for (int i = 0; i < n_r; ++i) {
                const int pi = i / Tr, ai = i - pi * Tr;
                double q[D];
                const double k = kern_scalar<D>(a.X_r + pi * D, xp, il2, os, q);   // r = x_real - x_row
                M[(long)i * ld + row] = kern_entry<D>(q, k, il2, ai, task);
            }
            for (int i = n_r - 1; i >= 0; --i) {          // in place: out[i] needs in[j <= i] only
                double acc = 0.0;
                for (int j = 0; j <= i; ++j) acc += LinvT[(long)j * n_r + i] * M[(long)j * ld + row];
                M[(long)i * ld + row] = acc;
            }
        }
        __syncthreads();

        // ---- hallucinated columns (left-looking) -----------------------------------------------------------
        for (int c = 0; c < n_ho; ++c) {
            const double* xc;
            int tc;
            row_point(c, xc, tc);
            const int ncol = n_r + c;
            double val[4];                                 // rows handled by this thread: tid + r*nt (nrow <= 4*nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tid + r * nt;
                val[r] = 0.0;
                if (row >= c && row < nrow) {
                    double kv;
                    if (row == wrow) {
                        kv = Yh[a.h_slots[c]];
                    } else {
                        const double* xp;
                        int task;
                        row_point(row, xp, task);
                        double q[D];
                        const double k = kern_scalar<D>(xp, xc, il2, os, q);       // r = x_row - x_c
                        kv = kern_entry<D>(q, k, il2, task, tc);
                        if (row == c) kv += gp.noise[tc];
                    }
                    double acc = 0.0;
                    for (int k = 0; k < ncol; ++k) acc += M[(long)k * ld + row] * M[(long)k * ld + c];
                    val[r] = kv - acc;
                    if (row == c) {
                        s_flag = !(val[r] > 0.0);
                        s_piv = sqrt(val[r]);
                    }
                }
            }
            __syncthreads();
            const double piv = s_piv;
            if (s_flag) info_acc |= GPMPC_INFO_TRAIN_CHOL_FAIL;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tid + r * nt;
                if (row >= c && row < nrow) M[(long)ncol * ld + row] = (row == c) ? piv : val[r] / piv;
            }
            __syncthreads();
        }

        // ---- posterior mean, covariance, variance ------------------------------------------------------------
        for (int tau = tid; tau < mT; tau += nt) {
            double acc = 0.0;
            for (int k = 0; k < n_o; ++k) acc += M[(long)k * ld + trow0 + tau] * M[(long)k * ld + wrow];
            muv[tau] = acc;
        }
        for (int e = tid; e < mT * mT; e += nt) {
            const int t2 = e / mT, t1 = e - t2 * mT;      // column t2, row t1 (column-major)
            if (t1 < t2) continue;
            const int j1 = t1 / T, b1 = t1 - j1 * T, j2 = t2 / T, b2 = t2 - j2 * T;
            double q[D];
            const double k = kern_scalar<D>(Xs + (long)j1 * D, Xs + (long)j2 * D, il2, os, q);
            double acc = 0.0;
            for (int kk = 0; kk < n_o; ++kk) acc += M[(long)kk * ld + trow0 + t1] * M[(long)kk * ld + trow0 + t2];
            Sm[e] = kern_entry<D>(q, k, il2, b1, b2) - acc;
        }
        __syncthreads();

        // ---- root: Cholesky with jitter-on-failure (A.7) ----------------------------------------------------
        int level = 0;                 // 0 = plain, 1..3 = retries
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
        while (!rooted) {
            for (int e = tid; e < mT * mT; e += nt) {
                const int t2 = e / mT, t1 = e - t2 * mT;
                if (t1 >= t2) Rm[e] = Sm[e] + ((t1 == t2) ? jit_total : 0.0);
            }
            __syncthreads();
            bool failed = false;
            for (int c = 0; c < mT; ++c) {
                double val = 0.0;
                const int row = tid;                      // mT <= 256 enforced by the host
                if (row >= c && row < mT) {
                    double acc = 0.0;
                    for (int k = 0; k < c; ++k) acc += Rm[(long)k * mT + row] * Rm[(long)k * mT + c];
                    val = Rm[(long)c * mT + row] - acc;
                    if (row == c) {
                        s_flag = !(val > 0.0);
                        s_piv = sqrt(val);
                    }
                }
                __syncthreads();
                const double piv = s_piv;
                if (s_flag) {
                    failed = true;
                    break;                                // uniform
                }
                if (row >= c && row < mT) Rm[(long)c * mT + row] = (row == c) ? piv : val / piv;
                __syncthreads();
            }
            __syncthreads();
            if (!failed) {
                rooted = true;
            } else {
                if (level == 3) break;
                // total jitter after retry i is jitter*10^i, accumulated incrementally like the library does
                const double jn = gp.jitter * ((level == 0) ? 1.0 : (level == 1) ? 10.0 : 100.0);
                const double jp = (level == 0) ? 0.0 : gp.jitter * ((level == 1) ? 1.0 : 10.0);
                jit_total += (jn - jp);
                ++level;
            }
        }
        info_acc |= (level << 1);
        if (!rooted) info_acc |= GPMPC_INFO_ROOT_FAIL;

        // ---- sample + post-processing (reference src/agent.py:641-708) --------------------------------------
        const double* zc = a.z + chain * (long)mT;
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
                double acc = 0.0;
                if (rooted) {
                    for (int c = 0; c <= tau; ++c) acc += Rm[(long)c * mT + tau] * zc[c];
                } else {
                    acc = __builtin_nan("");
                }
                yy[b] = acc + mm[b];
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
        // OR-reduce info over the block
        __syncthreads();
        if (tid == 0) s_flag = 0;
        __syncthreads();
        if (info_acc) atomicOr(&s_flag, info_acc);
        __syncthreads();
        if (tid == 0) a.info[chain] = s_flag;
        __syncthreads();
    }
}

static long joint_chain_doubles(int n_r, int n_ho, int m, int T, int* ld_out) {
    const int mT = m * T;
    int ld = n_ho + 1 + mT;
    ld = (ld + 3) & ~3;
    if (ld_out) *ld_out = ld;
    return (long)(n_r + n_ho) * ld + 2L * mT * mT + mT + 4;
}

static long joint_grid(long nchains) {
    const long cap = 256L * 8;
    return nchains < cap ? nchains : cap;
}

}  // namespace gpmpc

using namespace gpmpc;

extern "C" {

size_t gpmpc_joint_workspace_bytes(const gpmpc_gp_desc_t* gp, int64_t Ns, int32_t n_ho, int32_t m) {
    if (check_gp(gp) != GPMPC_OK) return 0;
    const long per = joint_chain_doubles(observed_real_slots(gp), n_ho, m, gp->T, nullptr);
    return align_up((size_t)joint_grid(Ns * gp->g_ny) * per * sizeof(double), 256);
}

int gpmpc_joint_sample(const gpmpc_gp_desc_t* gp, const void* plan, const double* X_r, int64_t Ns, int32_t n_h,
                       const double* X_h, const double* Y_h, const int32_t* h_slots, int32_t n_ho, int32_t m,
                       const double* X_s, const double* z, double var_zero_thr, double beta, int32_t apply_clip,
                       double* mean, double* var, double* y, double* covar, int32_t* info, void* ws,
                       size_t ws_bytes, void* stream) {
    if (int rc = check_gp(gp)) return rc;
    if (!plan || !X_r || !X_s || !z || !mean || !var || !y || !info || !ws)
        return fail(GPMPC_E_ARG, "gpmpc_joint_sample: NULL pointer");
    if (Ns < 1 || m < 1 || n_ho < 0 || n_h < 0) return fail(GPMPC_E_ARG, "gpmpc_joint_sample: bad sizes");
    if (n_ho > 0 && (!X_h || !Y_h || !h_slots)) return fail(GPMPC_E_ARG, "gpmpc_joint_sample: hallucinated data missing");
    if (n_ho > n_h * gp->T) return fail(GPMPC_E_ARG, "gpmpc_joint_sample: n_ho > n_h*T");
    if (gp->D != 2) return fail(GPMPC_E_UNSUPPORTED, "only D = 2 is instantiated");
    const int mT = m * gp->T;
    if (mT > 256) return fail(GPMPC_E_UNSUPPORTED, "joint: m*T > 256");
    if (n_ho + 1 + mT > 4 * 256) return fail(GPMPC_E_UNSUPPORTED, "joint: more than 1024 label rows per chain");
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
    a.ws_chain_stride = joint_chain_doubles(a.gp.n_r, n_ho, m, gp->T, &a.ld);
    const long grid = joint_grid(Ns * gp->g_ny);
    if (ws_bytes < (size_t)grid * a.ws_chain_stride * sizeof(double))
        return fail(GPMPC_E_WORKSPACE, "gpmpc_joint_sample: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (gp->T == 1)
        hipLaunchKernelGGL(joint_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, a);
    else if (gp->T == 3)
        hipLaunchKernelGGL(joint_kernel<3>, dim3((unsigned)grid), dim3(256), 0, st, a);
    else
        return fail(GPMPC_E_UNSUPPORTED, "joint: only T = 1 and T = 3 (D = 2) are instantiated");
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // extern "C"
