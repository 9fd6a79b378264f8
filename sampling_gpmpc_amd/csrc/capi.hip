// libgpmpc_hip.so - C-ABI basics, the shared real-data plan, device self test.  gfx950 only.
#include "gpmpc_host.hpp"

namespace gpmpc {

std::string& last_error() {
    static thread_local std::string s;
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
// plan: factorise K_rr + Sigma_r once per output (shared by all samples).  One workgroup per output; the matrix
// lives in LDS (n_r <= 143 -> <= 160 KiB).  Right-looking Cholesky; then L^-1 column by column (thread per column).
// ---------------------------------------------------------------------------------------------------------------
// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 16, row stride 16) in LDS, executed by ONE WAVE
// (lane k owns element k of the rotated rows / columns): A -> diagonal (eigenvalues), V -> eigenvectors in columns.
// High relative accuracy for the SPD kernel matrices it is used on.
__device__ void jacobi_eig16_wave(double* A, int n, double* V) {
    const int lane = threadIdx.x & 63;
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    for (int e = lane; e < n * n; e += 64) V[(e / n) * 16 + (e % n)] = ((e / n) == (e % n)) ? 1.0 : 0.0;
    sync();
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dia = 0.0;
        if (lane < n) {
            dia = A[lane * 16 + lane] * A[lane * 16 + lane];
            for (int j = 0; j < lane; ++j) off += A[lane * 16 + j] * A[lane * 16 + j];
        }
        off = wave_sum_shfl(off);
        dia = wave_sum_shfl(dia);
        if (off <= 1e-34 * dia) break;                       // uniform
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * 16 + q];            // uniform (broadcast reads)
                if (apq != 0.0) {
                    const double theta = (A[q * 16 + q] - A[p * 16 + p]) / (2.0 * apq);
                    const double t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                    sync();
                    if (lane < n) {                          // columns p, q (element k = lane) and the eigenvectors
                        const int k = lane;
                        const double akp = A[k * 16 + p], akq = A[k * 16 + q];
                        A[k * 16 + p] = c * akp - sn * akq;
                        A[k * 16 + q] = sn * akp + c * akq;
                        const double vkp = V[k * 16 + p], vkq = V[k * 16 + q];
                        V[k * 16 + p] = c * vkp - sn * vkq;
                        V[k * 16 + q] = sn * vkp + c * vkq;
                    }
                    sync();
                    if (lane < n) {                          // rows p, q
                        const int k = lane;
                        const double apk = A[p * 16 + k], aqk = A[q * 16 + k];
                        A[p * 16 + k] = c * apk - sn * aqk;
                        A[q * 16 + k] = sn * apk + c * aqk;
                    }
                    sync();
                }
            }
    }
    sync();
}

template <int D>
__global__ __launch_bounds__(256) void plan_kernel(GpParams gp, const double* __restrict__ X_r,
                                                   const double* __restrict__ Y_r, double* __restrict__ plan,
                                                   int* __restrict__ info, int no_rec) {
    extern __shared__ __attribute__((aligned(16))) double A[];
    __shared__ int s_fail;
    const int o = blockIdx.x;
    const int n = gp.n_r;
    const int Tr = gp.real_has_grad ? gp.T : 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double* inv_l2 = gp.inv_l2[o];
    if (tid == 0) s_fail = 0;

    for (int e = tid; e < n * n; e += nt) {
        const int s1 = e / n, s2 = e - s1 * n;
        const int i1 = s1 / Tr, a1 = s1 - i1 * Tr;
        const int i2 = s2 / Tr, a2 = s2 - i2 * Tr;
        double q[D];
        const double k = kern_scalar<D>(X_r + i1 * D, X_r + i2 * D, inv_l2, gp.os[o], q);
        double v = kern_entry<D>(q, k, inv_l2, a1, a2);
        if (s1 == s2) v += gp.noise[a1];
        A[e] = v;
    }
    __syncthreads();

    for (int j = 0; j < n; ++j) {
        const double d = A[j * n + j];
        if (!(d > 0.0)) {
            if (tid == 0) s_fail = 1;
            break;                                   // uniform: every thread reads the same d
        }
        const double sj = sqrt(d);
        __syncthreads();
        for (int i = j + 1 + tid; i < n; i += nt) A[i * n + j] /= sj;
        if (tid == 0) A[j * n + j] = sj;
        __syncthreads();
        // trailing update: A[i][k] -= L[i][j] L[k][j]   for j < k <= i
        const int m = n - j - 1;
        for (int e = tid; e < m * m; e += nt) {
            const int ii = e / m, kk = e - ii * m;
            if (kk <= ii) {
                const int i = j + 1 + ii, k = j + 1 + kk;
                A[i * n + k] -= A[i * n + j] * A[k * n + j];
            }
        }
        __syncthreads();
    }
    __syncthreads();
    double* L = plan + o * gp.plan_stride;
    double* LinvT = L + (long)n * n;
    double* w = LinvT + (long)n * n;
    double* alpha = w + n;
    if (s_fail) {
        if (tid == 0) info[o] = GPMPC_INFO_TRAIN_CHOL_FAIL;
        for (long e = tid; e < gp.plan_stride; e += nt) L[e] = __builtin_nan("");
        return;
    }
    if (tid == 0) info[o] = 0;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, k = e - i * n;
        L[e] = (k <= i) ? A[e] : 0.0;
    }
    // column c of L^-1:  x_c = 1/L_cc ; x_i = -(sum_{k=c}^{i-1} L_ik x_k) / L_ii   (stored LinvT[c*n + i])
    for (int c = tid; c < n; c += nt) {
        double* x = LinvT + (long)c * n;
        for (int i = 0; i < c; ++i) x[i] = 0.0;
        x[c] = 1.0 / A[c * n + c];
        for (int i = c + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += A[i * n + k] * x[k];
            x[i] = -s / A[i * n + i];
        }
    }
    __syncthreads();
    __threadfence_block();
    for (int i = tid; i < n; i += nt) {                 // w = L^-1 y
        double s = 0.0;
        for (int j = 0; j <= i; ++j) {
            const int pj = j / Tr, aj = j - pj * Tr;
            s += LinvT[(long)j * n + i] * Y_r[((long)o * gp.N_r + pj) * gp.T + aj];
        }
        w[i] = s;
    }
    __syncthreads();
    __threadfence_block();
    for (int j = tid; j < n; j += nt) {                 // alpha = L^-T w
        double s = 0.0;
        for (int i = j; i < n; ++i) s += LinvT[(long)j * n + i] * w[i];
        alpha[j] = s;
    }
    // ---- grid root (see plan_doubles_per_output): eigen-decomposition of the two axis kernel matrices ------------------
    if (plan_has_grid_root(gp.grid_n0, gp.grid_n1, gp.real_has_grad)) {
        __shared__ double Ka[16 * 16], Kb[16 * 16], Va[16 * 16], Vb[16 * 16];
        const int n0 = gp.grid_n0, n1 = gp.grid_n1;
        double* Qa = alpha + n;
        double* Qb = Qa + n0 * n0;
        double* dsc = Qb + n1 * n1;
        double* wE = dsc + n0 * n1;
        double* m1 = wE + n0 * n1;
        double* m2 = m1 + n0 * n1;
        for (int e = tid; e < n0 * n0; e += nt) {
            const int i = e / n0, j = e - i * n0;
            const double r = X_r[(long)(i * n1) * D + 0] - X_r[(long)(j * n1) * D + 0];
            Ka[i * 16 + j] = exp(-0.5 * r * r * inv_l2[0]);
        }
        for (int e = tid; e < n1 * n1; e += nt) {
            const int i = e / n1, j = e - i * n1;
            const double r = X_r[(long)i * D + 1] - X_r[(long)j * D + 1];
            Kb[i * 16 + j] = exp(-0.5 * r * r * inv_l2[1]);
        }
        __syncthreads();
        if (tid < 64) jacobi_eig16_wave(Ka, n0, Va);        // eigenvalues end up on the diagonal of Ka / Kb,
        else if (tid < 128) jacobi_eig16_wave(Kb, n1, Vb);  // eigenvectors in the columns of Va / Vb (one wave each)
        __syncthreads();
        for (int e = tid; e < n0 * n0; e += nt) Qa[e] = Va[(e / n0) * 16 + (e % n0)];
        for (int e = tid; e < n1 * n1; e += nt) Qb[e] = Vb[(e / n1) * 16 + (e % n1)];
        for (int r = tid; r < n0 * n1; r += nt) {
            const int a = r / n1, c = r - a * n1;
            const double la = fmax(Ka[a * 16 + a], 0.0), lb = fmax(Kb[c * 16 + c], 0.0);
            const double dinv = 1.0 / sqrt(gp.os[o] * la * lb + gp.noise[0]);
            double acc = 0.0;                                // ((Qa (x) Qb)^T y)_r
            for (int a2 = 0; a2 < n0; ++a2) {
                double inner = 0.0;
                for (int c2 = 0; c2 < n1; ++c2) inner += Vb[c2 * 16 + c] * Y_r[((long)o * gp.N_r + a2 * n1 + c2) * gp.T];
                acc += Va[a2 * 16 + a] * inner;
            }
            dsc[r] = gp.os[o] * dinv;
            wE[r] = dinv * acc;
            m1[r] = (gp.os[o] * dinv) * (dinv * acc);
            m2[r] = (gp.os[o] * dinv) * (gp.os[o] * dinv);
        }
        if (n0 + n1 <= 14) {                                 // mode-I table (gpmpc_device.hpp: plan_tabi_*)
            __syncthreads();
            __threadfence_block();
            double* tab = L + plan_tabi_offset(n, n0, n1);
            const int nt_d = plan_tabi_doubles(n0, n1);
            // equispaced axes (the reference's linspace grids): x_q = x_0 + q h up to round-off, and the recurrence
            // exp(-il (r_0 + k h)^2 / 2) = E_0 rho^k G_k stays in range (il ((n-1) h)^2 <= 200, rollout_indep.hip)
            const double xa0 = X_r[0], xb0 = X_r[1];
            const double ha = (n0 > 1) ? (X_r[(long)((n0 - 1) * n1) * D + 0] - xa0) / (n0 - 1) : 0.0;
            const double hb = (n1 > 1) ? (X_r[(long)(n1 - 1) * D + 1] - xb0) / (n1 - 1) : 0.0;
            bool eq = !no_rec;
            for (int q = 0; q < n0; ++q) {
                const double xq = X_r[(long)(q * n1) * D + 0];
                eq = eq && fabs(xq - (xa0 + q * ha)) <= 1.5e-14 * fmax(fabs(xa0), fabs(xa0 + (n0 - 1) * ha));
            }
            for (int q = 0; q < n1; ++q) {
                const double xq = X_r[(long)q * D + 1];
                eq = eq && fabs(xq - (xb0 + q * hb)) <= 1.5e-14 * fmax(fabs(xb0), fabs(xb0 + (n1 - 1) * hb));
            }
            eq = eq && inv_l2[0] * (n0 - 1) * (n0 - 1) * ha * ha <= 200.0 && inv_l2[1] * (n1 - 1) * (n1 - 1) * hb * hb <= 200.0;
            const int ax = plan_tabi_axis(n0, n1);
            for (int e = tid; e < nt_d; e += nt) {
                double v = 0.0;
                if (e == 0) v = xa0;
                else if (e == 1) v = eq ? inv_l2[0] * ha : __builtin_nan("");
                else if (e == 2) v = xb0;
                else if (e == 3) v = inv_l2[1] * hb;
                else if (e < 4 + (n0 - 1)) { const double k = e - 3; v = exp(-0.5 * inv_l2[0] * ha * ha * k * k); }
                else if (e < 4 + (n0 - 1) + (n1 - 1)) { const double k = e - 3 - (n0 - 1); v = exp(-0.5 * inv_l2[1] * hb * hb * k * k); }
                else if (e >= ax && e < ax + n0) v = X_r[(long)((e - ax) * n1) * D + 0];
                else if (e >= ax + n0 && e < ax + n0 + n1) v = X_r[(long)(e - ax - n0) * D + 1];
                else if (e >= plan_tabi_qa(n0, n1) && e < plan_tabi_qa(n0, n1) + n0 * n0) v = Qa[e - plan_tabi_qa(n0, n1)];
                else if (e >= plan_tabi_qb(n0, n1) && e < plan_tabi_qb(n0, n1) + n1 * n1) v = Qb[e - plan_tabi_qb(n0, n1)];
                else if (e >= plan_tabi_m1(n0, n1) && e < plan_tabi_m1(n0, n1) + n0 * n1) v = m1[e - plan_tabi_m1(n0, n1)];
                else if (e >= plan_tabi_m2(n0, n1) && e < plan_tabi_m2(n0, n1) + n0 * n1) v = m2[e - plan_tabi_m2(n0, n1)];
                tab[e] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// device self test of the cross-lane primitives (run once by the test-suite / smoke on the GPU box)
// ---------------------------------------------------------------------------------------------------------------
__global__ void selftest_kernel(int* out) {
    const int lane = threadIdx.x;
    int bad = 0;
    const double v = 1.0 + 0.25 * lane + 1e-9 * lane * lane;
    double ref = 0.0;
    for (int l = 0; l < 64; ++l) ref += 1.0 + 0.25 * l + 1e-9 * l * l;
    const double s1 = wave_sum(v), s2 = wave_sum_shfl(v);
    if (fabs(s1 - ref) > 1e-9 * fabs(ref)) bad |= 1;
    if (fabs(s2 - ref) > 1e-9 * fabs(ref)) bad |= 2;
    {   // four sums through the lane-swap tree: quantities with different lane patterns
        const double q0 = v, q1 = 2.0 - 0.125 * lane, q2 = (lane & 1) ? 3.0 : -1.5, q3 = 1e-3 * lane * lane;
        double r0, r1, r2, r3;
        wave_sum4(q0, q1, q2, q3, r0, r1, r2, r3);
        const double e1 = wave_sum_shfl(q1), e2 = wave_sum_shfl(q2), e3 = wave_sum_shfl(q3);
        if (fabs(r0 - ref) > 1e-9 * fabs(ref) || fabs(r1 - e1) > 1e-9 * fabs(e1) || fabs(r2 - e2) > 1e-9 * fabs(e2) ||
            fabs(r3 - e3) > 1e-9 * fabs(e3))
            bad |= 64;
    }
    for (int l = 0; l < 64; l += 7) {
        const double r = readlane_f64(v, l);
        if (r != 1.0 + 0.25 * l + 1e-9 * l * l) bad |= 4;
    }
    double S[3][3] = {{4.0, 2.0, 0.6}, {2.0, 5.0, 1.0}, {0.6, 1.0, 3.0}}, L[3][3];
    if (!chol_small<3>(S, L)) bad |= 8;
    double err = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a += L[i][k] * L[j][k];
            err += fabs(a - S[i][j]);
        }
    if (err > 1e-12) bad |= 16;
    double Sn[3][3] = {{1.0, 2.0, 0.0}, {2.0, 1.0, 0.0}, {0.0, 0.0, 1.0}}, R[3][3];
    const int inf = root_small<3>(Sn, 1e-6, R);
    if (!(inf & GPMPC_INFO_ROOT_FAIL)) bad |= 32;
    {   // nine sums in lockstep against the single-quantity reductions
        const double qa[4] = {v, 2.0 - 0.125 * lane, (lane & 1) ? 3.0 : -1.5, 1e-3 * lane * lane};
        const double qb[4] = {0.5 * lane, 1.0 / (1 + lane), -v, (lane & 3) * 0.25};
        const double qc = 7.0 - 0.01 * lane;
        double ra[4], rb[4], rc;
        wave_sum9(qa, qb, qc, ra, rb, rc);
        for (int i = 0; i < 4; ++i) {
            if (ra[i] != wave_sum(qa[i]) && fabs(ra[i] - wave_sum_shfl(qa[i])) > 1e-12 * fabs(ra[i])) bad |= 512;
            if (fabs(rb[i] - wave_sum_shfl(qb[i])) > 1e-12 * (1.0 + fabs(rb[i]))) bad |= 512;
        }
        if (rc != wave_sum(qc)) bad |= 512;
    }
    {   // lockstep exponentials against the library's exp over the argument range of the kernels
        const double xs[3] = {-1e-3 * lane * lane, -0.37 * lane - 1e-7, -11.0 * lane - 0.123};
        double es[3];
        exp3_neg(xs, es);
        for (int i = 0; i < 3; ++i) {
            const double ref_e = exp(xs[i]);
            if (!(fabs(es[i] - ref_e) <= 4e-16 * ref_e)) bad |= 128;
        }
        const double xz[3] = {-800.0, -1e6, 0.0};
        exp3_neg(xz, es);
        if (es[0] != 0.0 || es[1] != 0.0 || es[2] != 1.0) bad |= 256;
    }
    const unsigned long long any = __ballot(bad != 0);
    if (lane == 0) out[0] = (any != 0ull) ? (bad | 0x1000) : 0;
    if (bad) atomicOr(out + 1, bad);
}

}  // namespace gpmpc

using namespace gpmpc;

extern "C" {

int gpmpc_abi_version(void) { return GPMPC_ABI_VERSION; }

const char* gpmpc_last_error_string(void) { return last_error().c_str(); }

int gpmpc_device_info(int dev, char* name, int cap, int* cu_count, int* lds_bytes) {
    hipDeviceProp_t prop;
    GPMPC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (name && cap > 0) {
        std::snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    return GPMPC_OK;
}

int gpmpc_selftest(void* stream) {
    int* d = nullptr;
    int h[2] = {-1, -1};
    GPMPC_HIP_CHECK(hipMalloc(&d, 2 * sizeof(int)));
    GPMPC_HIP_CHECK(hipMemsetAsync(d, 0, 2 * sizeof(int), (hipStream_t)stream));
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
    GPMPC_HIP_CHECK(hipGetLastError());
    GPMPC_HIP_CHECK(hipMemcpyAsync(h, d, 2 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GPMPC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    GPMPC_HIP_CHECK(hipFree(d));
    if (h[0] != 0 || h[1] != 0) {
        char buf[96];
        std::snprintf(buf, sizeof(buf), "device self test failed: mask 0x%x / 0x%x", h[0], h[1]);
        return fail(GPMPC_E_HIP, buf);
    }
    return GPMPC_OK;
}

size_t gpmpc_plan_bytes(const gpmpc_gp_desc_t* gp) {
    if (check_gp(gp) != GPMPC_OK) return 0;
    const int n_r = observed_real_slots(gp);
    return align_up((size_t)gp->g_ny * plan_doubles_per_output(n_r, gp->grid_n0, gp->grid_n1) * sizeof(double), 256);
}

int gpmpc_plan_build(const gpmpc_gp_desc_t* gp, const double* X_r, const double* Y_r, void* plan, int32_t* info,
                     void* stream) {
    if (int rc = check_gp(gp)) return rc;
    if (!X_r || !Y_r || !plan || !info) return fail(GPMPC_E_ARG, "gpmpc_plan_build: NULL pointer");
    GpParams p = make_gp_params(gp);
    const size_t lds = (size_t)p.n_r * p.n_r * sizeof(double);
    // 160 KiB per workgroup minus the kernel's static LDS (the four 16 x 16 Jacobi scratch tiles = 8 KiB, flags)
    constexpr size_t PLAN_STATIC_LDS = 4 * 16 * 16 * sizeof(double) + 256;
    if (lds + PLAN_STATIC_LDS > 160 * 1024)
        return fail(GPMPC_E_UNSUPPORTED, "plan: n_r too large for the LDS-resident factorisation (n_r^2 * 8 B + 8.25 KiB > 160 KiB)");
    if (gp->D != 2) return fail(GPMPC_E_UNSUPPORTED, "only D = 2 is instantiated");
    auto kern = plan_kernel<2>;
    GPMPC_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // GPMPC_DISABLE_EXP_RECURRENCE=1: the mode-I table is marked "axes not equispaced" (tests: the direct-exponential path)
    const char* er = std::getenv("GPMPC_DISABLE_EXP_RECURRENCE");
    hipLaunchKernelGGL(kern, dim3(gp->g_ny), dim3(256), lds, (hipStream_t)stream, p, X_r, Y_r, (double*)plan,
                       (int*)info, (er && er[0] == '1') ? 1 : 0);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // extern "C"
