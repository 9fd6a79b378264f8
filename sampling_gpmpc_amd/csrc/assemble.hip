// gpmpc_assemble_jacobians / gpmpc_pack_plin: the elementwise tail of the SQP linearisation (gfx950).
// HBM-bound streaming kernels: one thread per output row, contiguous stores.
#include "gpmpc_host.hpp"

namespace gpmpc {

// Replaces reference src/agent.py:532-557 for the two environments:
//   y_full[row] = [f | df/dx | df/du](row) + sum_o B_d[row][o] * scatter_{pad_g}(transform(y[o]))
template <int T>
__global__ __launch_bounds__(256) void jacobians_kernel(GpParams gp, EnvParams env, long Ns, int H,
                                                        const double* __restrict__ xu, const double* __restrict__ y,
                                                        double* __restrict__ gp_val, double* __restrict__ y_grad,
                                                        double* __restrict__ u_grad) {
    constexpr int I1 = (T == 1) ? 0 : 1, I2 = (T == 1) ? 0 : 2;   // value-only model: one column feeds all pads
    const int nx = env.nx, nu = env.nu, W = 1 + nx + nu;
    const long total = Ns * nx * H;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e % H);
        const int row = (int)((e / H) % nx);
        const long s = e / ((long)H * nx);
        const double* xr = xu + ((s * nx + 0) * H + h) * (nx + nu);      // replica 0 of the state row
        double out[1 + GPMPC_MAX_NX + GPMPC_MAX_NU];
        for (int c = 0; c < W; ++c) out[c] = 0.0;
        if (env.env_id == GPMPC_ENV_PENDULUM1D) {
            // known part: theta+ = theta + omega dt ; omega+ = omega          (pendulum1D.py:137-188)
            if (row == 0) {
                out[0] = xr[0] + xr[1] * env.dt;
                out[1] = 1.0;
                out[2] = env.dt;
            } else {
                out[0] = xr[1];
                out[2] = 1.0;
                const double* yo = y + ((s * gp.g_ny + 0) * H + h) * T;  // B_d = [0,1]^T, pad_g = [0,1,3]
                out[0] += yo[0];
                out[1] += yo[I1];                                        // T == 1: the value is broadcast (quirk K)
                out[3] += yo[I2];
            }
        } else {
            // known part: identity on (X,Y,phi,v), dv+/da = dt                (car_model_residual.py:101-161)
            out[0] = xr[row] + ((row == 3) ? xr[5] * env.dt : 0.0);
            out[1 + row] = 1.0;
            if (row == 3) out[6] = env.dt;
            if (row < 3) {
                // transform_sensitivity (211-224): [g, dg/dphi, dg/ddelta] -> [v g, v dg/dphi, g, v dg/ddelta]
                // scattered to pad_g = [0,3,4,5]; B_d = I_{4x3}
                const double* yo = y + ((s * gp.g_ny + row) * H + h) * T;
                const double v = xr[3];
                const double y0 = yo[0], y1 = yo[I1], y2 = yo[I2];
                out[0] += v * y0;
                out[3] += v * y1;
                out[4] += y0;
                out[5] += v * y2;
            }
        }
        gp_val[e] = out[0];
        for (int c = 0; c < nx; ++c) y_grad[e * nx + c] = out[1 + c];
        for (int c = 0; c < nu; ++c) u_grad[e * nu + c] = out[1 + nx + c];
    }
}

// Replaces the O(Ns^2) host loop of reference src/solver.py:98-131.
__global__ __launch_bounds__(256) void plin_kernel(int nx, int nu, long Ns, int H, const double* __restrict__ y_grad,
                                                   const double* __restrict__ u_grad, const double* __restrict__ gp_val,
                                                   const double* __restrict__ x_h, const double* __restrict__ u_h,
                                                   const double* __restrict__ xg, const double* __restrict__ w,
                                                   const double* __restrict__ tilde_eps, const double* __restrict__ Kfb,
                                                   double* __restrict__ p_lin) {
    const long per = (long)nx * nx + (long)nx * nu + 2L * nx;
    const long tail = nu + 2 + (nx + nu + 1);
    const long len = Ns * per + tail;
    const long total = len * H;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e / len);
        const long k = e - (long)h * len;
        double v;
        if (k < Ns * per) {
            const long i = k / per;
            const int r = (int)(k - i * per);
            if (r < nx * nx) {
                const int a = r / nx, b = r - a * nx;
                v = y_grad[((i * nx + a) * H + h) * nx + b];
                if (Kfb) {                                        // feedback: A_i = y_grad + u_grad K (reference src/solver.py:90)
                    const double* ug = u_grad + ((i * nx + a) * H + h) * nu;
                    double acc = 0.0;
                    for (int j = 0; j < nu; ++j) acc += ug[j] * Kfb[j * nx + b];
                    v += acc;
                }
            } else if (r < nx * nx + nx * nu) {
                const int rr = r - nx * nx;
                const int a = rr / nu, b = rr - a * nu;
                v = u_grad[((i * nx + a) * H + h) * nu + b];
            } else if (r < nx * nx + nx * nu + nx) {
                v = x_h[(long)h * Ns * nx + i * nx + (r - nx * nx - nx * nu)];
            } else {
                v = gp_val[(i * nx + (r - nx * nx - nx * nu - nx)) * H + h];
            }
        } else {
            const int r = (int)(k - Ns * per);
            if (r < nu) v = u_h[(long)h * nu + r];
            else if (r == nu) v = xg[h];
            else if (r == nu + 1) v = w[h];
            else v = tilde_eps[(long)h * (nx + nu + 1) + (r - nu - 2)];
        }
        p_lin[e] = v;
    }
}

// batch_x_hat from the solver's iterate in ONE launch (reference src/agent.py:480-527: a reshape, a broadcast of the inputs and
// an nx-fold replication of the state row - four torch ops and their temporaries):
//   xu[s][r][h][0..nx) = x_h[h][s nx + c],  xu[s][r][h][nx..nx+nu) = u_h[h][(s)][j]   for every replica r < nx
__global__ __launch_bounds__(256) void x_hat_kernel(int nx, int nu, long Ns, int H, const double* __restrict__ x_h,
                                                    const double* __restrict__ u_h, int u_per_sample, double* __restrict__ xu) {
    const int W = nx + nu;
    const long total = Ns * nx * H * W;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % W);
        const int h = (int)((e / W) % H);
        const long s = e / ((long)W * H * nx);
        xu[e] = (c < nx) ? x_h[(long)h * Ns * nx + s * nx + c]
                         : (u_per_sample ? u_h[((long)h * Ns + s) * nu + (c - nx)] : u_h[(long)h * nu + (c - nx)]);
    }
}

// gpmpc_assemble_jacobians + gpmpc_pack_plin_fb in ONE launch: thread (s, row, h) forms its row of [f | df/dx | df/du] as
// jacobians_kernel does, writes it to the three arrays AND to its places in stage h's parameter vector (row `row` of A_i and
// B_i, x_hat_i[row], f_i[row]); thread (0, 0, h) adds the stage's tail.  The reference's solver consumes only p_lin
// (src/solver.py:98-131): with this form the three arrays never have to leave the device.
template <int T>
__global__ __launch_bounds__(256) void jacobians_plin_kernel(GpParams gp, EnvParams env, long Ns, int H,
                                                             const double* __restrict__ xu, const double* __restrict__ y,
                                                             double* __restrict__ gp_val, double* __restrict__ y_grad,
                                                             double* __restrict__ u_grad, const double* __restrict__ u_h,
                                                             const double* __restrict__ xg, const double* __restrict__ w,
                                                             const double* __restrict__ tilde_eps, const double* __restrict__ Kfb,
                                                             double* __restrict__ p_lin) {
    constexpr int I1 = (T == 1) ? 0 : 1, I2 = (T == 1) ? 0 : 2;
    const int nx = env.nx, nu = env.nu, W = 1 + nx + nu;
    const long per = (long)nx * nx + (long)nx * nu + 2L * nx;
    const long len = Ns * per + nu + 2 + (nx + nu + 1);
    const long total = Ns * nx * H;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e % H);
        const int row = (int)((e / H) % nx);
        const long s = e / ((long)H * nx);
        const double* xr = xu + ((s * nx + 0) * H + h) * (nx + nu);
        double out[1 + GPMPC_MAX_NX + GPMPC_MAX_NU];
        for (int c = 0; c < W; ++c) out[c] = 0.0;
        if (env.env_id == GPMPC_ENV_PENDULUM1D) {
            if (row == 0) {
                out[0] = xr[0] + xr[1] * env.dt;
                out[1] = 1.0;
                out[2] = env.dt;
            } else {
                out[0] = xr[1];
                out[2] = 1.0;
                const double* yo = y + ((s * gp.g_ny + 0) * H + h) * T;
                out[0] += yo[0];
                out[1] += yo[I1];
                out[3] += yo[I2];
            }
        } else {
            out[0] = xr[row] + ((row == 3) ? xr[5] * env.dt : 0.0);
            out[1 + row] = 1.0;
            if (row == 3) out[6] = env.dt;
            if (row < 3) {
                const double* yo = y + ((s * gp.g_ny + row) * H + h) * T;
                const double v = xr[3];
                const double y0 = yo[0], y1 = yo[I1], y2 = yo[I2];
                out[0] += v * y0;
                out[3] += v * y1;
                out[4] += y0;
                out[5] += v * y2;
            }
        }
        gp_val[e] = out[0];
        for (int c = 0; c < nx; ++c) y_grad[e * nx + c] = out[1 + c];
        for (int c = 0; c < nu; ++c) u_grad[e * nu + c] = out[1 + nx + c];
        double* ps = p_lin + (long)h * len + s * per;
        for (int b = 0; b < nx; ++b) {                            // A_i[row][b] (+ u_grad K under feedback, src/solver.py:90)
            double v = out[1 + b];
            if (Kfb) {
                double acc = 0.0;
                for (int j = 0; j < nu; ++j) acc += out[1 + nx + j] * Kfb[j * nx + b];
                v += acc;
            }
            ps[row * nx + b] = v;
        }
        for (int b = 0; b < nu; ++b) ps[nx * nx + row * nu + b] = out[1 + nx + b];
        ps[nx * nx + nx * nu + row] = xr[row];                    // x_hat_i[row] = x_h[h][s nx + row]
        ps[nx * nx + nx * nu + nx + row] = out[0];                // f_i[row]
        if (s == 0 && row == 0) {
            double* pt = p_lin + (long)h * len + Ns * per;
            for (int r = 0; r < nu; ++r) pt[r] = u_h[(long)h * nu + r];
            pt[nu] = xg[h];
            pt[nu + 1] = w[h];
            for (int r = 0; r < nx + nu + 1; ++r) pt[nu + 2 + r] = tilde_eps[(long)h * (nx + nu + 1) + r];
        }
    }
}

static unsigned stream_grid(long total) {
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;            // ~8 workgroups per CU, grid-stride for the rest
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace gpmpc

using namespace gpmpc;

extern "C" {

int gpmpc_assemble_jacobians(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int64_t Ns, int32_t H,
                             const double* xu, const double* y, double* gp_val, double* y_grad, double* u_grad,
                             void* stream) {
    if (int rc = check_gp(gp)) return rc;
    if (int rc = check_env(gp, env)) return rc;
    if (!xu || !y || !gp_val || !y_grad || !u_grad) return fail(GPMPC_E_ARG, "gpmpc_assemble_jacobians: NULL pointer");
    if (Ns < 1 || H < 1) return fail(GPMPC_E_ARG, "gpmpc_assemble_jacobians: bad sizes");
    GpParams g = make_gp_params(gp);
    EnvParams e = make_env_params(env);
    const long total = Ns * env->nx * H;
    hipStream_t st = (hipStream_t)stream;
    if (gp->T == 1)
        hipLaunchKernelGGL(jacobians_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, st, g, e, (long)Ns, H, xu, y,
                           gp_val, y_grad, u_grad);
    else if (gp->T == 3)
        hipLaunchKernelGGL(jacobians_kernel<3>, dim3(stream_grid(total)), dim3(256), 0, st, g, e, (long)Ns, H, xu, y,
                           gp_val, y_grad, u_grad);
    else
        return fail(GPMPC_E_UNSUPPORTED, "jacobians: only T = 1 and T = 3 are instantiated");
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

int gpmpc_build_x_hat(int32_t nx, int32_t nu, int64_t Ns, int32_t H, const double* x_h, const double* u_h,
                      int32_t u_per_sample, double* xu, void* stream) {
    if (!x_h || !u_h || !xu) return fail(GPMPC_E_ARG, "gpmpc_build_x_hat: NULL pointer");
    if (nx < 1 || nu < 1 || Ns < 1 || H < 1) return fail(GPMPC_E_ARG, "gpmpc_build_x_hat: bad sizes");
    const long total = (long)Ns * nx * H * (nx + nu);
    hipLaunchKernelGGL(x_hat_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, nx, nu, (long)Ns, H, x_h, u_h,
                       (int)u_per_sample, xu);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

int gpmpc_assemble_jacobians_plin(const gpmpc_gp_desc_t* gp, const gpmpc_env_desc_t* env, int64_t Ns, int32_t H,
                                  const double* xu, const double* y, double* gp_val, double* y_grad, double* u_grad,
                                  const double* u_h, const double* xg, const double* w, const double* tilde_eps, const double* K,
                                  double* p_lin, void* stream) {
    if (int rc = check_gp(gp)) return rc;
    if (int rc = check_env(gp, env)) return rc;
    if (!xu || !y || !gp_val || !y_grad || !u_grad || !u_h || !xg || !w || !tilde_eps || !p_lin)
        return fail(GPMPC_E_ARG, "gpmpc_assemble_jacobians_plin: NULL pointer");
    if (Ns < 1 || H < 1) return fail(GPMPC_E_ARG, "gpmpc_assemble_jacobians_plin: bad sizes");
    GpParams g = make_gp_params(gp);
    EnvParams e = make_env_params(env);
    const long total = Ns * env->nx * H;
    hipStream_t st = (hipStream_t)stream;
    if (gp->T == 1)
        hipLaunchKernelGGL(jacobians_plin_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, st, g, e, (long)Ns, H, xu, y, gp_val,
                           y_grad, u_grad, u_h, xg, w, tilde_eps, K, p_lin);
    else if (gp->T == 3)
        hipLaunchKernelGGL(jacobians_plin_kernel<3>, dim3(stream_grid(total)), dim3(256), 0, st, g, e, (long)Ns, H, xu, y, gp_val,
                           y_grad, u_grad, u_h, xg, w, tilde_eps, K, p_lin);
    else
        return fail(GPMPC_E_UNSUPPORTED, "jacobians: only T = 1 and T = 3 are instantiated");
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

int64_t gpmpc_plin_len(int32_t nx, int32_t nu, int64_t Ns) {
    return Ns * ((int64_t)nx * nx + (int64_t)nx * nu + 2 * nx) + nu + 2 + (nx + nu + 1);
}

int gpmpc_pack_plin(int32_t nx, int32_t nu, int64_t Ns, int32_t H, const double* y_grad, const double* u_grad,
                    const double* gp_val, const double* x_h, const double* u_h, const double* xg, const double* w,
                    const double* tilde_eps, double* p_lin, void* stream) {
    if (!y_grad || !u_grad || !gp_val || !x_h || !u_h || !xg || !w || !tilde_eps || !p_lin)
        return fail(GPMPC_E_ARG, "gpmpc_pack_plin: NULL pointer");
    if (nx < 1 || nu < 1 || Ns < 1 || H < 1) return fail(GPMPC_E_ARG, "gpmpc_pack_plin: bad sizes");
    const long total = gpmpc_plin_len(nx, nu, Ns) * H;
    hipLaunchKernelGGL(plin_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, nx, nu, (long)Ns, H,
                       y_grad, u_grad, gp_val, x_h, u_h, xg, w, tilde_eps, (const double*)nullptr, p_lin);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

int gpmpc_pack_plin_fb(int32_t nx, int32_t nu, int64_t Ns, int32_t H, const double* y_grad, const double* u_grad,
                       const double* gp_val, const double* x_h, const double* u_h, const double* xg, const double* w,
                       const double* tilde_eps, const double* K, double* p_lin, void* stream) {
    if (!y_grad || !u_grad || !gp_val || !x_h || !u_h || !xg || !w || !tilde_eps || !p_lin)
        return fail(GPMPC_E_ARG, "gpmpc_pack_plin_fb: NULL pointer");
    if (nx < 1 || nu < 1 || Ns < 1 || H < 1) return fail(GPMPC_E_ARG, "gpmpc_pack_plin_fb: bad sizes");
    const long total = gpmpc_plin_len(nx, nu, Ns) * H;
    hipLaunchKernelGGL(plin_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, nx, nu, (long)Ns, H,
                       y_grad, u_grad, gp_val, x_h, u_h, xg, w, tilde_eps, K, p_lin);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

// bitwise OR over n int32 words (the per-chain info words of a launch) into out[0], which the caller has zeroed - or keeps
// accumulating into
__global__ __launch_bounds__(256) void or_reduce_kernel(const int* __restrict__ v, long n, int* __restrict__ out) {
    int acc = 0;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) acc |= v[e];
    for (int m = 32; m >= 1; m >>= 1) acc |= __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicOr(out, acc);
}

int gpmpc_or_reduce_words(const int32_t* v, int64_t n, int32_t* out, void* stream) {
    if (!out || (n > 0 && !v) || n < 0) return fail(GPMPC_E_ARG, "gpmpc_or_reduce_words: bad arguments");
    if (n == 0) return GPMPC_OK;
    long g = (n + 255) / 256;
    if (g > 256) g = 256;
    hipLaunchKernelGGL(or_reduce_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const int*)v, (long)n, (int*)out);
    GPMPC_HIP_CHECK(hipGetLastError());
    return GPMPC_OK;
}

}  // extern "C"
