// Device-side building blocks shared by the gfx950 kernels of libgpmpc_hip.so.
// Wave = 64 lanes everywhere (CDNA4); nothing here is written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpmpc_hip.h"

namespace gpmpc {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------------------------------
// kernel-argument copies of the C-ABI descriptors (plain data, passed by value)
// ---------------------------------------------------------------------------------------------------------------
struct GpParams {
    int g_ny, D, T, N_r, real_has_grad, n_r;          // n_r = observed real label slots
    int grid_n0, grid_n1;                             // tensor-product training grid (0/0 = unstructured)
    double inv_l2[GPMPC_MAX_NY][GPMPC_MAX_D];         // 1 / ell^2
    double os[GPMPC_MAX_NY];                          // outputscale
    double noise[GPMPC_MAX_T];
    double jitter, var_floor;
    long plan_stride;                                 // doubles per output inside the plan buffer
};

struct EnvParams {
    int env_id, nx, nu, use_feedback;
    double dt, p0, p1;
    double K[GPMPC_MAX_NU][GPMPC_MAX_NX];
    double x_goal[GPMPC_MAX_NX];
};

// plan buffer layout per output o (doubles): [ L (n_r*n_r, row-major) | LinvT (n_r*n_r, LinvT[j*n_r+i]=Linv[i][j])
//                                            | w (n_r) = L^-1 y | alpha (n_r) = L^-T w ]
//                                            | grid root (value-only real labels on a tensor grid n0 x n1, else absent):
//                                              Qa (n0*n0, row-major, column a = eigenvector a of the axis-0 kernel matrix)
//                                              | Qb (n1*n1) | dsc (N_r) | wE (N_r) | m1 (N_r) = dsc wE | m2 (N_r) = dsc^2 ]
// The grid root: K_rr + s2 I = os Ka (x) Kb + s2 I = (Qa (x) Qb) D (Qa (x) Qb)^T, D = os la_a lb_c + s2, so
// W = D^-1/2 (Qa (x) Qb)^T satisfies W^T W = (K_rr + s2 I)^-1 and can stand in for L_rr^-1 everywhere (the Schur
// complement, the posterior mean and covariance only see W^T W).  W k_r for a separable kernel row costs n0 + n1 pivots
// instead of N_r.  dsc[r] = os / sqrt(D_r) (the outputscale of k_r folded in), wE = W y_r.
// Mode-I table (rollout_indep.hip), appended to the grid root: the tables the grid-root mode-I kernel keeps in
// registers (16 entries per VGPR pair), packed in units of 8 doubles, 64-byte aligned inside the plan:
//   [ rec (16): x_first(axis 0), il0*h0, x_first(axis 1), il1*h1, then G0_k = exp(-il0 (k h0)^2 / 2), k = 1..n0-1, and
//               G1_k, k = 1..n1-1 - the constants of the equispaced-axis recurrence for the kernel factors
//               (il0*h0 = NaN: the axes are not equispaced / the recurrence is out of range, use the axis points)
//     | axis (16): axis-0 points (n0), axis-1 points (n1)
//     | Qa flat (n0*n0) pad 8 | Qb flat (n1*n1) pad 8 | m1 flat (n0*n1) pad 8 | m2 flat (n0*n1) pad 8 ],
//   the unit count rounded up to even
__host__ __device__ constexpr int plan_tabi_c8(int n) { return (n + 7) / 8; }
__host__ __device__ constexpr int plan_tabi_axis(int, int) { return 16; }
__host__ __device__ constexpr int plan_tabi_qa(int, int) { return 32; }
__host__ __device__ constexpr int plan_tabi_qb(int n0, int n1) { return plan_tabi_qa(n0, n1) + 8 * plan_tabi_c8(n0 * n0); }
__host__ __device__ constexpr int plan_tabi_m1(int n0, int n1) { return plan_tabi_qb(n0, n1) + 8 * plan_tabi_c8(n1 * n1); }
__host__ __device__ constexpr int plan_tabi_m2(int n0, int n1) { return plan_tabi_m1(n0, n1) + 8 * plan_tabi_c8(n0 * n1); }
__host__ __device__ constexpr int plan_tabi_doubles(int n0, int n1) {
    return 16 * ((plan_tabi_m2(n0, n1) + 8 * plan_tabi_c8(n0 * n1) + 15) / 16);
}
__host__ __device__ inline long plan_tabi_offset(int n_r, int n0, int n1) {       // doubles from the output's plan start
    return (2L * n_r * n_r + 2L * n_r + (long)n0 * n0 + (long)n1 * n1 + 4L * n0 * n1 + 7) & ~7L;
}
__host__ __device__ inline long plan_doubles_per_output(int n_r, int n0 = 0, int n1 = 0) {
    if (n0 > 0 && n1 > 0 && n0 + n1 <= 14) return plan_tabi_offset(n_r, n0, n1) + plan_tabi_doubles(n0, n1);
    const long grid = (n0 > 0 && n1 > 0) ? ((long)n0 * n0 + (long)n1 * n1 + 4L * n0 * n1 + 1) & ~1L : 0;
    return (2L * n_r * n_r + 2L * n_r + grid + 7) & ~7L;
}
__device__ inline const double* plan_L(const double* plan, const GpParams& gp, int o) { return plan + o * gp.plan_stride; }
__device__ inline const double* plan_LinvT(const double* plan, const GpParams& gp, int o) {
    return plan + o * gp.plan_stride + (long)gp.n_r * gp.n_r;
}
__device__ inline const double* plan_w(const double* plan, const GpParams& gp, int o) {
    return plan + o * gp.plan_stride + 2L * gp.n_r * gp.n_r;
}
__device__ inline const double* plan_alpha(const double* plan, const GpParams& gp, int o) {
    return plan + o * gp.plan_stride + 2L * gp.n_r * gp.n_r + gp.n_r;
}
__device__ inline const double* plan_grid_Qa(const double* plan, const GpParams& gp, int o) {
    return plan + o * gp.plan_stride + 2L * gp.n_r * gp.n_r + 2L * gp.n_r;
}
__device__ inline const double* plan_grid_Qb(const double* plan, const GpParams& gp, int o) {
    return plan_grid_Qa(plan, gp, o) + (long)gp.grid_n0 * gp.grid_n0;
}
__device__ inline const double* plan_grid_dsc(const double* plan, const GpParams& gp, int o) {
    return plan_grid_Qb(plan, gp, o) + (long)gp.grid_n1 * gp.grid_n1;
}
__device__ inline const double* plan_grid_w(const double* plan, const GpParams& gp, int o) {
    return plan_grid_dsc(plan, gp, o) + (long)gp.grid_n0 * gp.grid_n1;
}
// m1 / m2: the mean and variance weights of the mode-I kernel (rollout_indep.hip): mu = sum_r m1_r A_a B_c,
// k_r^T (K_rr + s2 I)^-1 k_r = sum_r m2_r A_a^2 B_c^2 with A = Qa^T ea, B = Qb^T eb
__device__ inline const double* plan_grid_m1(const double* plan, const GpParams& gp, int o) {
    return plan_grid_w(plan, gp, o) + (long)gp.grid_n0 * gp.grid_n1;
}
__device__ inline const double* plan_grid_m2(const double* plan, const GpParams& gp, int o) {
    return plan_grid_m1(plan, gp, o) + (long)gp.grid_n0 * gp.grid_n1;
}
__device__ inline const double* plan_grid_tabi(const double* plan, const GpParams& gp, int o) {
    return plan + o * gp.plan_stride + plan_tabi_offset(gp.n_r, gp.grid_n0, gp.grid_n1);
}
__host__ __device__ inline bool plan_has_grid_root(int grid_n0, int grid_n1, int real_has_grad) {
    return grid_n0 > 0 && grid_n1 > 0 && grid_n0 <= 16 && grid_n1 <= 16 && !real_has_grad;
}

// ---------------------------------------------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    // v_readlane_b32 x2 -> SGPR pair: the broadcast value feeds v_fma_f64 as a scalar operand
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    // lanes whose source is out of range / masked off receive +0.0 (old = 0, bound_ctrl = 0)
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes of a wave; the result is returned (uniform) to every lane.
// DPP ladder: row_shr 1,2,3 -> row_shr 4 -> row_shr 8 -> row_bcast15 -> row_bcast31, total lands in lane 63.
__device__ __forceinline__ double wave_sum(double v) {
    double t = v + dpp_f64<0x111, 0xf, 0xf>(v);          // row_shr:1
    t += dpp_f64<0x112, 0xf, 0xf>(v);                    // row_shr:2
    t += dpp_f64<0x113, 0xf, 0xf>(v);                    // row_shr:3
    t += dpp_f64<0x114, 0xf, 0xe>(t);                    // row_shr:4  bank_mask 0xe
    t += dpp_f64<0x118, 0xf, 0xc>(t);                    // row_shr:8  bank_mask 0xc
    t += dpp_f64<0x142, 0xa, 0xf>(t);                    // row_bcast:15 row_mask 0xa
    t += dpp_f64<0x143, 0xc, 0xf>(t);                    // row_bcast:31 row_mask 0xc
    return readlane_f64(t, 63);
}

// Four wave sums at once.  A wave is four DPP rows of 16 lanes; the gfx950 lane swaps fold the rows of two registers
// into one (v_permlane32_swap a, b: a's rows 2,3 <-> b's rows 0,1; v_permlane16_swap a, b: a's odd rows <-> b's even
// rows), so after two levels ONE register carries the 16-lane partial sums of all four quantities, one per DPP row,
// and a single row_shr ladder finishes them: 6 swaps + 3 adds + 5 DPP steps + 8 v_readlane instead of 4 x (7 DPP steps
// + 2 v_readlane).
__device__ __forceinline__ void lane_swap32_f64(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void lane_swap16_f64(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void wave_sum4(double a, double b, double c, double d, double& sa, double& sb, double& sc,
                                          double& sd) {
    lane_swap32_f64(a, b);                 // a = [a.r0 a.r1 b.r0 b.r1], b = [a.r2 a.r3 b.r2 b.r3]
    lane_swap32_f64(c, d);
    double ab = a + b;                     // rows 0,1: partial sums of a; rows 2,3: of b
    double cd = c + d;
    lane_swap16_f64(ab, cd);               // ab = [ab.r0 cd.r0 ab.r2 cd.r2], cd = [ab.r1 cd.r1 ab.r3 cd.r3]
    const double v = ab + cd;              // row 0: a, row 1: c, row 2: b, row 3: d  (16-lane partial sums)
    double t = v + dpp_f64<0x111, 0xf, 0xf>(v);          // row_shr:1
    t += dpp_f64<0x112, 0xf, 0xf>(v);                    // row_shr:2
    t += dpp_f64<0x113, 0xf, 0xf>(v);                    // row_shr:3
    t += dpp_f64<0x114, 0xf, 0xe>(t);                    // row_shr:4  bank_mask 0xe
    t += dpp_f64<0x118, 0xf, 0xc>(t);                    // row_shr:8  bank_mask 0xc  -> lane 15 of each row
    sa = readlane_f64(t, 15);
    sc = readlane_f64(t, 31);
    sb = readlane_f64(t, 47);
    sd = readlane_f64(t, 63);
}

// Nine wave sums (the per-step reductions of the tuned rollout: three means, six covariance terms) as two lane-swap
// trees of four and one DPP ladder advancing in LOCKSTEP (rows pinned with sched_barrier): each tree is a chain of ~13
// dependent steps, and a single in-order wave does not overlap them by itself.  Per quantity the operations and their
// order are those of wave_sum4 / wave_sum => bit-identical sums.
__device__ __forceinline__ void wave_sum9(const double (&a)[4], const double (&b)[4], double c, double (&sa)[4], double (&sb)[4],
                                          double& sc) {
    double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    __builtin_amdgcn_sched_barrier(0);
    lane_swap32_f64(a0, a1);
    lane_swap32_f64(a2, a3);
    lane_swap32_f64(b0, b1);
    lane_swap32_f64(b2, b3);
    const double c1 = dpp_f64<0x111, 0xf, 0xf>(c), c2 = dpp_f64<0x112, 0xf, 0xf>(c), c3 = dpp_f64<0x113, 0xf, 0xf>(c);
    __builtin_amdgcn_sched_barrier(0);
    double aab = a0 + a1, acd = a2 + a3, bab = b0 + b1, bcd = b2 + b3;
    double tc = c + c1;
    __builtin_amdgcn_sched_barrier(0);
    lane_swap16_f64(aab, acd);
    lane_swap16_f64(bab, bcd);
    tc += c2;
    __builtin_amdgcn_sched_barrier(0);
    const double va = aab + acd, vb = bab + bcd;          // DPP row 0: first, row 1: third, row 2: second, row 3: fourth
    tc += c3;
    __builtin_amdgcn_sched_barrier(0);
    const double va1 = dpp_f64<0x111, 0xf, 0xf>(va), va2 = dpp_f64<0x112, 0xf, 0xf>(va), va3 = dpp_f64<0x113, 0xf, 0xf>(va);
    const double vb1 = dpp_f64<0x111, 0xf, 0xf>(vb), vb2 = dpp_f64<0x112, 0xf, 0xf>(vb), vb3 = dpp_f64<0x113, 0xf, 0xf>(vb);
    double yc = dpp_f64<0x114, 0xf, 0xe>(tc);
    __builtin_amdgcn_sched_barrier(0);
    double ta = va + va1, tb = vb + vb1;
    tc += yc;
    __builtin_amdgcn_sched_barrier(0);
    ta += va2, tb += vb2;
    yc = dpp_f64<0x118, 0xf, 0xc>(tc);
    __builtin_amdgcn_sched_barrier(0);
    ta += va3, tb += vb3;
    tc += yc;
    __builtin_amdgcn_sched_barrier(0);
    double ya = dpp_f64<0x114, 0xf, 0xe>(ta), yb = dpp_f64<0x114, 0xf, 0xe>(tb);
    yc = dpp_f64<0x142, 0xa, 0xf>(tc);                    // row_bcast:15
    __builtin_amdgcn_sched_barrier(0);
    ta += ya, tb += yb, tc += yc;
    __builtin_amdgcn_sched_barrier(0);
    ya = dpp_f64<0x118, 0xf, 0xc>(ta), yb = dpp_f64<0x118, 0xf, 0xc>(tb);
    yc = dpp_f64<0x143, 0xc, 0xf>(tc);                    // row_bcast:31
    __builtin_amdgcn_sched_barrier(0);
    ta += ya, tb += yb, tc += yc;
    __builtin_amdgcn_sched_barrier(0);
    sa[0] = readlane_f64(ta, 15), sa[2] = readlane_f64(ta, 31), sa[1] = readlane_f64(ta, 47), sa[3] = readlane_f64(ta, 63);
    sb[0] = readlane_f64(tb, 15), sb[2] = readlane_f64(tb, 31), sb[1] = readlane_f64(tb, 47), sb[3] = readlane_f64(tb, 63);
    sc = readlane_f64(tc, 63);
}

// portable butterfly (ds_bpermute based) - used by the self test to cross-check the DPP ladder
__device__ __forceinline__ double wave_sum_shfl(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// RBF(+gradient) kernel block, SURVEY.md App. A.2.  r = x - x' (x: first argument / row, x': second / column)
//   q[d] = r_d / l_d^2, k = outputscale * exp(-1/2 sum r_d q_d)
//   cov(task_a(x), task_b(x')):  (0,0) k ; (0,b) +k q_b ; (a,0) -k q_a ; (a,b) k (delta_ab / l_a^2 - q_a q_b)
// ---------------------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ double kern_entry(const double (&q)[D], double k, const double* inv_l2, int a, int b) {
    if (a == 0) return (b == 0) ? k : k * q[b - 1];
    if (b == 0) return -k * q[a - 1];
    double v = -q[a - 1] * q[b - 1];
    if (a == b) v += inv_l2[a - 1];
    return k * v;
}

// q[d] = r_d / l_d^2 and the scaled squared distance sum_d r_d q_d (kern_scalar without the exponential)
template <int D>
__device__ __forceinline__ double kern_sqdist(const double* x, const double* xp, const double* inv_l2, double (&q)[D]) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const double r = x[d] - xp[d];
        q[d] = r * inv_l2[d];
        s += r * q[d];
    }
    return s;
}

template <int D>
__device__ __forceinline__ double kern_scalar(const double* x, const double* xp, const double* inv_l2, double os,
                                              double (&q)[D]) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const double r = x[d] - xp[d];
        q[d] = r * inv_l2[d];
        s += r * q[d];
    }
    return os * exp(-0.5 * s);
}

// ---------------------------------------------------------------------------------------------------------------
// Two to four exponentials of non-positive arguments in lockstep.  A single wave issues in order, so three separate exp()
// calls cost three full dependency chains (~25 dependent FP64 operations each, the scheduler does not interleave
// them); here the chains advance row by row (one operation of each per row, rows pinned with sched_barrier), the
// degree-11 polynomial is evaluated in Estrin form (depth 4 instead of 11) and the coefficient constants are
// materialised once for all three.  Cody-Waite reduction x = n ln2 + r, |r| <= ln2/2; coefficients = the minimax set of
// the ROCm device library's exp; result 2^n p(r) through v_ldexp_f64 (underflows to 0 for x < -745 by itself).
// Max. relative deviation from exp() over [-700, 0]: < 4e-16 (checked by gpmpc_selftest).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double bits_f64(unsigned long long u) { return __builtin_bit_cast(double, u); }
#define GPMPC_ROW3(stmt)               \
    _Pragma("unroll") for (int i_ = 0; i_ < N; ++i_) { stmt; } \
    __builtin_amdgcn_sched_barrier(0)
template <int N>
__device__ __forceinline__ void expn_neg(const double (&x)[N], double (&e)[N]) {
    static_assert(N >= 1 && N <= 4, "one to four chains");
    const double log2e = bits_f64(0x3FF71547652B82FEull), nln2h = bits_f64(0xBFE62E42FEFA39EFull),
                 nln2l = bits_f64(0xBC7ABC9E3B39803Full);
    const double c2 = bits_f64(0x3FE000000000000Bull), c3 = bits_f64(0x3FC5555555555511ull), c4 = bits_f64(0x3FA55555555502A1ull),
                 c5 = bits_f64(0x3F81111111122322ull), c6 = bits_f64(0x3F56C16C1852B7B0ull), c7 = bits_f64(0x3F2A01A014761F6Eull),
                 c8 = bits_f64(0x3EFA01997C89E6B0ull), c9 = bits_f64(0x3EC71DEE623FDE64ull), c10 = bits_f64(0x3E928AF3FCA7AB0Cull),
                 c11 = bits_f64(0x3E5ADE156A5DCB37ull);
    double n[N], r[N], r2[N], r4[N], a0[N], a1[N], a2[N], a3[N], a4[N], a5[N];
    __builtin_amdgcn_sched_barrier(0);
    GPMPC_ROW3(n[i_] = rint(x[i_] * log2e));
    GPMPC_ROW3(r[i_] = fma(n[i_], nln2h, x[i_]));
    GPMPC_ROW3(r[i_] = fma(n[i_], nln2l, r[i_]));
    GPMPC_ROW3(r2[i_] = r[i_] * r[i_]);
    GPMPC_ROW3(a0[i_] = 1.0 + r[i_]);
    GPMPC_ROW3(a1[i_] = fma(c3, r[i_], c2));
    GPMPC_ROW3(a2[i_] = fma(c5, r[i_], c4));
    GPMPC_ROW3(a3[i_] = fma(c7, r[i_], c6));
    GPMPC_ROW3(a4[i_] = fma(c9, r[i_], c8));
    GPMPC_ROW3(a5[i_] = fma(c11, r[i_], c10));
    GPMPC_ROW3(r4[i_] = r2[i_] * r2[i_]);
    GPMPC_ROW3(a0[i_] = fma(a1[i_], r2[i_], a0[i_]));       // b0 = (1 + r) + (c2 + c3 r) r^2
    GPMPC_ROW3(a2[i_] = fma(a3[i_], r2[i_], a2[i_]));       // b1
    GPMPC_ROW3(a4[i_] = fma(a5[i_], r2[i_], a4[i_]));       // b2
    GPMPC_ROW3(a2[i_] = fma(a4[i_], r4[i_], a2[i_]));       // b1 + b2 r^4
    GPMPC_ROW3(a0[i_] = fma(a2[i_], r4[i_], a0[i_]));       // p
    GPMPC_ROW3(e[i_] = ldexp(a0[i_], (int)n[i_]));
    // all results are due HERE: keeps the optimiser from sinking a chain to its (later) first use, out of the lockstep rows
    if constexpr (N == 4) asm volatile("" ::"v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]));
    else if constexpr (N == 3) asm volatile("" ::"v"(e[0]), "v"(e[1]), "v"(e[2]));
    else if constexpr (N == 2) asm volatile("" ::"v"(e[0]), "v"(e[1]));
}
__device__ __forceinline__ void exp3_neg(const double (&x)[3], double (&e)[3]) { expn_neg<3>(x, e); }

// ---------------------------------------------------------------------------------------------------------------
// small dense Cholesky (LAPACK dpotrf lower semantics: fail on pivot <= 0 or NaN), T x T, in registers
// ---------------------------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ bool chol_small(const double (&S)[T][T], double (&L)[T][T]) {
#pragma unroll
    for (int j = 0; j < T; ++j) {
        double d = S[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        L[j][j] = d;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < T; ++i) {
            double s = S[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s * inv;
        }
#pragma unroll
        for (int i = 0; i < j; ++i) L[i][j] = 0.0;
    }
    return true;
}

// Root of a T x T posterior covariance, SURVEY.md App. A.7:
//   1x1 -> sqrt (no jitter; NaN if negative);  else Cholesky, on failure up to 3 retries adding jitter*10^i (total)
//   to the diagonal.  Returns info bits (GPMPC_INFO_ROOT_*).  The whole-batch eigh fallback is the caller's job.
template <int T>
__device__ __forceinline__ int root_small(const double (&S)[T][T], double jitter, double (&R)[T][T]) {
    if constexpr (T == 1) {
        R[0][0] = sqrt(S[0][0]);
        return (S[0][0] < 0.0) ? GPMPC_INFO_NEG_1x1 : 0;
    } else {
        if (chol_small<T>(S, R)) return 0;
        double A[T][T];
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j) A[i][j] = S[i][j];
        double prev = 0.0, jn = jitter;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const double add = jn - prev;            // the library accumulates increments on a cloned matrix
#pragma unroll
            for (int i = 0; i < T; ++i) A[i][i] += add;
            prev = jn;
            if (chol_small<T>(A, R)) return (t + 1) << 1;
            jn *= 10.0;
        }
        return (3 << 1) | GPMPC_INFO_ROOT_FAIL;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// fast FP64 reciprocal square root: v_rsq_f64 seed + 2 Newton steps (~1 ulp); sqrt(x) = x * rsqrt(x) with one
// Heron correction.  Used for the T x T pivots (the generic sqrt/div sequences are ~5x longer and sit on the
// per-step critical path of a single wave).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double rsqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}
// returns sqrt(x), writes 1/sqrt(x)
__device__ __forceinline__ double sqrt_rsqrt_fast(double x, double& inv) {
    const double y = rsqrt_fast(x);
    double s = x * y;
    const double d = fma(-s, s, x);
    s = fma(d, 0.5 * y, s);
    inv = y;
    return s;
}

// Cholesky of a T x T matrix with rsqrt pivots; also returns 1/L_jj.  Same failure rule as chol_small.
template <int T>
__device__ __forceinline__ bool chol_small_fast(const double (&S)[T][T], double (&L)[T][T], double (&linv)[T]) {
#pragma unroll
    for (int j = 0; j < T; ++j) {
        double d = S[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = fma(-L[j][k], L[j][k], d);
        if (!(d > 0.0)) return false;
        double inv;
        L[j][j] = sqrt_rsqrt_fast(d, inv);
        linv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < T; ++i) {
            double s = S[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s = fma(-L[i][k], L[j][k], s);
            L[i][j] = s * inv;
        }
#pragma unroll
        for (int i = 0; i < j; ++i) L[i][j] = 0.0;
    }
    return true;
}

// Two 3 x 3 Choleskys (rsqrt pivots, as chol_small_fast) advancing in lockstep: the chain of one factorisation is ~28
// dependent FP64 operations, and a single in-order wave does not overlap two of them by itself.  Same operations in the
// same order per matrix as chol_small_fast => bit-identical factors; no early exit: ok[m] reports whether every pivot
// of matrix m was positive (the factor of a failed matrix is garbage, as it is unspecified there).
#define GPMPC_ROW2(stmt)                                        \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_) { stmt; }  \
    __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void chol3_pair_fast(const double (&A)[3][3], const double (&B)[3][3], double (&LA)[3][3],
                                                double (&LB)[3][3], double (&invA)[3], double (&invB)[3], bool& okA,
                                                bool& okB) {
    double s10[2] = {A[1][0], B[1][0]}, s20[2] = {A[2][0], B[2][0]}, s21[2] = {A[2][1], B[2][1]};
    double s11[2] = {A[1][1], B[1][1]}, s22[2] = {A[2][2], B[2][2]};
    double d0[2] = {A[0][0], B[0][0]}, d1[2], d2[2], y0[2], y1[2], y2[2], h[2], t[2], e[2];
    double l10[2], l20[2], l21[2], n21[2], p2[2], r0[2], r1[2], r2[2], dd[2], hy[2];
    __builtin_amdgcn_sched_barrier(0);
#define GPMPC_NEWTON(y)                                         \
    GPMPC_ROW2(t[m_] = h[m_] * y[m_]);                          \
    GPMPC_ROW2(e[m_] = fma(-t[m_], y[m_], 0.5));                \
    GPMPC_ROW2(y[m_] = fma(y[m_], e[m_], y[m_]));               \
    GPMPC_ROW2(t[m_] = h[m_] * y[m_]);                          \
    GPMPC_ROW2(e[m_] = fma(-t[m_], y[m_], 0.5));                \
    GPMPC_ROW2(y[m_] = fma(y[m_], e[m_], y[m_]))
    GPMPC_ROW2(y0[m_] = __builtin_amdgcn_rsq(d0[m_]); h[m_] = 0.5 * d0[m_]);
    GPMPC_NEWTON(y0);
    GPMPC_ROW2(l10[m_] = s10[m_] * y0[m_]; l20[m_] = s20[m_] * y0[m_]; r0[m_] = d0[m_] * y0[m_]);
    GPMPC_ROW2(d1[m_] = fma(-l10[m_], l10[m_], s11[m_]); n21[m_] = fma(-l20[m_], l10[m_], s21[m_]);
               dd[m_] = fma(-r0[m_], r0[m_], d0[m_]); hy[m_] = 0.5 * y0[m_]);
    GPMPC_ROW2(y1[m_] = __builtin_amdgcn_rsq(d1[m_]); h[m_] = 0.5 * d1[m_]; r0[m_] = fma(dd[m_], hy[m_], r0[m_]);
               p2[m_] = fma(-l20[m_], l20[m_], s22[m_]));
    GPMPC_NEWTON(y1);
    GPMPC_ROW2(l21[m_] = n21[m_] * y1[m_]; r1[m_] = d1[m_] * y1[m_]);
    GPMPC_ROW2(d2[m_] = fma(-l21[m_], l21[m_], p2[m_]); dd[m_] = fma(-r1[m_], r1[m_], d1[m_]); hy[m_] = 0.5 * y1[m_]);
    GPMPC_ROW2(y2[m_] = __builtin_amdgcn_rsq(d2[m_]); h[m_] = 0.5 * d2[m_]; r1[m_] = fma(dd[m_], hy[m_], r1[m_]));
    GPMPC_NEWTON(y2);
    GPMPC_ROW2(r2[m_] = d2[m_] * y2[m_]);
    GPMPC_ROW2(dd[m_] = fma(-r2[m_], r2[m_], d2[m_]); hy[m_] = 0.5 * y2[m_]);
    GPMPC_ROW2(r2[m_] = fma(dd[m_], hy[m_], r2[m_]));
#undef GPMPC_NEWTON
    // both chains are due here (otherwise the optimiser sinks the one whose results are used later out of the rows)
    asm volatile("" ::"v"(r0[0]), "v"(r1[0]), "v"(r2[0]), "v"(r0[1]), "v"(r1[1]), "v"(r2[1]));
    okA = (d0[0] > 0.0) && (d1[0] > 0.0) && (d2[0] > 0.0);
    okB = (d0[1] > 0.0) && (d1[1] > 0.0) && (d2[1] > 0.0);
    LA[0][0] = r0[0], LA[1][0] = l10[0], LA[2][0] = l20[0], LA[1][1] = r1[0], LA[2][1] = l21[0], LA[2][2] = r2[0];
    LB[0][0] = r0[1], LB[1][0] = l10[1], LB[2][0] = l20[1], LB[1][1] = r1[1], LB[2][1] = l21[1], LB[2][2] = r2[1];
    LA[0][1] = LA[0][2] = LA[1][2] = 0.0;
    LB[0][1] = LB[0][2] = LB[1][2] = 0.0;
    invA[0] = y0[0], invA[1] = y1[0], invA[2] = y2[0];
    invB[0] = y0[1], invB[1] = y1[1], invB[2] = y2[1];
}

// chol3_pair_fast (gpmpc_device.hpp) with ONE Newton step behind v_rsq_f64 instead of two (the seed is good to 2^-26, one
// step leaves 1.5 (2^-26)^2 = 3e-16 on 1 / sqrt(d)) and without the Heron correction of the pivots: 42 of the pair's ~95
// FP64 instructions less on the step's spine (one wave per SIMD: ~7 cycles each).  Used by rollout_one.hip (in rollout_tiles.hip it changed nothing measurable: its phase G is 5 % of a step).
#define GPMPC_ROW2L(stmt)                                          \
    _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_) { stmt; }  \
    __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void chol3_pair_lean(const double (&A)[3][3], const double (&B)[3][3], double (&LA)[3][3],
                                               double (&LB)[3][3], double (&invA)[3], double (&invB)[3], bool& okA, bool& okB) {
    double s10[2] = {A[1][0], B[1][0]}, s20[2] = {A[2][0], B[2][0]}, s21[2] = {A[2][1], B[2][1]};
    double s11[2] = {A[1][1], B[1][1]}, s22[2] = {A[2][2], B[2][2]};
    double d0[2] = {A[0][0], B[0][0]}, d1[2], d2[2], y0[2], y1[2], y2[2], h[2], t[2], e[2];
    double l10[2], l20[2], l21[2], n21[2], p2[2], r0[2], r1[2], r2[2];
    __builtin_amdgcn_sched_barrier(0);
#define GPMPC_NEWTON1(y)                                         \
    GPMPC_ROW2L(t[m_] = h[m_] * y[m_]);                          \
    GPMPC_ROW2L(e[m_] = fma(-t[m_], y[m_], 0.5));                \
    GPMPC_ROW2L(y[m_] = fma(y[m_], e[m_], y[m_]))
    GPMPC_ROW2L(y0[m_] = __builtin_amdgcn_rsq(d0[m_]); h[m_] = 0.5 * d0[m_]);
    GPMPC_NEWTON1(y0);
    // (the pivots themselves are d * y: y = 1 / sqrt(d) to 1.5 ulp after the Newton step, so sqrt(d) to ~2 ulp; the Heron
    // correction of chol3_pair_fast - four instructions per pivot and matrix on the step's spine - bought the last ulp)
    GPMPC_ROW2L(l10[m_] = s10[m_] * y0[m_]; l20[m_] = s20[m_] * y0[m_]; r0[m_] = d0[m_] * y0[m_]);
    GPMPC_ROW2L(d1[m_] = fma(-l10[m_], l10[m_], s11[m_]); n21[m_] = fma(-l20[m_], l10[m_], s21[m_]));
    GPMPC_ROW2L(y1[m_] = __builtin_amdgcn_rsq(d1[m_]); h[m_] = 0.5 * d1[m_]; p2[m_] = fma(-l20[m_], l20[m_], s22[m_]));
    GPMPC_NEWTON1(y1);
    GPMPC_ROW2L(l21[m_] = n21[m_] * y1[m_]; r1[m_] = d1[m_] * y1[m_]);
    GPMPC_ROW2L(d2[m_] = fma(-l21[m_], l21[m_], p2[m_]));
    GPMPC_ROW2L(y2[m_] = __builtin_amdgcn_rsq(d2[m_]); h[m_] = 0.5 * d2[m_]);
    GPMPC_NEWTON1(y2);
    GPMPC_ROW2L(r2[m_] = d2[m_] * y2[m_]);
#undef GPMPC_NEWTON1
    asm volatile("" ::"v"(r0[0]), "v"(r1[0]), "v"(r2[0]), "v"(r0[1]), "v"(r1[1]), "v"(r2[1]));
    okA = (d0[0] > 0.0) && (d1[0] > 0.0) && (d2[0] > 0.0);
    okB = (d0[1] > 0.0) && (d1[1] > 0.0) && (d2[1] > 0.0);
    LA[0][0] = r0[0], LA[1][0] = l10[0], LA[2][0] = l20[0], LA[1][1] = r1[0], LA[2][1] = l21[0], LA[2][2] = r2[0];
    LB[0][0] = r0[1], LB[1][0] = l10[1], LB[2][0] = l20[1], LB[1][1] = r1[1], LB[2][1] = l21[1], LB[2][2] = r2[1];
    LA[0][1] = LA[0][2] = LA[1][2] = 0.0;
    LB[0][1] = LB[0][2] = LB[1][2] = 0.0;
    invA[0] = y0[0], invA[1] = y1[0], invA[2] = y2[0];
    invB[0] = y0[1], invB[1] = y1[1], invB[2] = y2[1];
}

// the jitter-on-failure retries of root_small_fast (after a failed un-jittered attempt)
template <int T>
__device__ __forceinline__ int root_small_fast_retry(const double (&S)[T][T], double jitter, double (&R)[T][T]) {
    double linv[T], A[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) A[i][j] = S[i][j];
    double prev = 0.0, jn = jitter;
#pragma unroll 1
    for (int t = 0; t < 3; ++t) {
        const double add = jn - prev;
#pragma unroll
        for (int i = 0; i < T; ++i) A[i][i] += add;
        prev = jn;
        if (chol_small_fast<T>(A, R, linv)) return (t + 1) << 1;
        jn *= 10.0;
    }
    return (3 << 1) | GPMPC_INFO_ROOT_FAIL;
}

template <int T>
__device__ __forceinline__ int root_small_fast(const double (&S)[T][T], double jitter, double (&R)[T][T]) {
    double linv[T];
    if constexpr (T == 1) {
        R[0][0] = sqrt(S[0][0]);
        return (S[0][0] < 0.0) ? GPMPC_INFO_NEG_1x1 : 0;
    } else {
        if (chol_small_fast<T>(S, R, linv)) return 0;
        double A[T][T];
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j) A[i][j] = S[i][j];
        double prev = 0.0, jn = jitter;
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            const double add = jn - prev;
#pragma unroll
            for (int i = 0; i < T; ++i) A[i][i] += add;
            prev = jn;
            if (chol_small_fast<T>(A, R, linv)) return (t + 1) << 1;
            jn *= 10.0;
        }
        return (3 << 1) | GPMPC_INFO_ROOT_FAIL;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// environment maps fused into the rollout
// ---------------------------------------------------------------------------------------------------------------
// u = u_ff + K (x - x_goal), written as the reference does: -((x_goal - x) @ K^T) + u_ff
__device__ __forceinline__ void apply_feedback(const EnvParams& e, const double* x, const double* u_ff, double* u) {
    for (int i = 0; i < e.nu; ++i) {
        double acc = 0.0;
        if (e.use_feedback) {
            for (int j = 0; j < e.nx; ++j) acc += (e.x_goal[j] - x[j]) * e.K[i][j];
            u[i] = -acc + u_ff[i];
        } else {
            u[i] = u_ff[i];
        }
    }
}

// GP input selection: pendulum1D g_idx_inputs = [0, 2] (theta, u); car g_idx_inputs = [2, 4] (phi, delta)
__device__ __forceinline__ void gp_input(const EnvParams& e, const double* x, const double* u, double* xi) {
    if (e.env_id == GPMPC_ENV_PENDULUM1D) {
        xi[0] = x[0];
        xi[1] = u[0];
    } else {
        xi[0] = x[2];
        xi[1] = u[0];
    }
}

// x+ = f(x,u) + B_d(x) g   with g[o] = value component of output o's sample
__device__ __forceinline__ void env_step(const EnvParams& e, const double* x, const double* u, const double* g,
                                         double* xn) {
    if (e.env_id == GPMPC_ENV_PENDULUM1D) {
        xn[0] = x[0] + x[1] * e.dt;          // known_dyn: theta + omega dt
        xn[1] = x[1] + g[0];                 // omega + B_d g, B_d = [0, 1]^T
    } else {
        const double v = x[3];
        xn[0] = x[0] + v * g[0];             // transform_sensitivity multiplies the value by v, B_d = I_{4x3}
        xn[1] = x[1] + v * g[1];
        xn[2] = x[2] + v * g[2];
        xn[3] = (x[3] + u[1] * e.dt);        // known_dyn: v + a dt
    }
}

}  // namespace gpmpc
