// How many software wait states remain to be padded when INDEPENDENT MFMAs stand between an FP64 4x4x4 MFMA and the reader of its
// result (gfx950).  tools/ubench/mfma64_hazard2.hip found, with s_nop only: D -> VALU read 6 wait states, D -> MFMA SrcA/B 6,
// D -> SrcC 4 (one wait state = 4 cycles: s_nop 5 -> 7 costs 8 cycles).  rollout_one.hip fills those slots with the next tile
// row's MFMAs instead of s_nop.
//   kind 4: MFMA D ; F independent MFMAs ; s_nop N ; v_fma_f64 reads D
//   kind 5: MFMA D ; F independent MFMAs ; s_nop N ; v_mov_b32_dpp reads D's low dword (row_ror:4)
//   kind 6: MFMA D ; F independent MFMAs ; s_nop N ; MFMA reads D as SrcB
// (physical registers: D = v[100:101], fills v[102:105], reader's result v[106:107])
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#define MF "v_mfma_f64_4x4x4_4b_f64 "
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"
template <int KIND, int F, int N>
__global__ void k(const double* A, const double* B, double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    double a0 = A[l], a1 = A[64 + l], b0 = B[l], b1 = B[64 + l], w = 0.0;
    asm volatile("v_mov_b64 v[100:101], 0\n\tv_mov_b64 v[102:103], 0\n\tv_mov_b64 v[104:105], 0\n\tv_mov_b64 v[106:107], 0\n\ts_nop 7" : : : CLOB);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile(MF "v[100:101], %0, %1, 0" : : "v"(a0), "v"(b0) : CLOB);
        if constexpr (F >= 1) asm volatile(MF "v[102:103], %0, %1, 0" : : "v"(a1), "v"(b1) : CLOB);
        if constexpr (F >= 2) asm volatile(MF "v[104:105], %0, %1, 0" : : "v"(a0), "v"(b1) : CLOB);
        if constexpr (N >= 0) asm volatile("s_nop %0" : : "n"(N < 0 ? 0 : N) : CLOB);
        if constexpr (KIND == 4) asm volatile("v_fma_f64 v[106:107], v[100:101], v[100:101], %0" : : "v"(a0) : CLOB);
        if constexpr (KIND == 5) asm volatile("v_mov_b32_dpp v106, v100 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32 v107, 0" : : : CLOB);
        if constexpr (KIND == 6) asm volatile(MF "v[106:107], %0, v[100:101], 0" : : "v"(a0) : CLOB);
        asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b64 %0, v[106:107]" : "=v"(w) : : CLOB);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    out[128 + l] = w;
}
static void mm(const double* A, const double* B, const double* C, double* D) {
    for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = C ? C[16 * i + 4 * b + j] : 0.0;
        for (int kk = 0; kk < 4; ++kk) s = fma(A[16 * kk + 4 * b + i], B[16 * kk + 4 * b + j], s);
        D[16 * i + 4 * b + j] = s;
    }
}
std::vector<double> A(128), B(128), o(192);
double *dA, *dB, *dO; long long* dC;
template <int KIND, int F, int N> static void run() {
    hipLaunchKernelGGL((k<KIND, F, N>), dim3(1), dim3(64), 0, 0, dA, dB, dO, dC, 1);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), dO, 192 * 8, hipMemcpyDeviceToHost);
    std::vector<double> v(64), w(64);
    double err = 0;
    mm(&A[0], &B[0], nullptr, v.data());
    if (KIND == 4) for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[128 + i] - fma(v[i], v[i], A[i])));
    if (KIND == 5) for (int i = 0; i < 64; ++i) {           // row_ror:4: lane i of a row gets lane (i - 4) mod 16; low dword only
        const int src = (i & ~15) | ((i - 4) & 15);
        long long bits, got;
        std::memcpy(&bits, &v[src], 8);
        std::memcpy(&got, &o[128 + i], 8);
        err = fmax(err, ((unsigned)(bits & 0xffffffffll) == (unsigned)(got & 0xffffffffll)) ? 0.0 : 1.0);
    }
    if (KIND == 6) { mm(&A[0], v.data(), nullptr, w.data()); for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[128 + i] - w[i])); }
    hipLaunchKernelGGL((k<KIND, F, N>), dim3(1), dim3(64), 0, 0, dA, dB, dO, dC, 4000);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    std::printf("kind %d  fill %d  s_nop %2d: max err %.2e  %s   %.1f cycles per iteration\n", KIND, F, N, err, err < 1e-9 ? "ok   " : "WRONG", c / 4000.0);
}
template <int KIND, int F> static void sweep() {
    run<KIND, F, -1>(); run<KIND, F, 0>(); run<KIND, F, 1>(); run<KIND, F, 2>(); run<KIND, F, 3>(); run<KIND, F, 4>(); run<KIND, F, 5>(); run<KIND, F, 6>();
}
int main() {
    for (int i = 0; i < 128; ++i) { A[i] = std::sin(0.37 * i + 1.0); B[i] = std::cos(0.11 * i) + 0.5; }
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dO, 192 * 8); hipMalloc(&dC, 64);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    std::printf("kind 4: MFMA D, F independent MFMAs, s_nop N, VALU reads D; 5: ... DPP mov reads D; 6: ... MFMA reads D as SrcB\n");
    sweep<4, 0>(); sweep<4, 1>(); sweep<4, 2>();
    sweep<5, 0>(); sweep<5, 1>(); sweep<5, 2>();
    sweep<6, 0>(); sweep<6, 1>(); sweep<6, 2>();
    return 0;
}
