// Minimal software wait states (s_nop N) for asm-issued FP64 4x4x4 MFMAs on gfx950, found empirically, and what they cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define STR2(x) #x
#define STR(x) STR2(x)
template <int N> struct Nop { static constexpr const char* s = ""; };
#define NOPSTR(N) "s_nop " STR(N) "\n\t"
// kind 0: accumulate chain (D of #1 = C of #2); 1: D of #1 read as SrcB by #2; 2: D read by a VALU op; 3: VALU write -> MFMA SrcB
template <int KIND, int N>
__global__ void k(const double* A, const double* B, double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    double a0 = A[l], a1 = A[64 + l], b0 = B[l], b1 = B[64 + l], acc = 1.0 + 0.01 * l, v = 0.0, w = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
            if constexpr (N < 0) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %3, %0\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %4, %0\n\t" : "+v"(acc) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
            else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %3, %0\n\ts_nop %5\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %4, %0\n\ts_nop %5\n\t" : "+v"(acc) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "n"(N < 0 ? 0 : N));
        } else if constexpr (KIND == 1) {
            if constexpr (N < 0) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, 0\n\tv_mfma_f64_4x4x4_4b_f64 %1, %2, %0, 0\n\ts_nop 7\n\ts_nop 7\n\t" : "=&v"(v), "=&v"(w) : "v"(a0), "v"(b0));
            else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, 0\n\ts_nop %4\n\tv_mfma_f64_4x4x4_4b_f64 %1, %2, %0, 0\n\ts_nop 7\n\ts_nop 7\n\t" : "=&v"(v), "=&v"(w) : "v"(a0), "v"(b0), "n"(N < 0 ? 0 : N));
        } else if constexpr (KIND == 2) {
            if constexpr (N < 0) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, 0\n\tv_fma_f64 %1, %0, %0, %2\n\t" : "=&v"(v), "=&v"(w) : "v"(a0), "v"(b0));
            else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %2, %3, 0\n\ts_nop %4\n\tv_fma_f64 %1, %0, %0, %2\n\t" : "=&v"(v), "=&v"(w) : "v"(a0), "v"(b0), "n"(N < 0 ? 0 : N));
        } else {
            if constexpr (N < 0) asm volatile("v_fma_f64 %0, %2, %3, %3\n\tv_mfma_f64_4x4x4_4b_f64 %1, %2, %0, 0\n\ts_nop 7\n\ts_nop 7\n\t" : "=&v"(v), "=&v"(w) : "v"(a0), "v"(b0));
            else asm volatile("v_fma_f64 %0, %2, %3, %3\n\ts_nop %4\n\tv_mfma_f64_4x4x4_4b_f64 %1, %2, %0, 0\n\ts_nop 7\n\ts_nop 7\n\t" : "=&v"(v), "=&v"(w) : "v"(a0), "v"(b0), "n"(N < 0 ? 0 : N));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    out[l] = acc; out[64 + l] = v; out[128 + l] = w;
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
template <int KIND, int N> static void run() {
    hipLaunchKernelGGL((k<KIND, N>), dim3(1), dim3(64), 0, 0, dA, dB, dO, dC, 1);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), dO, 192 * 8, hipMemcpyDeviceToHost);
    std::vector<double> acc(64), t(64), v(64), w(64);
    double err = 0;
    if (KIND == 0) { for (int i = 0; i < 64; ++i) acc[i] = 1.0 + 0.01 * i; mm(&A[0], &B[0], acc.data(), t.data()); mm(&A[64], &B[64], t.data(), acc.data()); for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[i] - acc[i])); }
    if (KIND == 1) { mm(&A[0], &B[0], nullptr, v.data()); mm(&A[0], v.data(), nullptr, w.data()); for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[128 + i] - w[i])); }
    if (KIND == 2) { mm(&A[0], &B[0], nullptr, v.data()); for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[128 + i] - fma(v[i], v[i], A[i]))); }
    if (KIND == 3) { for (int i = 0; i < 64; ++i) v[i] = fma(A[i], B[i], B[i]); mm(&A[0], v.data(), nullptr, w.data()); for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[128 + i] - w[i])); }
    hipLaunchKernelGGL((k<KIND, N>), dim3(1), dim3(64), 0, 0, dA, dB, dO, dC, 4000);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    std::printf("kind %d  s_nop %2d: max err %.2e  %s   %.1f cycles per iteration\n", KIND, N, err, err < 1e-9 ? "ok   " : "WRONG", c / 4000.0);
}
template <int KIND> static void sweep() {
    run<KIND, -1>(); run<KIND, 0>(); run<KIND, 1>(); run<KIND, 2>(); run<KIND, 3>(); run<KIND, 4>(); run<KIND, 5>(); run<KIND, 6>();
    run<KIND, 7>(); run<KIND, 9>(); run<KIND, 11>(); run<KIND, 13>(); run<KIND, 15>();
}
int main() {
    for (int i = 0; i < 128; ++i) { A[i] = std::sin(0.37 * i + 1.0); B[i] = std::cos(0.11 * i) + 0.5; }
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dO, 192 * 8); hipMalloc(&dC, 64);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    std::printf("kind 0: accumulate chain (2 MFMAs per iteration); 1: MFMA D -> MFMA SrcB; 2: MFMA D -> VALU read; 3: VALU write -> MFMA SrcB\n");
    sweep<0>(); sweep<1>(); sweep<2>(); sweep<3>();
    return 0;
}
