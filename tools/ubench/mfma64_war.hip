// WAR: how soon after issuing an FP64 4x4x4 MFMA may a VALU instruction overwrite one of its SOURCE registers (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
// WHICH 0: overwrite SrcA, 1: SrcB, 2: SrcC (D != C)
template <int WHICH, int N>
__global__ void k(const double* A, const double* B, double* out) {
    const int l = threadIdx.x;
    double a = A[l], b = B[l], c = 1.0 + 0.01 * l, d, junk = 777.0 + l;
    if constexpr (WHICH == 0) {
        if constexpr (N < 0) asm volatile("s_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3\n\tv_mov_b64 %1, %4\n\ts_nop 7\n\ts_nop 7" : "=&v"(d), "+v"(a) : "v"(b), "v"(c), "v"(junk));
        else asm volatile("s_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3\n\ts_nop %5\n\tv_mov_b64 %1, %4\n\ts_nop 7\n\ts_nop 7" : "=&v"(d), "+v"(a) : "v"(b), "v"(c), "v"(junk), "n"(N < 0 ? 0 : N));
    } else if constexpr (WHICH == 1) {
        if constexpr (N < 0) asm volatile("s_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %3\n\tv_mov_b64 %1, %4\n\ts_nop 7\n\ts_nop 7" : "=&v"(d), "+v"(b) : "v"(a), "v"(c), "v"(junk));
        else asm volatile("s_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %1, %3\n\ts_nop %5\n\tv_mov_b64 %1, %4\n\ts_nop 7\n\ts_nop 7" : "=&v"(d), "+v"(b) : "v"(a), "v"(c), "v"(junk), "n"(N < 0 ? 0 : N));
    } else {
        if constexpr (N < 0) asm volatile("s_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %1\n\tv_mov_b64 %1, %4\n\ts_nop 7\n\ts_nop 7" : "=&v"(d), "+v"(c) : "v"(a), "v"(b), "v"(junk));
        else asm volatile("s_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %3, %1\n\ts_nop %5\n\tv_mov_b64 %1, %4\n\ts_nop 7\n\ts_nop 7" : "=&v"(d), "+v"(c) : "v"(a), "v"(b), "v"(junk), "n"(N < 0 ? 0 : N));
    }
    out[l] = d;
}
std::vector<double> A(64), B(64), o(64), ref(64);
double *dA, *dB, *dO;
template <int W, int N> static void run() {
    hipLaunchKernelGGL((k<W, N>), dim3(1), dim3(64), 0, 0, dA, dB, dO);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), dO, 512, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[i] - ref[i]));
    std::printf("overwrite %s after s_nop %2d: max err %.2e %s\n", W == 0 ? "SrcA" : (W == 1 ? "SrcB" : "SrcC"), N, err, err < 1e-9 ? "ok" : "WRONG");
}
template <int W> static void sweep() { run<W, -1>(); run<W, 0>(); run<W, 1>(); run<W, 2>(); run<W, 3>(); run<W, 4>(); run<W, 5>(); run<W, 7>(); }
int main() {
    for (int i = 0; i < 64; ++i) { A[i] = std::sin(0.37 * i + 1.0); B[i] = std::cos(0.11 * i) + 0.5; }
    for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 1.0 + 0.01 * (16 * i + 4 * b + j);
        for (int kk = 0; kk < 4; ++kk) s = fma(A[16 * kk + 4 * b + i], B[16 * kk + 4 * b + j], s);
        ref[16 * i + 4 * b + j] = s;
    }
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dO, 512);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    sweep<0>(); sweep<1>(); sweep<2>();
    return 0;
}
