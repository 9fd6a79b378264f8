// The pattern of the tiled solve: MFMA -> VALU op on its result -> MFMA reading the VALU result.  How many wait states
// (a) between the MFMA and the VALU reader, (b) between the VALU op and the second MFMA are needed for correct results?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
template <int NA, int NB>
__global__ void k(const double* A, const double* B, double* out) {
    const int l = threadIdx.x;
    double a = A[l], b = B[l], one = 1.0 + 0.001 * l, d1, t, d2;
    asm volatile("s_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %3, %4, 0\n\t"
                 "s_nop %6\n\t"
                 "v_add_f64 %1, %0, %5\n\t"          // VALU reads the MFMA result
                 "s_nop %7\n\t"
                 "v_mfma_f64_4x4x4_4b_f64 %2, %1, %4, 0\n\t"   // MFMA reads the VALU result as SrcA
                 "s_nop 7\n\ts_nop 7"
                 : "=&v"(d1), "=&v"(t), "=&v"(d2) : "v"(a), "v"(b), "v"(one), "n"(NA), "n"(NB));
    out[l] = d2;
}
// same with the VALU op overwriting a SOURCE of the first MFMA (as hipcc's register allocator does)
template <int NA, int NB>
__global__ void k2(const double* A, const double* B, double* out) {
    const int l = threadIdx.x;
    double a = A[l], b = B[l], one = 1.0 + 0.001 * l, d1, d2;
    asm volatile("s_nop 7\n\tv_mfma_f64_4x4x4_4b_f64 %0, %2, %3, 0\n\t"
                 "s_nop %5\n\t"
                 "v_add_f64 %2, %0, %4\n\t"          // reads D, WRITES the register that was SrcA
                 "s_nop %6\n\t"
                 "v_mfma_f64_4x4x4_4b_f64 %1, %2, %3, 0\n\t"
                 "s_nop 7\n\ts_nop 7"
                 : "=&v"(d1), "=&v"(d2), "+v"(a) : "v"(b), "v"(one), "n"(NA), "n"(NB));
    out[l] = d2;
}
std::vector<double> A(64), B(64), o(64), ref(64);
double *dA, *dB, *dO;
static void mm(const double* X, const double* Y, double* D) {
    for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int kk = 0; kk < 4; ++kk) s = fma(X[16 * kk + 4 * b + i], Y[16 * kk + 4 * b + j], s);
        D[16 * i + 4 * b + j] = s;
    }
}
template <int V, int NA, int NB> static void run() {
    if (V == 0) hipLaunchKernelGGL((k<NA, NB>), dim3(1), dim3(64), 0, 0, dA, dB, dO);
    else hipLaunchKernelGGL((k2<NA, NB>), dim3(1), dim3(64), 0, 0, dA, dB, dO);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), dO, 512, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[i] - ref[i]));
    std::printf("%s: s_nop %2d before the VALU reader, s_nop %2d before the second MFMA: max err %.2e %s\n", V ? "WAR form " : "plain form", NA, NB, err, err < 1e-9 ? "ok" : "WRONG");
}
template <int V> static void sweep() {
    run<V, 0, 0>(); run<V, 0, 1>(); run<V, 0, 3>(); run<V, 0, 5>(); run<V, 0, 7>(); run<V, 0, 11>(); run<V, 0, 15>();
    run<V, 3, 1>(); run<V, 5, 1>(); run<V, 7, 1>(); run<V, 11, 1>(); run<V, 15, 1>(); run<V, 7, 5>(); run<V, 15, 5>();
}
int main() {
    for (int i = 0; i < 64; ++i) { A[i] = std::sin(0.37 * i + 1.0); B[i] = std::cos(0.11 * i) + 0.5; }
    std::vector<double> d1(64), t(64);
    mm(A.data(), B.data(), d1.data());
    for (int i = 0; i < 64; ++i) t[i] = d1[i] + (1.0 + 0.001 * i);
    mm(t.data(), B.data(), ref.data());
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dO, 512);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    sweep<0>(); sweep<1>();
    return 0;
}
