// Do asm-issued FP64 MFMAs need software wait states between dependent instructions on gfx950?  Chains with and without
// s_nop, checked against a host evaluation.  (hipcc pads builtins by itself; inline asm is not padded.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
template <int NOPS>
__global__ void chain(const double* A, const double* B, const double* G, double* out) {
    const int l = threadIdx.x;
    double a0 = A[l], a1 = A[64 + l], a2 = A[128 + l], b0 = B[l], b1 = B[64 + l], b2 = B[128 + l], g = G[l];
    double acc = 1.0 + 0.01 * l, v, s = 0.0, w;
    if constexpr (NOPS) {
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %3, %6, %0\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mfma_f64_4x4x4_4b_f64 %0, %4, %7, %0\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mfma_f64_4x4x4_4b_f64 %0, %5, %8, %0\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mfma_f64_4x4x4_4b_f64 %1, %9, %0, 0\n\ts_nop 7\n\ts_nop 7\n\t"        // acc as SrcB
                     "v_mfma_f64_4x4x4_4b_f64 %2, %1, %1, 0\n\ts_nop 7\n\ts_nop 7\n\t"        // v as SrcA and SrcB
                     "v_fma_f64 %10, %2, %2, %1\n\t"                                          // VALU reads both results
                     : "+v"(acc), "=&v"(v), "=&v"(s), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(g), "=&v"(w));
    } else {
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %3, %6, %0\n\t"
                     "v_mfma_f64_4x4x4_4b_f64 %0, %4, %7, %0\n\t"
                     "v_mfma_f64_4x4x4_4b_f64 %0, %5, %8, %0\n\t"
                     "v_mfma_f64_4x4x4_4b_f64 %1, %9, %0, 0\n\t"
                     "v_mfma_f64_4x4x4_4b_f64 %2, %1, %1, 0\n\t"
                     "v_fma_f64 %10, %2, %2, %1\n\t"
                     : "+v"(acc), "=&v"(v), "=&v"(s), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(g), "=&v"(w));
    }
    out[l] = acc;
    out[64 + l] = v;
    out[128 + l] = s;
    out[192 + l] = w;
}
// layout: A[i][k] lane 16k+4b+i, B[k][j] lane 16k+4b+j, D[i][j] lane 16i+4b+j
static void mm(const double* A, const double* B, const double* C, double* D) {
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = C ? C[16 * i + 4 * b + j] : 0.0;
                for (int k = 0; k < 4; ++k) s = fma(A[16 * k + 4 * b + i], B[16 * k + 4 * b + j], s);
                D[16 * i + 4 * b + j] = s;
            }
}
int main() {
    std::vector<double> A(192), B(192), G(64), acc(64), t(64), v(64), s(64), w(64), o(256);
    for (int i = 0; i < 192; ++i) { A[i] = std::sin(0.37 * i + 1.0); B[i] = std::cos(0.11 * i) + 0.5; }
    for (int i = 0; i < 64; ++i) { G[i] = 1.0 / (1.0 + (i % 7)); acc[i] = 1.0 + 0.01 * i; }
    mm(&A[0], &B[0], acc.data(), t.data()); acc = t;
    mm(&A[64], &B[64], acc.data(), t.data()); acc = t;
    mm(&A[128], &B[128], acc.data(), t.data()); acc = t;
    mm(G.data(), acc.data(), nullptr, v.data());
    mm(v.data(), v.data(), nullptr, s.data());
    for (int i = 0; i < 64; ++i) w[i] = fma(s[i], s[i], v[i]);
    double *dA, *dB, *dG, *dO;
    hipMalloc(&dA, 192 * 8); hipMalloc(&dB, 192 * 8); hipMalloc(&dG, 512); hipMalloc(&dO, 256 * 8);
    hipMemcpy(dA, A.data(), 192 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 192 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dG, G.data(), 512, hipMemcpyHostToDevice);
    for (int nops = 0; nops < 2; ++nops) {
        if (nops) hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, dA, dB, dG, dO);
        else hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, dA, dB, dG, dO);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), dO, 256 * 8, hipMemcpyDeviceToHost);
        double e[4] = {0, 0, 0, 0};
        for (int i = 0; i < 64; ++i) {
            e[0] = fmax(e[0], fabs(o[i] - acc[i]));
            e[1] = fmax(e[1], fabs(o[64 + i] - v[i]));
            e[2] = fmax(e[2], fabs(o[128 + i] - s[i]));
            e[3] = fmax(e[3], fabs(o[192 + i] - w[i]));
        }
        std::printf("%s: max abs err accumulate chain %.3e, diag (acc as SrcB) %.3e, gram (v as SrcA/B) %.3e, VALU reader %.3e\n",
                    nops ? "with s_nop 7 x2 after every MFMA" : "no software wait states     ", e[0], e[1], e[2], e[3]);
    }
    return 0;
}
