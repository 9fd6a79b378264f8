// Round-robin accumulate chains: 16 MFMAs over NACC accumulators with a given gap between consecutive MFMAs - which
// combinations are correct (MFMA D -> SrcC needs 4 wait states; does an MFMA in between count as one?) and what they cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define M(c) "v_mfma_f64_4x4x4_4b_f64 %" #c ", %4, %5, %" #c "\n\t"
template <int NACC, int GAP>   // GAP: -1 none, else s_nop GAP after every MFMA; GAP = 100 + g: s_nop g after every second MFMA
__global__ void k(const double* A, const double* B, double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    double a = A[l], b = B[l], c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define G1 "s_nop %6\n\t"
        if constexpr (NACC == 4 && GAP == -1) asm volatile(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) "s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "n"(0));
        else if constexpr (NACC == 4 && GAP < 100) asm volatile(M(0) G1 M(1) G1 M(2) G1 M(3) G1 M(0) G1 M(1) G1 M(2) G1 M(3) G1 M(0) G1 M(1) G1 M(2) G1 M(3) G1 M(0) G1 M(1) G1 M(2) G1 M(3) "s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "n"(GAP < 0 ? 0 : GAP));
        else if constexpr (NACC == 4) asm volatile(M(0) M(1) G1 M(2) M(3) G1 M(0) M(1) G1 M(2) M(3) G1 M(0) M(1) G1 M(2) M(3) G1 M(0) M(1) G1 M(2) M(3) "s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "n"(GAP - 100));
        else if constexpr (NACC == 2 && GAP == -1) asm volatile(M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) "s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "n"(0));
        else asm volatile(M(0) G1 M(1) G1 M(0) G1 M(1) G1 M(0) G1 M(1) G1 M(0) G1 M(1) G1 M(0) G1 M(1) G1 M(0) G1 M(1) G1 M(0) G1 M(1) G1 M(0) G1 M(1) "s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "n"(GAP < 0 ? 0 : GAP));
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    out[l] = (c0 + c1) + (c2 + c3);
}
std::vector<double> A(64), B(64), o(64), ref(64);
double *dA, *dB, *dO; long long* dC;
template <int NACC, int GAP> static void run(const char* name) {
    hipLaunchKernelGGL((k<NACC, GAP>), dim3(1), dim3(64), 0, 0, dA, dB, dO, dC, 1);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), dO, 512, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 64; ++i) err = fmax(err, fabs(o[i] - ref[i]) / fabs(ref[i]));
    hipLaunchKernelGGL((k<NACC, GAP>), dim3(1), dim3(64), 0, 0, dA, dB, dO, dC, 2000);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    std::printf("%d accumulators, %-34s: max rel err %.2e %s  %.1f cycles per MFMA\n", NACC, name, err, err < 1e-12 ? "ok   " : "WRONG", c / 2000.0 / 16.0);
}
int main() {
    for (int i = 0; i < 64; ++i) { A[i] = std::sin(0.37 * i + 1.0); B[i] = std::cos(0.11 * i) + 0.5; }
    for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int kk = 0; kk < 4; ++kk) s = fma(A[16 * kk + 4 * b + i], B[16 * kk + 4 * b + j], s);
        ref[16 * i + 4 * b + j] = 16.0 * s;
    }
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dO, 512); hipMalloc(&dC, 64);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    run<4, -1>("no gap"); run<4, 100>("s_nop 0 after every second MFMA"); run<4, 101>("s_nop 1 after every second MFMA");
    run<4, 0>("s_nop 0 after every MFMA"); run<4, 1>("s_nop 1 after every MFMA");
    run<2, -1>("no gap"); run<2, 0>("s_nop 0 after every MFMA"); run<2, 1>("s_nop 1 after every MFMA"); run<2, 2>("s_nop 2 after every MFMA"); run<2, 3>("s_nop 3 after every MFMA");
    return 0;
}
