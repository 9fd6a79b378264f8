// Probe the lane maps of v_mfma_f64_4x4x4_4b_f64 (gfx950): one-hot A lanes against distinct B values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* D) {   // D[la][lane]
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        const double a = (l == la) ? 1.0 : 0.0, b = 1.0 + l;
        double acc = 0.0;
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
        D[la * 64 + l] = acc;
    }
}
int main() {
    double* d;
    hipMalloc(&d, 64 * 64 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipDeviceSynchronize();
    std::vector<double> h(64 * 64);
    hipMemcpy(h.data(), d, 64 * 64 * 8, hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        std::printf("A lane %2d ->", la);
        for (int l = 0; l < 64; ++l)
            if (h[la * 64 + l] != 0.0) std::printf(" D[%2d]=B[%2d]", l, (int)h[la * 64 + l] - 1);
        std::printf("\n");
    }
    return 0;
}
