// accuracy of v_rsq_f64 and of rsqrt_fast (seed + 2 Newton steps) over 600 decades
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../../sampling_gpmpc_amd/csrc/gpmpc_device.hpp"
__global__ void k(const double* x, double* raw, double* fast, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { raw[i] = __builtin_amdgcn_rsq(x[i]); fast[i] = gpmpc::rsqrt_fast(x[i]); }
}
int main() {
    const int n = 1 << 16;
    double *hx = new double[n], *hr = new double[n], *hf = new double[n];
    for (int i = 0; i < n; ++i) hx[i] = pow(10.0, -300.0 + 600.0 * i / n) * (1.0 + 0.37 * ((i * 7919) % 100) / 100.0);
    double *dx, *dr, *df;
    hipMalloc(&dx, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&df, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dr, df, n);
    hipMemcpy(hr, dr, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hf, df, n * 8, hipMemcpyDeviceToHost);
    double wr = 0, wf = 0; int ir = 0, iff = 0;
    for (int i = 0; i < n; ++i) {
        const double ref = 1.0 / sqrt(hx[i]);
        const double er = fabs(hr[i] - ref) / ref, ef = fabs(hf[i] - ref) / ref;
        if (er > wr) { wr = er; ir = i; }
        if (ef > wf) { wf = ef; iff = i; }
    }
    printf("v_rsq_f64 worst rel err %.3e at x=%.3e ; rsqrt_fast worst %.3e at x=%.3e\n", wr, hx[ir], wf, hx[iff]);
    // by decade ranges
    for (int lo = -300; lo < 300; lo += 50) {
        double w1 = 0, w2 = 0;
        for (int i = 0; i < n; ++i) { double l = log10(hx[i]); if (l >= lo && l < lo + 50) { const double ref = 1.0 / sqrt(hx[i]); w1 = fmax(w1, fabs(hr[i] - ref) / ref); w2 = fmax(w2, fabs(hf[i] - ref) / ref); } }
        printf("  1e%d..1e%d: raw %.2e fast %.2e\n", lo, lo + 50, w1, w2);
    }
    return 0;
}
