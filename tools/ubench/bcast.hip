// micro-benchmark: broadcasting 3 pivots x 3 right-hand sides (9 doubles) per block: v_readlane vs ds_bpermute
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double bp(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(l << 2, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(l << 2, __double2loint(v)));
}
template <int MODE>
__global__ void k_block(double* out, long long* cyc, int iters, double a) {
    double v[3] = {threadIdx.x * 1e-3, threadIdx.x * 1e-3 + 1, threadIdx.x * 1e-3 + 2};
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int p = (i * 6 + blk * 3) & 63;
            double sp[3][3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int b = 0; b < 3; ++b) sp[c][b] = (MODE == 0) ? rl(v[b], (p + c) & 63) : bp(v[b], (p + c) & 63);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int b = 0; b < 3; ++b) v[b] = fma(-a, sp[c][b], v[b]);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v[0] + v[1] + v[2];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 64);
    long long h; const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_block<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1e-9); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("readlane  block of 3 pivots x 3 rhs: %7.1f cycles per pivot\n", (double)h / (iters * 6.0));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_block<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1e-9); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("bpermute  block of 3 pivots x 3 rhs: %7.1f cycles per pivot\n", (double)h / (iters * 6.0));
    // occupancy sweep: does a second / third wave on the same SIMD hide the v_readlane cost?
    for (int waves = 1; waves <= 4; ++waves) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_block<0>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, iters, 1e-9); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("readlane, %d wave(s) per SIMD: %7.1f cycles per pivot per wave -> %6.1f cycles per pivot per SIMD-slot\n", waves, (double)h / (iters * 6.0), (double)h / (iters * 6.0) / waves);
    }
    return 0;
}
