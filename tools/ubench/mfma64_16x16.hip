// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: cycles per MFMA for (a) IND independent accumulators in turn, (b) a dependent
// chain on one accumulator (SrcC = own vDst), with W waves per SIMD.  2048 FLOP per MFMA: 64 cycles would be the 78.6 TFLOP/s
// vendor peak (32 FLOP / cycle / SIMD).
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/mfma64_16x16.hip -o tools/ubench/mfma64_16x16.bin && tools/ubench/mfma64_16x16.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int IND>
__global__ __launch_bounds__(512) void k(const double* A, double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    const double a = A[l & 63], b = A[64 + (l & 63)];
    d4 acc[IND];
#pragma unroll
    for (int i = 0; i < IND; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < IND; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    const long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < IND; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int IND> static void run(int waves_per_simd, int blocks, const double* dA, double* dO, long long* dC) {
    const int iters = 2000, threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<IND>), dim3(blocks), dim3(threads), 0, 0, dA, dO, dC, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<IND>), dim3(blocks), dim3(threads), 0, 0, dA, dO, dC, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    const double n = 4.0 * IND * iters;
    const double flops = (double)blocks * (threads / 64) * n * 2048.0 / (ms * 1e-3);
    printf("IND=%d waves/SIMD=%d blocks=%4d: %7.1f cycles per MFMA per wave (%6.1f per SIMD slot), %.3f ms, %.1f TFLOP/s, clock %.2f GHz\n", IND,
           waves_per_simd, blocks, c / n, c / n / waves_per_simd, ms, flops / 1e12, c / (ms * 1e-3) / 1e9);
}
int main() {
    double hA[128];
    for (int i = 0; i < 128; ++i) hA[i] = 1e-3 * (i % 7);
    double *dA, *dO; long long* dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dO, 8 * 512 * 1024); hipMalloc(&dC, 8);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) {
        run<1>(w, 256, dA, dO, dC);
        run<2>(w, 256, dA, dO, dC);
        run<4>(w, 256, dA, dO, dC);
    }
    run<1>(1, 1, dA, dO, dC);
    run<4>(1, 1, dA, dO, dC);
    run<4>(2, 1, dA, dO, dC);
    return 0;
}
