// Sizing of a car variant of rollout_one_kernel (profiles/r5_one_car_sizing.md): what a tile row of the one-chain-per-wave solve costs
// a LONE wave (one wave per SIMD, nothing to hide behind) when its panels - the A operands of the row's MFMAs - come out of
//   kind 0: registers (the shipped pendulum kernel: the whole factor pinned in AGPRs),
//   kind 1: LDS, one ds_read_b64 per panel, requested ONE ROW AHEAD into a second register set (software prefetch inside the wave),
//   kind 2: LDS, requested at the start of the row that uses them (no register set to spare).
// A row here = G dependent-by-two-accumulators v_mfma_f64_4x4x4_4b_f64 (the real row: g + 2 of them) + a tail of 10 dependent VALU
// instructions (cross-block sum, W = G acc, merge) whose result is the next row's B operand - the serial spine of the real solve.
//   hipcc --offload-arch=gfx950 -O3 -o one_panel_source.bin one_panel_source.hip && ./one_panel_source.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND, int G>
__global__ __launch_bounds__(64) void k(const double* P, double* out, long long* cyc, int rows) {
    __shared__ double lds[2 * G * 64];
    const int l = threadIdx.x;
    double pan[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        pan[g] = P[g * 64 + l];
        lds[g * 64 + l] = pan[g];
        lds[(G + g) * 64 + l] = pan[g] * 0.5;
    }
    __syncthreads();
    double b = 1.0 + 1e-3 * l, w = 0.0;
    double nxt[G];
    if (KIND == 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) nxt[g] = lds[g * 64 + l];
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rows; ++r) {
        double a[G];
        const int set = (r & 1) * G;
        if (KIND == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) a[g] = pan[g];
        } else if (KIND == 1) {
#pragma unroll
            for (int g = 0; g < G; ++g) a[g] = nxt[g];
            // the next row's panels: requested now, used one row later
#pragma unroll
            for (int g = 0; g < G; ++g) nxt[g] = lds[(G - set + g) * 64 + l];
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) a[g] = lds[(set + g) * 64 + l];
        }
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g & 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[g], b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[g], b, acc0, 0, 0, 0);
        }
        double s = acc0 + acc1;
        // ten dependent VALU instructions (the real tail: two DPP row rotations + adds, the diagonal-tile MFMA's feed, the merge)
#pragma unroll
        for (int u = 0; u < 9; ++u) s = __builtin_fma(s, 0.999, 1e-9);
        b = s * 1e-3 + 1.0;
        w += s;
        asm volatile("" : "+v"(b));
    }
    const long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + l] = w;
}

template <int KIND, int G>
static void run(const char* name, double* dP, double* dO, long long* dC) {
    const int rows = 4096, blocks = 1024;        // one wave per SIMD on 256 CUs: every wave is alone on its SIMD
    hipLaunchKernelGGL((k<KIND, G>), dim3(blocks), dim3(64), 0, 0, dP, dO, dC, rows);
    hipLaunchKernelGGL((k<KIND, G>), dim3(blocks), dim3(64), 0, 0, dP, dO, dC, rows);
    hipDeviceSynchronize();
    std::vector<long long> c(blocks);
    hipMemcpy(c.data(), dC, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v;
    mean /= blocks;
    printf("%-46s G = %2d MFMAs per row: %7.1f cycles per row (%5.1f per MFMA + tail)\n", name, G, mean / rows, mean / rows / G);
}

int main() {
    double *dP, *dO;
    long long* dC;
    std::vector<double> P(16 * 64);
    for (size_t i = 0; i < P.size(); ++i) P[i] = 1e-3 * (double)((i * 37) % 101);
    hipMalloc(&dP, P.size() * 8);
    hipMalloc(&dO, 1024 * 64 * 8);
    hipMalloc(&dC, 1024 * 8);
    hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    run<0, 6>("panels in registers", dP, dO, dC);
    run<1, 6>("panels from LDS, requested one row ahead", dP, dO, dC);
    run<2, 6>("panels from LDS, requested in their own row", dP, dO, dC);
    run<0, 9>("panels in registers", dP, dO, dC);
    run<1, 9>("panels from LDS, requested one row ahead", dP, dO, dC);
    run<2, 9>("panels from LDS, requested in their own row", dP, dO, dC);
    run<0, 12>("panels in registers", dP, dO, dC);
    run<1, 12>("panels from LDS, requested one row ahead", dP, dO, dC);
    run<2, 12>("panels from LDS, requested in their own row", dP, dO, dC);
    return 0;
}
