// Cycles of rollout_one.hip's forward substitution (one_solve<7>: 22 tile rows, 138 MFMAs) for a LONE wave per CU, per schedule
// variant of tools/gen_rollout_one.py (--variant ...): what the row chain costs and what stands on it.
//   hipcc --offload-arch=gfx950 -O3 -DVARIANT_INC='"one_gen_<v>.inc"' -o one_solve_chain_<v>.bin one_solve_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
namespace gpmpc {
#include VARIANT_INC
__global__ __launch_bounds__(64, 1) void k(const double* in, double* out, long long* cyc, int iters, int nh) {
    const int l = threadIdx.x;
    OnePanels P;
    one_init(P, in[l] * 1e-3);                       // (diagonal tiles tiny: the iteration stays bounded)
    double Vu[8], RN[8];
    const int bm = (l >> 2) & 3;
    const double MK[4] = {(bm == 0) ? 1.0 : 0.0, (bm == 1) ? 1.0 : 0.0, (bm == 2) ? 1.0 : 0.0, (bm == 3) ? 1.0 : 0.0};
    for (int g = 0; g < 8; ++g) Vu[g] = in[64 * g + l], RN[g] = in[64 * (8 + g) + l];
    long long best = 1ll << 60;
    for (int rep = 0; rep < 3; ++rep) {
        const long long t0 = __builtin_readcyclecounter();
        double S0 = 0, S1 = 0;
        for (int it = 0; it < iters; ++it) one_solve<7>(P, Vu, RN, MK, nh, S0, S1);
        Vu[0] += 1e-300 * (S0 + S1);
        const long long t1 = __builtin_readcyclecounter();
        best = (t1 - t0 < best) ? t1 - t0 : best;
    }
    if (l == 0 && blockIdx.x == 0) cyc[0] = best;
    double s = 0;
    for (int g = 0; g < 8; ++g) s += Vu[g];
    out[blockIdx.x * 64 + l] = s;
}
}
int main() {
    std::vector<double> h(64 * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01 * ((i * 37) % 17) - 0.05;
    double *din, *dout; long long* dc;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, 256 * 64 * 8); hipMalloc(&dc, 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const int iters = 200;
    for (int nh : {88, 80}) {
        for (int blocks : {1, 256}) {
            hipLaunchKernelGGL(gpmpc::k, dim3(blocks), dim3(64), 0, 0, din, dout, dc, iters, nh);
            hipDeviceSynchronize();
            long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
            const int rows = (nh + 3) / 4;
            printf("n_h %2d (%2d rows) blocks %3d: %8.1f cycles / solve, %6.1f / row\n", nh, rows, blocks, (double)c / iters, (double)c / iters / rows);
        }
    }
    return 0;
}
